"""Video-consistency score VC_n of saved predictions, mirroring the reference's VC_perclip.py: over every window of
`clip_num` frames of every video of the split, the share of pixels whose ground truth is constant across the window that
are also predicted constantly and correctly-constant (utils.get_common, the same definition test_clip2.py uses); videos
with at most `clip_num` frames are skipped; the score is the nan-mean over all windows.  Host-side numpy on PNG files, as
in the reference; its hard-coded paths (VC_perclip.py:28-38) are flags."""
import argparse
import os

import numpy as np
from PIL import Image

from .utils import get_common


def main(args, log=print):
    with open(os.path.join(args.dataroot, args.split), "r") as f:
        videos = [line[:-1] for line in f.readlines()]
    total_acc = []
    for video in videos:
        if video[0] == ".":
            continue
        images = sorted(os.listdir(os.path.join(args.dataroot, "data", video, "mask")))
        if len(images) <= args.clip_num:
            continue
        gts, preds = [], []
        h = w = 0
        for name in images:
            if name[0] == ".":
                continue
            gt = Image.open(os.path.join(args.dataroot, "data", video, "mask", name))
            w, h = gt.size
            gts.append(np.array(gt))
            preds.append(np.array(Image.open(os.path.join(args.pred, video, name))))
        accs = get_common(gts, preds, args.clip_num, h, w)
        log(sum(accs) / len(accs))
        total_acc.extend(accs)
    acc = np.nanmean(np.array(total_acc))
    log(args.pred)
    log("*" * 10)
    log("VC{} score: {} on {} set".format(args.clip_num, acc, args.split))
    log("*" * 10)
    return acc


def build_parser():
    p = argparse.ArgumentParser(description="video consistency (VC_n) of saved VSPW predictions")
    p.add_argument("--dataroot", type=str, default="/your/path/to/VSPW_480p")
    p.add_argument("--pred", type=str, default="./predicts")
    p.add_argument("--split", type=str, default="val.txt")
    p.add_argument("--clip_num", type=int, default=16)
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
