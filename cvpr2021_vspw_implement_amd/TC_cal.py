"""Temporal-consistency score of saved predictions, mirroring the reference's TC_cal.py: for every pair of consecutive
frames of a video the frozen RAFT estimates the flow from frame t to t+1 (20 iterations, frames zero-padded to
multiples of 8), the prediction of frame t+1 is carried back onto frame t with a NEAREST flow-warp, and the agreement of
prediction t with it is accumulated in one confusion matrix; TC = its mIoU (TC_cal.py:66-119).

Here the flow network and the warp run in HIP (models/raft.py, csrc/raft.hip, vspw_flowwarp_nearest); the file
handling and the confusion matrix stay on the host as in the reference.  The reference's hard-coded paths
(TC_cal.py:44-48,59) are flags."""
import argparse
import os
from collections import OrderedDict

import numpy as np
import torch
from PIL import Image

from . import ops
from .models.netwarp import _pad_to_8
from .utils import Evaluator


def flowwarp(x, flo):
    """TC_cal.py:12-38: grid_sample(x, meshgrid + flo, mode='nearest', align_corners=False) with the (dim - 1)
    normalisation of the reference."""
    return ops.flowwarp_nearest(x, flo)


def load_raft(weights, device):
    """TC_cal.py:61-68: RAFT() with the `module.`-prefixed checkpoint."""
    from .models.raft import RAFT

    model = RAFT()
    if weights:
        to_load = torch.load(weights, map_location="cpu")
        model.load_state_dict(OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in to_load.items()))
    return model.to(device).eval()


def _frame(path, device):
    a = np.array(Image.open(path))
    return torch.from_numpy(a).to(device).unsqueeze(0).float().permute(0, 3, 1, 2)  # [1,3,H,W], values 0..255, NHWC memory


def _labels(path, device):
    return torch.from_numpy(np.array(Image.open(path))).to(device).unsqueeze(0).unsqueeze(0).float()


def pair_flow(model, image1, image2):
    """TC_cal.py:88-99: InputPadder pad -> RAFT(iters=20, test_mode=True) -> unpad."""
    with torch.no_grad():
        a, pad = _pad_to_8(image1)
        b, _ = _pad_to_8(image2)
        _, flow = model(a, b, iters=20, test_mode=True)
        hh, ww = flow.shape[-2:]
        return ops.plane_shift(flow, (hh - pad[2] - pad[3], ww - pad[0] - pad[1]), -pad[2], -pad[0])


def main(args, model=None, log=print):
    device = torch.device("cuda", args.gpu)
    torch.cuda.set_device(device)
    if model is None:
        model = load_raft(args.raft_weights, device)
    with open(os.path.join(args.dataroot, args.split), "r") as f:
        videos = [v[:-1] for v in f.readlines()]
    evaluator = Evaluator(args.num_class)
    for video in videos[:args.max_videos]:
        if video[0] == ".":
            continue
        frames = sorted(os.listdir(os.path.join(args.dataroot, "data", video, "origin")))
        for i, name in enumerate(frames[:-1]):
            if name[0] == ".":
                continue
            nxt = frames[i + 1]
            flow = pair_flow(model, _frame(os.path.join(args.dataroot, "data", video, "origin", name), device),
                             _frame(os.path.join(args.dataroot, "data", video, "origin", nxt), device))
            pred = np.array(Image.open(os.path.join(args.pred, video, name.split(".")[0] + ".png")))
            next_pred = _labels(os.path.join(args.pred, video, nxt.split(".")[0] + ".png"), device)
            warp_pred = flowwarp(next_pred, flow).int().squeeze(1).cpu().numpy()
            evaluator.add_batch(pred[None], warp_pred)
    TC = evaluator.Mean_Intersection_over_Union()
    log("TC score is {}".format(TC))
    log(args.split)
    log(args.pred)
    return TC


def build_parser():
    p = argparse.ArgumentParser(description="temporal consistency (TC) of saved VSPW predictions")
    p.add_argument("--dataroot", type=str, default="/your/path/to/VSPW_480p")
    p.add_argument("--pred", type=str, default="./prediction")
    p.add_argument("--split", type=str, default="val.txt")
    p.add_argument("--num_class", type=int, default=124)
    p.add_argument("--raft_weights", type=str, default="./RAFT_core/raft-things.pth-no-zip")
    p.add_argument("--max_videos", type=int, default=100)
    p.add_argument("--gpu", type=int, default=0)
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
