"""VSPW -> VSPW_480p, mirroring the reference's change2_480p.py: every frame of every video resized to height 480 and
width int(480 * w / h) - frames with PIL's bilinear filter, label masks with nearest neighbour - into the same tree layout
(`data/<video>/origin/*.jpg`, `data/<video>/mask/*.png`).

The resampling itself runs on the GPU with the kernels of the input pipeline (csrc/data.hip: Pillow's 8-bit two-pass
resize with its fixed-point coefficient tables, bit-exact; tests/test_tools_gpu.py); decode and encode stay on the host
with PIL, so the files written equal the reference's.  Its hard-coded directories (change2_480p.py:6-8) are flags."""
import argparse
import os

import numpy as np
import torch
from PIL import Image

from .dataset2 import DeviceTransform, FrameSpec


def target_size(w, h):
    """change2_480p.py:17: (width, height) of the 480p frame."""
    return int(480 * w / h), 480


def change(transform, src, dst, video, image):
    """change2_480p.py:11-32 for one frame (and its mask when there is one)."""
    img = Image.open(os.path.join(src, "data", video, "origin", image))
    w, h = img.size
    nw, nh = target_size(w, h)
    stem = image.split(".")[0]
    mask_path = os.path.join(src, "data", video, "mask", stem + ".png")
    has_mask = os.path.isfile(mask_path)
    frame = np.array(img)
    mask_img = Image.open(mask_path) if has_mask else None
    seg = np.array(mask_img) if has_mask else np.zeros((h, w), np.uint8)
    if frame.ndim != 3 or frame.shape[2] != 3 or seg.dtype != np.uint8 or seg.ndim != 2:
        raise ValueError("%s/%s: expected an RGB frame and a single-channel 8-bit mask" % (video, image))
    st = torch.cuda.current_stream(transform.device).cuda_stream
    spec = FrameSpec(h, w, 0, (nh, nw))
    out, out_seg = transform._resize(transform._dev(frame), transform._dev(seg), spec, st)
    os.makedirs(os.path.join(dst, "data", video, "origin"), exist_ok=True)
    Image.fromarray(out.cpu().numpy(), "RGB").save(os.path.join(dst, "data", video, "origin", image))
    if has_mask:
        os.makedirs(os.path.join(dst, "data", video, "mask"), exist_ok=True)
        res = Image.fromarray(out_seg.cpu().numpy(), mask_img.mode if mask_img.mode in ("L", "P") else "L")
        if mask_img.mode == "P" and mask_img.getpalette() is not None:
            res.putpalette(mask_img.getpalette())
        res.save(os.path.join(dst, "data", video, "mask", stem + ".png"))


def main(args, log=print):
    transform = DeviceTransform(torch.device("cuda", args.gpu))
    for video in sorted(os.listdir(os.path.join(args.src, "data"))):
        if video[0] == ".":
            continue
        for image in sorted(os.listdir(os.path.join(args.src, "data", video, "origin"))):
            if image[0] == ".":
                continue
            change(transform, args.src, args.dst, video, image)
            log("Processing video {} image {}".format(video, image))
    log("finish")


def build_parser():
    p = argparse.ArgumentParser(description="resize a VSPW tree to 480p")
    p.add_argument("--src", type=str, default="/your/path/to/VSPW")
    p.add_argument("--dst", type=str, default="/your/path/to/VSPW_480p")
    p.add_argument("--gpu", type=int, default=0)
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
