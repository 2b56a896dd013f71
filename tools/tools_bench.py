"""Throughput of the evaluation-side / data-preparation pieces: the 1080p -> 480p resize of
change2_480p (device resampling kernels vs PIL on one host core, decode / encode excluded) and one frame pair of the
temporal-consistency metric (RAFT 20 iterations at 480 x 853 + nearest flow-warp)."""
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr2021_vspw_implement_amd import TC_cal  # noqa: E402
from cvpr2021_vspw_implement_amd.dataset2 import DeviceTransform, FrameSpec  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    frame = rs.randint(0, 256, (1080, 1920, 3)).astype(np.uint8)
    mask = rs.randint(0, 125, (1080, 1920)).astype(np.uint8)
    tf = DeviceTransform(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    spec = FrameSpec(1080, 1920, 0, (480, 853))
    f_dev, m_dev = tf._dev(frame), tf._dev(mask)
    for _ in range(3):
        out = tf._resize(f_dev, m_dev, spec, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        out = tf._resize(f_dev, m_dev, spec, st)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(5):
        a = Image.fromarray(frame, "RGB").resize((853, 480), Image.BILINEAR)
        b = Image.fromarray(mask, "L").resize((853, 480), Image.NEAREST)
    pil_ms = (time.perf_counter() - t0) / 5 * 1e3
    same = bool(np.array_equal(out[0].cpu().numpy(), np.array(a)) and np.array_equal(out[1].cpu().numpy(), np.array(b)))
    print(json.dumps({"workload": "change2_480p: 1920x1080 frame + mask -> 853x480 (resize only)",
                      "device_ms_per_frame": round(gpu_ms, 3), "pil_ms_per_frame_one_core": round(pil_ms, 2),
                      "identical_to_pil": same}))
    torch.manual_seed(0)
    model = TC_cal.load_raft("", dev)
    i1 = torch.from_numpy(rs.randint(0, 256, (1, 480, 853, 3)).astype(np.uint8)).to(dev).float().permute(0, 3, 1, 2)
    i2 = torch.from_numpy(rs.randint(0, 256, (1, 480, 853, 3)).astype(np.uint8)).to(dev).float().permute(0, 3, 1, 2)
    lab = torch.from_numpy(rs.randint(0, 124, (1, 1, 480, 853)).astype(np.float32)).to(dev)

    def pair():
        flow = TC_cal.pair_flow(model, i1, i2)
        return TC_cal.flowwarp(lab, flow)

    for _ in range(2):
        pair()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        w = pair()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps({"workload": "TC_cal: one frame pair at 480x853 (RAFT 20 iterations + nearest flow-warp)",
                      "ms_per_pair": round(ms, 2), "pairs_per_s": round(1e3 / ms, 1), "finite": bool(torch.isfinite(w).all().item())}))


if __name__ == "__main__":
    main()
