"""Time the frozen RAFT flow network on the HIP kernels as NetWarp runs it: B=2 frame pairs, 480x856 (479x853 zero-padded
to multiples of 8), iters=20, test_mode=True.  Prints one JSON line (ms per forward, frame pairs / s)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr2021_vspw_implement_amd.models import raft as raft_mod  # noqa: E402
from cvpr2021_vspw_implement_amd.models.raft import RAFT  # noqa: E402


def main():
    B, H, W, iters = 2, 480, 856, 20
    dev = torch.device("cuda:0")
    torch.manual_seed(304)
    m = RAFT().to(dev).eval()
    a = (torch.rand(B, 3, H, W, device=dev) * 255).float()
    b = torch.roll(a, (3, -2), (2, 3))
    for _ in range(2):
        m(a, b, iters=iters, test_mode=True)
    torch.cuda.synchronize()
    reps = 5
    raft_mod.GEMM_FLOPS["total"] = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        low, up = m(a, b, iters=iters, test_mode=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    gflop = raft_mod.GEMM_FLOPS["total"] / reps / 1e9  # convolutions + the all-pairs correlation GEMM
    print(json.dumps({"workload": "RAFT-basic forward, B=2 pairs, 480x856, iters=20", "ms_per_forward": round(ms, 2),
                      "pairs_per_s": round(B / ms * 1e3, 2), "gemm_gflop_per_forward": round(gflop, 1),
                      "effective_tflops": round(gflop / ms, 1), "frac_of_fp32_mfma_peak": round(gflop / ms / 157.3, 3),
                      "finite": bool(torch.isfinite(up).all().item())}))


if __name__ == "__main__":
    main()
