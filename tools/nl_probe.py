"""Throughput of the fused non-local dot kernel at the cfg 5b size (B=2 clips, N = 7*60*60 positions, C = 128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cvpr2021_vspw_implement_amd import ops
dev = torch.device("cuda:0")
for B, N, C in ((2, 25200, 128), (2, 3600, 128)):
    q, k, v = (torch.randn(B, N, C, device=dev) for _ in range(3))
    for _ in range(2): y = ops._nl_dot(q, k, v, 1.0 / N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): y = ops._nl_dot(q, k, v, 1.0 / N)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("nl_dot B%d N%d C%d: %.3f ms  %.1f TFLOP/s (4*B*N^2*C flop)" % (B, N, C, ms, 4.0 * B * N * N * C / ms / 1e9))
