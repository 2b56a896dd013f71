"""Asymptotic throughput of the conv GEMM kernels on plain GEMM shapes (1x1 convs), to separate steady-state loop
efficiency from prologue/epilogue/tail losses."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cvpr2021_vspw_implement_amd import ops, _C
dev = torch.device("cuda:0")
def run(M, N, K, iters=10):
    x = torch.randn(1, K, M, 1, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(N, K, 1, 1, device=dev) * 0.01
    for _ in range(3):
        y, _, d = ops.conv2d_forward(x, w, None, 1, 0, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y, _, d = ops.conv2d_forward(x, w, None, 1, 0, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("fwd  M%6d N%5d K%5d  %.3f ms  %.1f TFLOP/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
    dy = torch.randn_like(y)
    for _ in range(2): dw = ops.conv2d_backward_weight(dy, x, d)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): dw = ops.conv2d_backward_weight(dy, x, d)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("wgrad M%6d N%5d K%5d  %.3f ms  %.1f TFLOP/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
for shp in [(8192, 4096, 8192), (16384, 2048, 4096), (36000, 1024, 256), (36000, 256, 1024), (36000, 2048, 2048), (36864, 1024, 1024)]:
    run(*shp)
