"""Time the two thin convolutions of RAFT's update block at its size (2 x 60 x 107 pixels): implicit GEMM vs direct."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpr2021_vspw_implement_amd import _C
from cvpr2021_vspw_implement_amd._C import ConvDesc
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
n, h, w = 2, 60, 107
for name, c, k, kh, pad in (("flow_head.conv2 256->2 3x3", 256, 2, 3, 1), ("convf1 2->128 7x7", 2, 128, 7, 3)):
    x = torch.randn(n, h, w, c, device=dev); wt = torch.randn(k, kh, kh, c, device=dev) * 0.05; b = torch.randn(k, device=dev)
    y = torch.empty(n, h, w, k, device=dev); y2 = torch.empty_like(y)
    d = ConvDesc(n, h, w, c, h, w, k, kh, kh, 1, pad, 1, pad)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    t0 = timeit(lambda: _C.call("vspw_conv2d_fwd_ex", ctypes.byref(d), p(x), c, p(wt), p(b), None, 1, p(y), k, st))
    if not int(_C.query("vspw_conv2d_thin_supported", ctypes.byref(d), c, k)):
        print("%s: implicit GEMM %.1f us (no direct form: the LDS filter-bank kernel tried in round 5 took 73 us)" % (name, t0))
        continue
    t1 = timeit(lambda: _C.call("vspw_conv2d_thin", ctypes.byref(d), p(x), c, p(wt), p(b), 1, p(y2), k, st))
    print("%s: implicit GEMM %.1f us, direct %.1f us, max |diff| %.2e" % (name, t0, t1, float((y - y2).abs().max())))
