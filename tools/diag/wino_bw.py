"""Bandwidth of the Winograd transform kernels vs the plane stride (16 planes T*K*4 bytes apart): does the stride's
alignment to the HBM channel interleave matter?  usage: python tools/diag/wino_bw.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpr2021_vspw_implement_amd import _C, ops
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for n, h, w, K, dil in ((10, 60, 60, 256, 2), (10, 60, 60, 288, 2), (10, 60, 60, 320, 2), (10, 60, 62, 256, 2), (10, 58, 60, 256, 2), (10, 60, 60, 512, 2), (10, 60, 60, 544, 2)):
    x = ops.empty_nhwc(n, K, h, w, dev).normal_()
    d = ops._conv_desc(x, K, 3, 3, 1, dil, dil)
    T = int(_C.query("vspw_wino_tiles", ctypes.byref(d)))
    v = torch.empty(16, T, K, device=dev); m = torch.randn(16, T, K, device=dev); y = ops.empty_nhwc(n, K, h, w, dev)
    part = torch.empty(_C.query("vspw_wino_stat_partials", ctypes.byref(d)), 2, K, device=dev)
    ti = timeit(lambda: _C.call("vspw_wino_input", ctypes.byref(d), x.data_ptr(), K, v.data_ptr(), st))
    to = timeit(lambda: _C.call("vspw_wino_output", ctypes.byref(d), m.data_ptr(), K, None, y.data_ptr(), None, None, None, None, part.data_ptr(), None, 0, st))
    bi = (x.numel() + v.numel()) * 4; bo = (m.numel() + y.numel()) * 4
    print("n%d %dx%d K=%d T=%d plane stride %d B (mod 32K = %5d): input %.1f us %.2f TB/s | output %.1f us %.2f TB/s"
          % (n, h, w, K, T, T * K * 4, (T * K * 4) % 32768, ti, bi / ti / 1e6, to, bo / to / 1e6))
