"""DIAGNOSTIC (-DVSPW_NT_TIMING=4): clock and duration of ONE GEMM launch as a function of what the chip did before it:
back-to-back launches, launches separated by idle gaps, launches separated by an HBM-bound kernel (the training step's mix)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cvpr2021_vspw_implement_amd import _C
dev = torch.device("cuda:0"); lib = _C.load()
st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.vspw_debug_nt_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
M, N, K = 36000, 256, 1024
a = torch.randn(1, M, K, device=dev); b = torch.randn(1, N, K, device=dev); c = torch.empty(1, M, N, device=dev)
big = torch.randn(64 * 1024 * 1024, device=dev); big2 = torch.empty_like(big)
f = lambda: _C.call("vspw_bmm_nt", a.data_ptr(), b.data_ptr(), c.data_ptr(), 1, M, N, K, st_)
def clock():
    buf = np.zeros(8192 * 5, dtype=np.uint64); lib.vspw_debug_nt_stamps(buf.ctypes.data, buf.size)
    st = buf.reshape(-1, 5); st = st[st[:, 0] > 0]; t = st[:, :4].astype(np.int64)
    t = t[t[:, 1] > t[:, 1].max() - 100000]
    return float(np.median((t[:, 3] - t[:, 0]) / ((t[:, 2] - t[:, 1]) * 10.0)))
def one():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for _ in range(50): f()
torch.cuda.synchronize()
for what, between in (("back to back (50 launches queued before)", lambda: [f() for _ in range(50)]),
                      ("after 5 ms of idle", lambda: (torch.cuda.synchronize(), time.sleep(0.005))),
                      ("after 50 ms of idle", lambda: (torch.cuda.synchronize(), time.sleep(0.05))),
                      ("after a 0.5 GB device copy (HBM-bound, ~0.2 ms)", lambda: big2.copy_(big)),
                      ("alternating with a 0.25 GB copy for 50 rounds", lambda: [(f(), big2[:32 * 1024 * 1024].copy_(big[:32 * 1024 * 1024])) for _ in range(50)])):
    r = []
    for _ in range(5):
        between()
        us = one(); r.append((us, clock()))
    r = np.array(r)
    print("%-50s: %.1f us (%.1f TFLOP/s) at %.3f GHz" % (what, np.median(r[:, 0]), 2.0 * M * N * K / np.median(r[:, 0]) / 1e6, np.median(r[:, 1])))
