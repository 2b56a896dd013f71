"""DIAGNOSTIC (library built with -DVSPW_NT_TIMING=3: wall-clock stamps, 100 MHz, common to all XCDs): how much of a
GEMM launch's duration lies OUTSIDE the interval in which its workgroups run?  ONE launch at a time, synchronised, so
that the stamps of the table belong to the launch that was timed."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cvpr2021_vspw_implement_amd import _C
dev = torch.device("cuda:0")
lib = _C.load()
st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.vspw_debug_nt_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for M, N, K in [(36000, 256, 1024), (36000, 1024, 256), (36000, 256, 256), (36000, 2048, 512)]:
    a = torch.randn(1, M, K, device=dev); b = torch.randn(1, N, K, device=dev); c = torch.empty(1, M, N, device=dev)
    f = lambda: _C.call("vspw_bmm_nt", a.data_ptr(), b.data_ptr(), c.data_ptr(), 1, M, N, K, st_)
    for _ in range(5): f()
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        buf = np.zeros(8192 * 5, dtype=np.uint64)
        lib.vspw_debug_nt_stamps(buf.ctypes.data, buf.size)
        st = buf.reshape(-1, 5); st = st[st[:, 0] > 0]; t = st[:, :4].astype(np.int64)
        t = t[t[:, 0] > t[:, 0].max() - 100000]  # this launch (stale slots of larger earlier grids dropped)
        span = (t[:, 3].max() - t[:, 0].min()) / 100.0  # us
        ramp = (np.percentile(t[:, 0], 90) - t[:, 0].min()) / 100.0
        life = np.median(t[:, 3] - t[:, 0]) / 100.0
        res.append((e0.elapsed_time(e1) * 1e3, span, ramp, life, len(t)))
    r = np.array(res)[1:].mean(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) * 1e3 / 20
    print("%5d x %4d x %4d: %4d WGs; events around ONE launch %.1f us; back to back %.1f us per launch (%.1f TFLOP/s); workgroups "
          "run for %.1f us (first start -> last end; 90 %% started after %.1f us; median life %.1f us) = %.0f %% of the back-to-back time"
          % (M, N, K, r[4], r[0], b2b, 2.0 * M * N * K / b2b / 1e6, r[1], r[2], r[3], 100 * r[1] / b2b))
