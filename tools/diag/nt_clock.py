"""DIAGNOSTIC (-DVSPW_NT_TIMING=4): the rate of s_memtime against the 100 MHz wall clock INSIDE a GEMM launch = the shader
clock the chip sustains under that load (prologue and epilogue lie between the two stamp pairs: lives >> those)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cvpr2021_vspw_implement_amd import _C
dev = torch.device("cuda:0"); lib = _C.load()
st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.vspw_debug_nt_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for M, N, K, zero in [(36000, 256, 1024, 0), (36000, 256, 1024, 1), (36000, 2048, 512, 0), (36000, 256, 4096, 0)]:
    a = torch.randn(1, M, K, device=dev); b = torch.randn(1, N, K, device=dev); c = torch.empty(1, M, N, device=dev)
    if zero: a.zero_(); b.zero_()
    f = lambda: _C.call("vspw_bmm_nt", a.data_ptr(), b.data_ptr(), c.data_ptr(), 1, M, N, K, st_)
    for _ in range(20): f()
    torch.cuda.synchronize()
    f(); torch.cuda.synchronize()
    buf = np.zeros(8192 * 5, dtype=np.uint64); lib.vspw_debug_nt_stamps(buf.ctypes.data, buf.size)
    st = buf.reshape(-1, 5); st = st[st[:, 0] > 0]; t = st[:, :4].astype(np.int64)
    t = t[t[:, 1] > t[:, 1].max() - 200000]
    ghz = (t[:, 3] - t[:, 0]) / ((t[:, 2] - t[:, 1]) * 10.0)  # ticks per ns
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print("%d x %d x %d %s: s_memtime runs at %.3f GHz (median over %d workgroups, p5 %.3f p95 %.3f); %.1f us per launch, %.1f TFLOP/s"
          % (M, N, K, "ZERO operands" if zero else "random operands", np.median(ghz), len(ghz), np.percentile(ghz, 5), np.percentile(ghz, 95), us, 2.0 * M * N * K / us / 1e6))
