"""DIAGNOSTIC (library built with -DVSPW_NT_TIMING): how evenly do the co-resident workgroups of ONE GEMM launch progress?
Per workgroup: start, loop start, loop end, end (s_memtime ticks, per-XCD clocks).  Prints the distribution of workgroup
lives and of their end times relative to the launch's first start - a single-round launch (tiles <= slots) ends when its
SLOWEST workgroup ends."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cvpr2021_vspw_implement_amd import _C
dev = torch.device("cuda:0")
lib = _C.load()
st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = [int(v) for v in sys.argv[1:4]]
a = torch.randn(1, M, K, device=dev); b = torch.randn(1, N, K, device=dev); c = torch.empty(1, M, N, device=dev)
for _ in range(3):
    _C.call("vspw_bmm_nt", a.data_ptr(), b.data_ptr(), c.data_ptr(), 1, M, N, K, st_)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    _C.call("vspw_bmm_nt", a.data_ptr(), b.data_ptr(), c.data_ptr(), 1, M, N, K, st_)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
buf = np.zeros(8192 * 5, dtype=np.uint64)
lib.vspw_debug_nt_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.vspw_debug_nt_stamps(buf.ctypes.data, buf.size)
st = buf.reshape(-1, 5); st = st[st[:, 0] > 0]
xcc = (st[:, 4] >> np.uint64(32)).astype(np.int64); hwid = (st[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64)
t = st[:, :4].astype(np.int64)
print("%d x %d x %d: %d workgroups, %.1f us per launch (%.1f TFLOP/s), env %s" % (M, N, K, len(t), us, 2.0 * M * N * K / us / 1e6,
      {k: v for k, v in os.environ.items() if k.startswith("VSPW_") and k != "VSPW_HIP_LIB"}))
pc = lambda v: "min %d p10 %d p50 %d p90 %d max %d" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
for x in range(8):
    m = xcc == x
    if m.sum() == 0: continue
    tx = t[m]
    # stamps of the LAST launch only: starts within 2x the median life of the latest start
    life = tx[:, 3] - tx[:, 0]
    keep = tx[:, 0] > tx[:, 0].max() - 3 * np.median(life)
    tx = tx[keep]; life = life[keep]
    t0 = tx[:, 0].min()
    cu = ((hwid[m][keep] >> 8) & 0xF) + 16 * ((hwid[m][keep] >> 12) & 1) + 32 * ((hwid[m][keep] >> 13) & 0x7)
    span = tx[:, 3].max() - t0
    busy = life.sum() / float(len(np.unique(cu)))  # workgroup-ticks per CU
    print("XCD %d: %3d WGs, span %d ticks; starts %s; lives %s; ends %s; mean residency over the span %.2f"
          % (x, len(tx), span, pc(tx[:, 0] - t0), pc(life), pc(tx[:, 3] - t0), busy / span))
