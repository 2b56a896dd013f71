"""DIAGNOSTIC (library built with -DVSPW_NT_TIMING): residency over time of one short-K GEMM launch - how many
workgroups does a CU hold, how long do they live, where is the time between the ideal 3 rounds and the measured span?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cvpr2021_vspw_implement_amd import _C
dev = torch.device("cuda:0")
lib = _C.load()
st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (144000, 256, 256))]
a = torch.randn(1, M, K, device=dev); b = torch.randn(1, N, K, device=dev); c = torch.empty(1, M, N, device=dev)
for _ in range(3):
    _C.call("vspw_bmm_nt", a.data_ptr(), b.data_ptr(), c.data_ptr(), 1, M, N, K, st_)
torch.cuda.synchronize()
buf = np.zeros(8192 * 5, dtype=np.uint64)
lib.vspw_debug_nt_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.vspw_debug_nt_stamps(buf.ctypes.data, buf.size)
st = buf.reshape(-1, 5); print("rows with stamps:", int((st[:, 0] > 0).sum()), "of", len(st)); st = st[st[:, 0] > 0]
xcc = (st[:, 4] >> np.uint64(32)).astype(np.int64); hwid = (st[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64)
t = st[:, :4].astype(np.int64)
# (the cycle counters of the XCDs are not synchronised: everything below is per XCD)
print("%d x %d x %d: %d workgroups stamped" % (M, N, K, len(t)))
for x in range(8):
    m = xcc == x
    if m.sum() == 0:
        continue
    tx = t[m] - t[m][:, 0].min()
    cu = ((hwid[m] >> 8) & 0xF) + 16 * ((hwid[m] >> 12) & 1) + 32 * ((hwid[m] >> 13) & 0x7)
    span = tx[:, 3].max()
    life = tx[:, 3] - tx[:, 0]
    ncu = len(np.unique(cu))
    grid = np.linspace(0, span, 21)[:-1]
    res = [((tx[:, 0] <= g) & (g < tx[:, 3])).sum() / float(ncu) for g in grid]
    print("XCD %d: %3d WGs on %2d CUs, span %7d cyc, life median %6d (prologue %5d loop %6d epilogue %5d); WGs per CU over time: %s"
          % (x, m.sum(), ncu, span, np.median(life), np.median(tx[:, 1] - tx[:, 0]), np.median(tx[:, 2] - tx[:, 1]),
             np.median(tx[:, 3] - tx[:, 2]), " ".join("%.1f" % r for r in res)))
    if x == 0:
        per = np.bincount(cu)
        print("   WGs per CU: min %d max %d" % (per[per > 0].min(), per.max()))
        k0 = cu[0]; seq = tx[cu == k0]; seq = seq[np.argsort(seq[:, 0])]
        for r in seq[:12]: print("      %7d %7d %7d %7d" % tuple(r))
