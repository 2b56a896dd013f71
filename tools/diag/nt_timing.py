"""DIAGNOSTIC (library built with VSPW_CXXFLAGS=-DVSPW_NT_TIMING): per-workgroup phase timing of one NT launch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cvpr2021_vspw_implement_amd import _C, ops
dev = torch.device("cuda:0")
lib = _C.load()
def run(name, n, h, w, c, k, ks, pad, dil, mode):
    x = ops.empty_nhwc(n, c, h, w, dev).normal_()
    wt = (torch.randn(k, ks, ks, c, device=dev) * 0.05).permute(0, 3, 1, 2)
    y, part, d = ops.conv2d_forward(x, wt, None, 1, pad, dil, want_stats=True)
    dy = torch.randn_like(y)
    for _ in range(3):
        if mode == "fwd": ops.conv2d_forward(x, wt, None, 1, pad, dil, want_stats=True)
        else: ops.conv2d_backward_data(dy, wt, d)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 5, dtype=np.uint64)
    lib.vspw_debug_nt_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = lib.vspw_debug_nt_stamps(buf.ctypes.data, buf.size)
    st = buf.reshape(-1, 5)
    st = st[st[:, 0] > 0]
    xcc = (st[:, 4] >> np.uint64(32)).astype(np.int64)
    hwid = (st[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    rel = np.zeros((len(st), 4))
    for x in np.unique(xcc):
        m = xcc == x
        t0 = st[m, 0].min()
        rel[m] = (st[m, :4] - t0).astype(np.float64)
    dur = rel[:, 3] - rel[:, 0]
    print("%s %s: %d WGs | prologue med %.0f | loop med %.0f (min %.0f max %.0f) | epilogue med %.0f | WG duration med %.0f | kernel span (max end) %.0f cycles"
          % (name, mode, len(st), np.median(rel[:, 1] - rel[:, 0]), np.median(rel[:, 2] - rel[:, 1]), (rel[:, 2] - rel[:, 1]).min(),
             (rel[:, 2] - rel[:, 1]).max(), np.median(rel[:, 3] - rel[:, 2]), np.median(dur), rel[:, 3].max()))
    late = rel[:, 0] > 0.2 * rel[:, 3].max()
    print("   WGs starting after 20%% of the kernel span: %d ; start-time percentiles 50/75/90/100: %s" % (late.sum(), np.percentile(rel[:, 0], [50, 75, 90, 100]).round()))
    # HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...
    cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    import collections
    cnt = collections.Counter(key.tolist())
    hist = collections.Counter(cnt.values())
    print("   CUs used %d ; WGs-per-CU histogram %s" % (len(cnt), dict(sorted(hist.items()))))


run("3x3 d2 256->256", 10, 60, 60, 256, 256, 3, 2, 2, "fwd")
run("3x3 d2 512->256", 10, 60, 60, 512, 256, 3, 2, 2, "fwd")
run("1x1 1024->256", 10, 60, 60, 1024, 256, 1, 0, 1, "fwd")
