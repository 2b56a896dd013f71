"""DIAGNOSTIC (library built with -DVSPW_NT_TIMING, selected through VSPW_HIP_LIB): are the workgroups of one NT launch
in lockstep?  Prints, over the kernel's span on one XCD, how many workgroups are in prologue / K loop / epilogue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cvpr2021_vspw_implement_amd import _C, ops
dev = torch.device("cuda:0")
lib = _C.load()
ops.set_wgrad_side_stream(False)


def run(name, n, h, w, c, k, ks, pad, dil, mode):
    x = ops.empty_nhwc(n, c, h, w, dev).normal_()
    wt = (torch.randn(k, ks, ks, c, device=dev) * 0.05).permute(0, 3, 1, 2)
    y, part, d = ops.conv2d_forward(x, wt, None, 1, pad, dil, want_stats=True)
    dy = torch.randn_like(y)
    add = torch.randn_like(x)
    for _ in range(3):
        if mode == "fwd":
            ops.conv2d_forward(x, wt, None, 1, pad, dil, want_stats=True)
        elif mode == "dgrad":
            ops.conv2d_backward_data(dy, wt, d)
        else:
            ops.conv2d_backward_data(dy, wt, d, addend=add)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 5, dtype=np.uint64)
    lib.vspw_debug_nt_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.vspw_debug_nt_stamps(buf.ctypes.data, buf.size)
    st = buf.reshape(-1, 5)
    st = st[st[:, 0] > 0]
    xcc = (st[:, 4] >> np.uint64(32)).astype(np.int64)
    hwid = (st[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    m = xcc == 0
    t = st[m, :4].astype(np.int64)
    keep = np.abs(t[:, 0] - np.median(t[:, 0])) < 2000000  # drop stale rows of earlier launches
    t, hwid_m = t[keep], hwid[m][keep]
    t -= t[:, 0].min()
    span = t[:, 3].max()
    print("%s %s: XCD0 %d WGs, span %d cycles; phase medians: prologue %d, loop %d, epilogue %d" % (
        name, mode, len(t), span, np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2])))
    grid = np.linspace(0, span, 41)[:-1]
    rows = []
    for g in grid:
        pro = ((t[:, 0] <= g) & (g < t[:, 1])).sum()
        loop = ((t[:, 1] <= g) & (g < t[:, 2])).sum()
        epi = ((t[:, 2] <= g) & (g < t[:, 3])).sum()
        rows.append((pro, loop, epi))
    print("   in-prologue:", " ".join("%3d" % r[0] for r in rows))
    print("   in-loop    :", " ".join("%3d" % r[1] for r in rows))
    print("   in-epilogue:", " ".join("%3d" % r[2] for r in rows))
    # one CU: the sequence of workgroups it ran
    cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
    key = (se * 100 + sh * 20 + cu)[m][keep]
    k0 = key[0]
    seq = t[key == k0]
    seq = seq[np.argsort(seq[:, 0])]
    print("   one CU (%d WGs): start/loop/epi/end" % len(seq))
    for r in seq[:14]:
        print("      %7d %7d %7d %7d" % tuple(r))


run("1x1 c1024 k256", 10, 60, 60, 1024, 256, 1, 0, 1, "dgrad")
run("1x1 c1024 k256", 10, 60, 60, 1024, 256, 1, 0, 1, "dgrad+add")
run("1x1 c256 k1024", 10, 60, 60, 256, 1024, 1, 0, 1, "fwd")


def run_tn(name, n, hw, c, k, ks, pad, dil):
    x = ops.empty_nhwc(n, c, hw, hw, dev).normal_()
    wt = (torch.randn(k, ks, ks, c, device=dev) * 0.05).permute(0, 3, 1, 2)
    y, part, d = ops.conv2d_forward(x, wt, None, 1, pad, dil, want_stats=False)
    dy = torch.randn_like(y)
    for _ in range(3):
        ops.conv2d_backward_weight(dy, x, d)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 5, dtype=np.uint64)
    lib.vspw_debug_nt_stamps(buf.ctypes.data, buf.size)
    st = buf.reshape(-1, 5)
    st = st[st[:, 0] > 0]
    xcc = (st[:, 4] >> np.uint64(32)).astype(np.int64)
    hwid = (st[:, 4] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    m = xcc == xcc[0]
    t = st[m, :4].astype(np.int64)
    keep = np.abs(t[:, 0] - np.median(t[:, 0])) < 4000000
    t = t[keep]
    t -= t[:, 0].min()
    print("TN %s: %d WGs on one XCD; medians: prologue %d, loop %d, epilogue %d; span %d" % (
        name, len(t), np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]),
        t[:, 3].max()))
    cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
    key = (se * 100 + sh * 20 + cu)[m][keep]
    seq = t[key == key[0]]
    seq = seq[np.argsort(seq[:, 0])]
    for r in seq[:8]:
        print("      %7d %7d %7d %7d" % tuple(r))


run_tn("3x3 256 d2", 10, 60, 256, 256, 3, 2, 2)
run_tn("1x1 256->1024", 10, 60, 256, 1024, 1, 0, 1)
run_tn("1x1 1024->256", 10, 60, 1024, 256, 1, 0, 1)
