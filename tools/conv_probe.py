"""Per-shape timing of the conv GEMM kernels on the bench workload's own shapes (TCB-PSP R101, N = 10 frames at 60x60):
forward, data gradient (plain / +skip addend / +fused BN-backward front end) and weight gradient, each through the C ABI
exactly as ops.py calls it.  Usage: python tools/conv_probe.py [filter-substring] [iters]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cvpr2021_vspw_implement_amd import _C, ops  # noqa: E402

dev = torch.device("cuda:0")
# name, n, h, w, cin, cout, k, stride, pad, dil
SHAPES = [
    ("l3.conv2 3x3 d2 256->256", 10, 60, 60, 256, 256, 3, 1, 2, 2),
    ("l3.conv1 1x1 1024->256", 10, 60, 60, 1024, 256, 1, 1, 0, 1),
    ("l3.conv3 1x1 256->1024", 10, 60, 60, 256, 1024, 1, 1, 0, 1),
    ("l4.conv2 3x3 d4 512->512", 10, 60, 60, 512, 512, 3, 1, 4, 4),
    ("l4.conv1 1x1 2048->512", 10, 60, 60, 2048, 512, 1, 1, 0, 1),
    ("l4.conv3 1x1 512->2048", 10, 60, 60, 512, 2048, 1, 1, 0, 1),
    ("deepsup 3x3 1024->512", 10, 60, 60, 1024, 512, 3, 1, 1, 1),
    ("conv_last 3x3 4096->512 n2", 2, 60, 60, 4096, 512, 3, 1, 1, 1),
    ("l2.conv2 3x3 128->128", 10, 60, 60, 128, 128, 3, 1, 1, 1),
    ("l2.conv3 1x1 128->512", 10, 60, 60, 128, 512, 1, 1, 0, 1),
    ("l2.conv1 1x1 512->128", 10, 60, 60, 512, 128, 1, 1, 0, 1),
    ("l1.conv2 3x3 64->64 120", 10, 120, 120, 64, 64, 3, 1, 1, 1),
    ("stem.conv3 3x3 64->128 240", 10, 240, 240, 64, 128, 3, 1, 1, 1),
    ("xK 3x3 d2 128->256", 10, 60, 60, 128, 256, 3, 1, 2, 2),
    ("xK 3x3 d2 512->256", 10, 60, 60, 512, 256, 3, 1, 2, 2),
    ("xK 3x3 d2 1024->256", 10, 60, 60, 1024, 256, 3, 1, 2, 2),
    ("xK 1x1 512->256", 10, 60, 60, 512, 256, 1, 1, 0, 1),
    ("xK 1x1 2048->256", 10, 60, 60, 2048, 256, 1, 1, 0, 1),
    ("xK 1x1 4096->256", 10, 60, 60, 4096, 256, 1, 1, 0, 1),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ops.set_wgrad_side_stream(False)
    print("%-30s %9s %9s %9s %9s %9s   (TFLOP/s; ms)" % ("shape", "fwd+stat", "dgrad", "dgrad+add", "dgrad+bn", "wgrad"))
    for name, n, h, w, c, k, ks, st, pad, dil in SHAPES:
        if flt and flt not in name:
            continue
        x = ops.empty_nhwc(n, c, h, w, dev).normal_()
        wt = (torch.randn(k, ks, ks, c, device=dev) * 0.05).permute(0, 3, 1, 2)
        y, part, d = ops.conv2d_forward(x, wt, None, st, pad, dil, want_stats=True)
        fl = 2.0 * n * d.oh * d.ow * k * ks * ks * c
        dy = torch.randn_like(y)
        add = torch.randn_like(x)
        z = torch.randn_like(x)
        link = ops.BNLink()
        link.y, link.mean, link.invstd = torch.randn_like(x), torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5
        res = []
        res.append(timeit(lambda: ops.conv2d_forward(x, wt, None, st, pad, dil, want_stats=True), iters))
        res.append(timeit(lambda: ops.conv2d_backward_data(dy, wt, d), iters))
        res.append(timeit(lambda: ops.conv2d_backward_data(dy, wt, d, addend=add), iters))
        if _C.query("vspw_conv2d_bwd_data_bn_partials", ctypes.byref(d)) > 0:
            res.append(timeit(lambda: ops.conv2d_backward_data(dy, wt, d, addend=add, bn_front=(z, link)), iters))
        else:
            res.append(float("nan"))
        res.append(timeit(lambda: ops.conv2d_backward_weight(dy, x, d), iters))
        print("%-30s " % name + " ".join("%5.1f/%5.3f" % (fl / ms / 1e9, ms) for ms in res))


if __name__ == "__main__":
    main()
