"""F(3x3,3x3) (csrc/winograd_f3.hip) against the shipped F(2x2,3x3) forms (16-GEMM / row-fused / fused operand, whatever
ops dispatches) on the bench workload's stride-1 3x3 shapes: time of all three passes through the SAME ops-level calls the
training step makes (conv2d_forward with BatchNorm statistics, conv2d_backward_data with the BatchNorm-backward front end,
conv2d_backward_weight with the kept V), rotating buffer sets, plus agreement of the results.
Usage: python tools/diag/wino3_probe.py [filter] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cvpr2021_vspw_implement_amd import ops  # noqa: E402
from cvpr2021_vspw_implement_amd import _ops_conv as OC  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # name, n, h, w, cin, cout, dil, launches per step (fwd = dgrad = wgrad count)
    ("l3 256->256 d2", 10, 60, 60, 256, 256, 2, 22),
    ("l3.0 256->256 d1", 10, 60, 60, 256, 256, 1, 1),
    ("l4 512->512 d4", 10, 60, 60, 512, 512, 4, 2),
    ("l4.0 512->512 d2", 10, 60, 60, 512, 512, 2, 1),
    ("deepsup 1024->512 d1", 10, 60, 60, 1024, 512, 1, 1),
    ("conv_last 4096->512 n2", 2, 60, 60, 4096, 512, 1, 1),
    ("l2 128->128 d1", 10, 60, 60, 128, 128, 1, 3),
]
SETS = 3


def timeit(fn, iters):
    for i in range(SETS):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % SETS)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


class Link:
    pass


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    ops.set_wgrad_side_stream(False)
    print("%-24s %-8s %9s %9s %9s %9s   %s" % ("shape", "tile", "fwd us", "dgrad us", "wgrad us", "sum us", "rel diff of (y, dx, dw) vs F(2x2)"))
    saved = {}
    total = {False: 0.0, 3: 0.0, 4: 0.0, 5: 0.0}
    for name, n, h, w, c, k, dil, per_step in SHAPES:
        if flt and flt not in name:
            continue
        xs = [ops.empty_nhwc(n, c, h, w, dev).normal_().relu_() for _ in range(SETS)]
        dys = [ops.empty_nhwc(n, k, h, w, dev).normal_() for _ in range(SETS)]
        zs = [ops.empty_nhwc(n, c, h, w, dev).normal_() for _ in range(SETS)]   # ReLU source / BN y of the producer
        wt = (torch.randn(k, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5).contiguous(memory_format=torch.channels_last)
        mean, invstd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        for f3 in (False, 3, 4, 5):
            ops.set_winograd_f3(f3)
            outs = {}

            def fwd(i):
                y, part, d = OC.conv2d_forward(xs[i], wt, None, 1, dil, dil, want_stats=True, wgrad=True)
                outs["y"], outs["d"] = y, d
                outs["v"] = getattr(y, "_vspw_wino_v", None)
                y._vspw_wino_v = None

            def dgrad(i):
                lk = Link()
                lk.y, lk.mean, lk.invstd = zs[i], mean, invstd
                outs["dx"] = OC.conv2d_backward_data(dys[i], wt, outs["d"], bn_front=(zs[i], lk))

            def wgrad(i):
                outs["dw"] = OC._wgrad_launch(dys[i], xs[i], outs["d"], wino_v=outs["v"])[0]

            tf = timeit(fwd, iters)
            fwd(0)
            td = timeit(dgrad, iters)
            tw = timeit(wgrad, iters)
            fwd(0), dgrad(0), wgrad(0)
            torch.cuda.synchronize()
            res = [outs[q].detach().float().clone() for q in ("y", "dx", "dw")]
            diff = ""
            if f3:
                diff = " ".join("%.2e" % float((a - b).norm() / b.norm()) for a, b in zip(res, saved[name]))
            else:
                saved[name] = res
            total[f3] += per_step * (tf + td + tw)
            print("%-24s %-8s %9.1f %9.1f %9.1f %9.1f   %s" % (name, "F(%dx%d)" % (f3 or 2, f3 or 2), tf, td, tw, tf + td + tw, diff), flush=True)
            outs.clear()
        del xs, dys, zs
        torch.cuda.empty_cache()
    print("per step (launch counts of TCB-PSP R101): F(2x2) %.2f ms, F(3x3) %.2f ms, F(4x4) %.2f ms, F(5x5) %.2f ms" % (total[False] / 1e3, total[3] / 1e3, total[4] / 1e3, total[5] / 1e3))


if __name__ == "__main__":
    main()
