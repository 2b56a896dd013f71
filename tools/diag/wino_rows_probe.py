"""Row-fused Winograd GEMM (csrc/wino_rows.hip) against the 16-GEMM form (csrc/winograd.hip) on the bench workload's
3x3 shapes: results (both must agree to rounding) and time of  [input transform] + GEMM + output transform, through the
C ABI.  Buffers rotate over several sets so that no launch finds its operands in the caches by accident.
Usage: python tools/diag/wino_rows_probe.py [filter] [iters]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cvpr2021_vspw_implement_amd import _C, ops  # noqa: E402
from cvpr2021_vspw_implement_amd._C import ConvDesc  # noqa: E402

dev = torch.device("cuda:0")
_p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
SHAPES = [  # name, n, h, w, cin, cout, dil
    ("l3 256->256 d2", 10, 60, 60, 256, 256, 2),
    ("l4 512->512 d4", 10, 60, 60, 512, 512, 4),
    ("l4.0 512->512 d2", 10, 60, 60, 512, 512, 2),
    ("deepsup 1024->512 d1", 10, 60, 60, 1024, 512, 1),
    ("conv_last 4096->512 n2", 2, 60, 60, 4096, 512, 1),
    ("l2 128->128 d1", 10, 60, 60, 128, 128, 1),
    ("ragged 256->128 d2 57x43", 3, 57, 43, 256, 128, 2),
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


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("%-26s %-22s %8s %8s %8s %8s  %s" % ("shape", "variant", "input", "gemm", "output", "total", "max|d| / max|y|"))
    for name, n, h, w, c, k, dil in SHAPES:
        if flt and flt not in name:
            continue
        d = ConvDesc(n, h, w, c, h, w, k, 3, 3, 1, dil, dil, dil)
        T = int(_C.query("vspw_wino_tiles", ctypes.byref(d)))
        xs = [ops.empty_nhwc(n, c, h, w, dev).normal_() for _ in range(SETS)]
        wt = torch.randn(k, c, 3, 3, device=dev).contiguous(memory_format=torch.channels_last) * (2.0 / (9 * c)) ** 0.5
        u = torch.empty(16, k, c, device=dev)
        _C.call("vspw_wino_weights", _p(wt), _p(u), k, c, 0, st)
        vs = [torch.empty(16, T, c, device=dev) for _ in range(SETS)]
        ms = [torch.empty(16, T, k, device=dev) for _ in range(SETS)]
        ys = [ops.empty_nhwc(n, k, h, w, dev) for _ in range(SETS)]
        part = torch.empty(int(_C.query("vspw_wino_stat_partials", ctypes.byref(d))), 2, k, device=dev)
        bias = torch.randn(k, device=dev)
        flops = 2.0 * 16 * T * k * c

        def t_input(i):
            _C.call("vspw_wino_input", ctypes.byref(d), _p(xs[i]), c, _p(vs[i]), st)

        def old_gemm(i):
            _C.call("vspw_bmm_nt", _p(vs[i]), _p(u), _p(ms[i]), 16, T, k, c, st)

        def old_gemm_f(i):
            _C.call("vspw_wino_gemm_fused", ctypes.byref(d), _p(xs[i]), c, _p(u), k, _p(ms[i]), st)

        def old_out(i):
            _C.call("vspw_wino_output", ctypes.byref(d), _p(ms[i]), k, _p(bias), _p(ys[i]), None, None, None, None,
                    _p(part), None, 0, st)

        for i in range(SETS):
            t_input(i)
        ti = timeit(t_input, iters)
        tg = timeit(old_gemm, iters)
        to = timeit(old_out, iters)
        old_out(0)
        torch.cuda.synchronize()
        y_ref = ys[0].clone()
        part_ref = part.clone()
        print("%-26s %-22s %8.1f %8.1f %8.1f %8.1f  (%.1f TFLOP/s)" % (name, "16 GEMMs", ti, tg, to, ti + tg + to,
                                                                      flops / tg * 1e-6))
        fused_ok = k <= 512
        if fused_ok:
            tgf = timeit(old_gemm_f, iters)
            old_out(0)
            torch.cuda.synchronize()
            e = (ys[0] - y_ref).abs().max().item()
            print("%-26s %-22s %8s %8.1f %8.1f %8.1f  %.2e" % ("", "16 GEMMs, fused V", "-", tgf, to, tgf + to, e))
        for tile in (12, 31, 22):
            _C.call("vspw_wino_rows_config", tile)
            tpad = int(_C.query("vspw_wino_rows_tpad", ctypes.byref(d), c, k, 0))
            if tpad == 0:
                print("%-26s rows tile %d: not supported" % ("", tile))
                continue
            tps = [torch.empty(8, tpad, k, device=dev) for _ in range(SETS)]

            def new_gemm(i):
                _C.call("vspw_wino_gemm_rows", ctypes.byref(d), _p(vs[i]), c, _p(u), k, _p(tps[i]), st)

            def new_gemm_f(i):
                _C.call("vspw_wino_gemm_fused_rows", ctypes.byref(d), _p(xs[i]), c, _p(u), k, _p(tps[i]), st)

            def new_out(i):
                _C.call("vspw_wino_output_rows", ctypes.byref(d), _p(tps[i]), tpad, k, _p(bias), _p(ys[i]), None, None,
                        None, None, _p(part), None, 0, st)

            ys[0].zero_()
            tg2 = timeit(new_gemm, iters)
            to2 = timeit(new_out, iters)
            new_gemm(0)
            new_out(0)
            torch.cuda.synchronize()
            e = (ys[0] - y_ref).abs().max().item()
            pe = (part - part_ref).abs().max().item() / part_ref.abs().max().item()
            print("%-26s %-22s %8.1f %8.1f %8.1f %8.1f  %.2e / %.2e  stats rel %.1e (%.1f TFLOP/s)" % (
                "", "rows tile %d" % tile, ti, tg2, to2, ti + tg2 + to2, e, y_ref.abs().max().item(), pe,
                flops / tg2 * 1e-6))
            if fused_ok and tile != 22:
                ys[0].zero_()
                tg3 = timeit(new_gemm_f, iters)
                new_gemm_f(0)
                new_out(0)
                torch.cuda.synchronize()
                e = (ys[0] - y_ref).abs().max().item()
                print("%-26s %-22s %8s %8.1f %8.1f %8.1f  %.2e" % ("", "rows tile %d, fused V" % tile, "-", tg3, to2,
                                                                  tg3 + to2, e))
        for tile, what in ((131, "tile 31, no stores"), (231, "tile 31, no K loop")):  # where the launch's time goes
            _C.call("vspw_wino_rows_config", tile)
            tpad = int(_C.query("vspw_wino_rows_tpad", ctypes.byref(d), c, k, 0))
            if tpad:
                tps = [torch.empty(8, tpad, k, device=dev) for _ in range(SETS)]
                t = timeit(lambda i: _C.call("vspw_wino_gemm_rows", ctypes.byref(d), _p(vs[i]), c, _p(u), k, _p(tps[i]), st),
                           iters)
                print("%-26s %-22s %8s %8.1f" % ("", what, "", t))
        _C.call("vspw_wino_rows_config", 0)


if __name__ == "__main__":
    main()
