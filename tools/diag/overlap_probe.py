"""Can an HBM-bound pass of one half-batch hide under the MFMA-bound GEMM of the other half?  Micro-experiment for the
"two half-batches in two graph branches" idea: layer3's 256 -> 1024 pointwise GEMM + BatchNorm apply,
(a) on the full 36 000-pixel batch in one stream, (b) as two 18 000-pixel halves on two streams, staggered with events so
that apply(A) runs beside GEMM(B) and apply(B) beside the next GEMM(A)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cvpr2021_vspw_implement_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
_p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
M, K, N = 36000, 256, 1024


def bufs(m):
    return (torch.randn(m, K, device=dev), torch.empty(m, N, device=dev), torch.empty(m, N, device=dev))


w = torch.randn(N, K, device=dev)
sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)


def gemm(x, y, stream):
    _C.call("vspw_bmm_nt", _p(x), _p(w), _p(y), 1, x.shape[0], N, K, ctypes.c_void_p(stream.cuda_stream))


def apply(y, z, stream):
    _C.call("vspw_bn_apply", _p(y), _p(sc), _p(sh), None, None, _p(z), y.shape[0], N, y.shape[0], 1,
            ctypes.c_void_p(stream.cuda_stream))


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


s0 = torch.cuda.current_stream()
xf, yf, zf = bufs(M)
t_gemm = timeit(lambda: gemm(xf, yf, s0))
t_apply = timeit(lambda: apply(yf, zf, s0))
t_full = timeit(lambda: (gemm(xf, yf, s0), apply(yf, zf, s0)))
xa, ya, za = bufs(M // 2)
xb, yb, zb = bufs(M // 2)
t_half_gemm = timeit(lambda: gemm(xa, ya, s0))
t_halves_serial = timeit(lambda: (gemm(xa, ya, s0), apply(ya, za, s0), gemm(xb, yb, s0), apply(yb, zb, s0)))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()


def staggered():
    # s1: GEMM(A) [after the previous GEMM(B)] -> apply(A);  s2: GEMM(B) [after GEMM(A)] -> apply(B)
    s1.wait_event(ev_b)
    gemm(xa, ya, s1)
    ev_a.record(s1)
    apply(ya, za, s1)
    s2.wait_event(ev_a)
    gemm(xb, yb, s2)
    ev_b.record(s2)
    apply(yb, zb, s2)


ev_b.record(s2)
s0.wait_stream(s1); s0.wait_stream(s2)
torch.cuda.synchronize()


def run_staggered():
    staggered()


for _ in range(5):
    staggered()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s0)
s1.wait_event(e0); s2.wait_event(e0)
for _ in range(40):
    staggered()
s0.wait_stream(s1); s0.wait_stream(s2)
e1.record(s0)
torch.cuda.synchronize()
t_stag = e0.elapsed_time(e1) / 40 * 1e3
print("GEMM 36000x256->1024: %.1f us; apply: %.1f us; one stream, full batch: %.1f us" % (t_gemm, t_apply, t_full))
print("half GEMM: %.1f us; two halves, one stream: %.1f us; two halves, two streams staggered: %.1f us" % (
    t_half_gemm, t_halves_serial, t_stag))
