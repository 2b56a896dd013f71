"""How much of a short-K GEMM launch is epilogue / K loop?  Needs a diagnostic build (VSPW_CXXFLAGS=-DVSPW_NT_DBG):
VSPW_NT_DBG=1 drops the epilogue's memory traffic, =2 the K loop, =3 both (launch + prologue only).
Shapes: the Winograd batch-16 GEMMs and the layer-3 pointwise convolutions of the bench step."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpr2021_vspw_implement_amd import _C, ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, B, M, N, K in (("wino 256", 16, 9000, 256, 256), ("wino 512 d4", 16, 10240, 512, 512), ("pw 256->1024", 1, 36000, 1024, 256),
                         ("pw 1024->256", 1, 36000, 256, 1024), ("wino 128", 16, 9000, 128, 128)):
    a = torch.randn(B, M, K, device=dev); b = torch.randn(B, N, K, device=dev); c = torch.empty(B, M, N, device=dev)
    row = []
    for dbg in (0, 1, 2, 3):
        os.environ["VSPW_NT_DBG"] = str(dbg)
        us = timeit(lambda: _C.call("vspw_bmm_nt", a.data_ptr(), b.data_ptr(), c.data_ptr(), B, M, N, K, st))
        row.append(us)
    os.environ["VSPW_NT_DBG"] = "0"
    gf = 2.0 * B * M * N * K / 1e9
    print("%-14s full %.1f us (%.1f TF) | no-epilogue-traffic %.1f (%.1f TF) | no-K-loop %.1f | neither %.1f" % (name, row[0], gf / row[0] * 1e3 / 1e3, row[1], gf / row[1], row[2], row[3]))
