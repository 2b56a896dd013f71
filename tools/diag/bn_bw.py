"""Streaming bandwidth of the BatchNorm apply passes at the layer-3 shapes (A/B of library builds / VSPW_STREAM_BLOCKS)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpr2021_vspw_implement_amd import _C
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
def timeit(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = []
for rows, c, res in ((36000, 256, False), (36000, 1024, True), (36000, 1024, False), (144000, 256, True)):
    # rotate over 6 buffer sets (> 256 MB in total) so that the Infinity Cache cannot serve the reads
    sets = [(torch.randn(rows, c, device=dev), torch.randn(rows, c, device=dev) if res else None, torch.empty(rows, c, device=dev))
            for _ in range(6)]
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    k = [0]
    def f():
        x, r, z = sets[k[0] % 6]; k[0] += 1
        _C.call("vspw_bn_apply", p(x), p(sc), p(sh), p(r), None, p(z), rows, c, rows, 1, st)
    us = timeit(f)
    nbytes = rows * c * 4 * (3 if res else 2)
    out.append("apply %dx%d%s %.1f us %.2f TB/s" % (rows, c, "+res" if res else "", us, nbytes / us / 1e6))
    del sets
print("blocks=%s | %s" % (os.environ.get("VSPW_STREAM_BLOCKS", "2048"), " | ".join(out)))
