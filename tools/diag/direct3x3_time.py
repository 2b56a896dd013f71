"""Time the direct 3x3 convolutions of the stem / layer1 / layer2.0 (forward and data gradient, the shapes of the bench
step) - A/B of VSPW_DIRECT_FOLD (folded chains, csrc/conv_igemm.hip FOLD)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpr2021_vspw_implement_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = []
for name, n, c, h, k, s in (("stem 64->64 240^2", 10, 64, 240, 64, 1), ("stem 64->128 240^2", 10, 64, 240, 128, 1),
                            ("layer1 64->64 120^2", 10, 64, 120, 64, 1), ("layer2.0 128->128 s2", 10, 128, 120, 128, 2)):
    x = torch.randn(n, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(k, c, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y, _, d = ops.conv2d_forward(x, w, None, s, 1, 1)
    gy = torch.randn_like(y)
    fl = 2.0 * y.numel() * c * 9
    t_f = timeit(lambda: ops.conv2d_forward(x, w, None, s, 1, 1))
    t_b = timeit(lambda: ops.conv2d_backward_data(gy, w, d))
    out.append("%s fwd %.0f us %.0f TF, dgrad %.0f us %.0f TF" % (name, t_f, fl / t_f / 1e6, t_b, fl / t_b / 1e6))
print("DIRECT_FOLD=%s | " % os.environ.get("VSPW_DIRECT_FOLD", "1") + " | ".join(out))
