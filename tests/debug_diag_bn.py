import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from cvpr2021_vspw_implement_amd import ops, _C
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (n, c, h, w, k, ks) in [(2, 512, 9, 9, 256, 1), (10, 256, 60, 60, 256, 3), (2, 512, 124, 1, 256, 1)]:
    x = torch.randn(n, c, h, w, device=dev) + 0.5
    wt = (torch.randn(k, c, ks, ks, device=dev) * (2.0 / (c * ks * ks)) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(k, device=dev)
    y, part, d = ops.conv2d_forward(x, wt, b, 1, ks // 2, 1, want_stats=True)
    rows = y.shape[0] * y.shape[2] * y.shape[3]
    yd = y.double()
    mean64 = yd.mean((0, 2, 3)); var64 = yd.var((0, 2, 3), unbiased=False)
    sums_f = torch.empty((2, k), device=dev, dtype=torch.float64)
    _C.call("vspw_bn_reduce_partials_f32", ops._p(part), part.shape[0], k, ops._p(sums_f), ops._stream())
    nb = _C.query("vspw_bn_stats_workspace", rows, k); ws = ops._ws(nb, dev)
    sums_d = torch.empty((2, k), device=dev, dtype=torch.float64)
    _C.call("vspw_bn_stats", ops._p(y), rows, k, ops._p(sums_d), ops._p(ws), nb, ops._stream())
    torch.cuda.synchronize()
    for nm, s in (("fused-epilogue", sums_f), ("bn_stats(f64)", sums_d)):
        m = s[0] / rows; v = s[1] / rows - m * m
        print("%s rows %d k %d | %-15s mean err %.2e  var relerr %.2e  |mean|/std max %.1f" % ((n, c, h, w), rows, k, nm, (m - mean64).abs().max().item(), ((v - var64).abs() / var64).max().item(), (mean64.abs() / var64.sqrt()).max().item()))
    # full BN train output vs fp64
    gamma = torch.rand(k, device=dev) + 0.5; beta = torch.randn(k, device=dev)
    ref = F.relu(F.batch_norm(yd, None, None, gamma.double(), beta.double(), True, 0.1, 1e-5))
    for fused in (True, False):
        z = ops.batch_norm_act(y, gamma, beta, torch.zeros(k, device=dev), torch.ones(k, device=dev), None, None, True, 0.1, 1e-5, True, part if fused else None)
        print("     bn out (fused=%s) max abs err %.2e  rel %.2e" % (fused, (z.double() - ref).abs().max().item(), ((z.double() - ref).norm() / ref.norm()).item()))
    zc = F.relu(F.batch_norm(y.cpu(), None, None, gamma.cpu(), beta.cpu(), True, 0.1, 1e-5))
    print("     torch CPU fp32 bn out max abs err %.2e rel %.2e" % ((zc.double() - ref.cpu()).abs().max().item(), ((zc.double() - ref.cpu()).norm() / ref.cpu().norm()).item()))
