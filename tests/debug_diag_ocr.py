import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from cvpr2021_vspw_implement_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rel(a, b): return ((a - b).norm() / b.norm()).item()
for (M, K, N) in [(81, 256, 124), (81, 124, 256), (162, 512, 256), (117, 64, 124), (81, 128, 124), (81, 124, 128), (200, 124, 256), (81, 92, 256)]:
    a = torch.randn(2, M, K, device=dev); bt = torch.randn(2, N, K, device=dev)
    ar, br = a.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    y = ar @ br.transpose(1, 2); gy = torch.randn_like(y); y.backward(gy)
    ad, bd = a.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    yd = ops.bmm_nt(ad, bd); yd.backward(gy)
    print("bmm_nt M%d K%d N%d  fwd %.2e da %.2e db %.2e" % (M, K, N, rel(yd, y), rel(ad.grad, ar.grad), rel(bd.grad, br.grad)))
for (R, M, N) in [(81, 124, 512), (81, 256, 124), (3600, 124, 512)]:
    a = torch.randn(2, R, M, device=dev); b = torch.randn(2, R, N, device=dev)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = ar.transpose(1, 2) @ br; gy = torch.randn_like(y); y.backward(gy)
    ad, bd = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yd = ops.bmm_tn(ad, bd); yd.backward(gy)
    print("bmm_tn R%d M%d N%d  fwd %.2e da %.2e db %.2e" % (R, M, N, rel(yd, y), rel(ad.grad, ar.grad), rel(bd.grad, br.grad)))
for (n, c, h, w, k) in [(2, 512, 9, 9, 256), (2, 512, 124, 1, 256), (2, 256, 9, 9, 512), (2, 1024, 9, 9, 512), (2, 256, 124, 1, 256)]:
    x = torch.randn(n, c, h, w, device=dev); wt = torch.randn(k, c, 1, 1, device=dev) * 0.05; b = torch.randn(k, device=dev)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, br); gy = torch.randn_like(y); y.backward(gy)
    xd, wd, bd = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yd = ops.conv2d(xd, wd, bd, 1, 0, 1); yd.backward(gy)
    print("conv1x1 %s fwd %.2e dx %.2e dw %.2e db %.2e" % ((n, c, h, w, k), rel(yd, y), rel(xd.grad, xr.grad), rel(wd.grad, wr.grad), rel(bd.grad, br.grad)))
