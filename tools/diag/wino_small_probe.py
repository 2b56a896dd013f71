"""Winograd vs direct on the small dilated maps of resnet18dilated at 65x65 (9x9 features): forward, data gradient and
weight gradient of plain convolutions against float64 on the CPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from cvpr2021_vspw_implement_amd import ops

dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
for (n, c, k, h, w, d) in [(2, 256, 128, 9, 9, 1), (2, 2560, 512, 9, 9, 1), (2, 256, 512, 9, 9, 1), (2, 256, 256, 9, 9, 2), (2, 512, 512, 9, 9, 4), (2, 256, 256, 9, 9, 1), (2, 128, 128, 9, 9, 1),
                           (2, 256, 512, 9, 9, 2), (2, 512, 512, 9, 9, 2), (2, 256, 256, 8, 8, 2), (2, 256, 256, 10, 10, 2),
                           (2, 512, 512, 8, 8, 4), (2, 512, 512, 12, 12, 4)]:
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, c, h, w, generator=g).relu()
    wt = torch.randn(k, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
    go = torch.randn(n, k, h, w, generator=g)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=d, dilation=d)
    yr.backward(go.double())
    out = []
    for wino in (True, False):
        ops.set_winograd(wino)
        xd = x.to(dev).requires_grad_(True)
        wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        before = ops._wino["launches"]
        yd = ops.conv2d(xd, wd, None, 1, d, d)
        yd.backward(go.to(dev))
        ops.join_side_streams(); torch.cuda.synchronize()
        out.append("%s(%d launches): y %.1e dx %.1e dw %.1e" % ("wino" if wino else "direct", ops._wino["launches"] - before,
                                                                   rel(yd.detach(), yr.detach()), rel(xd.grad, xr.grad), rel(wd.grad, wr.grad)))
    ops.set_winograd(True)
    print("n%d c%d->k%d %dx%d d%d | %s | %s" % (n, c, k, h, w, d, out[0], out[1]))
