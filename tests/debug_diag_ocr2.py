import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from cvpr2021_vspw_implement_amd.models.ocr_modules.spatial_ocr_block import SpatialOCR_Module
dev = torch.device("cuda:0")
def rel(a, b): return ((a.detach().cpu() - b.detach().cpu()).norm() / max(b.norm().item(), 1e-30)).item()
def run(training, dims, ref_dev="cpu", dtype=torch.float32):
    torch.manual_seed(3); g = torch.Generator().manual_seed(3)
    B, C, Kc, h, w, key = dims
    mod = SpatialOCR_Module(in_channels=C, key_channels=key, out_channels=C, scale=1, dropout=0.0)
    for m in mod.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    mod.train(training)
    x = torch.randn(B, C, h, w, generator=g); proxy = torch.randn(B, C, Kc, 1, generator=g)
    sd = {k: v.clone() for k, v in mod.state_dict().items()}
    def cv(t): return t.to(dtype)
    def bn(t, pre, params):
        return F.relu(F.batch_norm(t, cv(sd[pre + ".running_mean"]).clone(), cv(sd[pre + ".running_var"]).clone(), params[pre + ".weight"], params[pre + ".bias"], training, 0.1, 1e-5))
    def cbr(t, pre, params, i=0):
        t = F.conv2d(t, params["%s.%d.weight" % (pre, i)], params["%s.%d.bias" % (pre, i)])
        return bn(t, "%s.%d" % (pre, i + 1), params)
    params = {k: cv(v).clone().contiguous().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    xr, pr = cv(x).clone().requires_grad_(True), cv(proxy).clone().requires_grad_(True)
    ob = "object_context_block."
    q = cbr(cbr(xr, ob + "f_pixel", params), ob + "f_pixel", params, 3).view(B, key, -1).permute(0, 2, 1)
    kk = cbr(cbr(pr, ob + "f_object", params), ob + "f_object", params, 3).view(B, key, -1)
    v = cbr(pr, ob + "f_down", params).view(B, key, -1).permute(0, 2, 1)
    sim = F.softmax((key ** -0.5) * torch.matmul(q, kk), dim=-1)
    ctx = torch.matmul(sim, v).permute(0, 2, 1).contiguous().view(B, key, h, w)
    ctx = cbr(ctx, ob + "f_up", params)
    ref = cbr(torch.cat([ctx, xr], 1), "conv_bn_dropout", params)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(cv(gy))
    mod.to(dev)
    xd, pd = x.to(dev).requires_grad_(True), proxy.to(dev).requires_grad_(True)
    out = mod(xd, pd); out.backward(gy.to(dev))
    print("training", training, dims, "ref dtype", dtype)
    print("   out %.2e dx %.2e dproxy %.2e" % (rel(out, ref.float()), rel(xd.grad, xr.grad.float()), rel(pd.grad, pr.grad.float())))
    for k, p in mod.named_parameters():
        if params[k].grad.norm() > 1e-6:
            print("   %-50s %.2e" % (k, rel(p.grad, params[k].grad.float())))
run(False, (2, 512, 124, 9, 9, 256))
run(True, (2, 512, 124, 9, 9, 256))
run(True, (2, 512, 124, 9, 9, 256), dtype=torch.float64)
run(True, (2, 512, 124, 24, 24, 256), dtype=torch.float64)
