"""OCR building blocks on the HIP kernels vs a plain PyTorch CPU evaluation of the reference formulas
(models/ocr_modules/spatial_ocr_block.py:82-129, 247-289, 358-381), forward and backward."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol, what):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = max(b.abs().max().item(), 1.0)
    assert err <= tol * ref, "%s: max abs err %.3e (ref max %.3e)" % (what, err, ref)


def _ref_gather(feats, probs, T):
    n, k, h, w = probs.shape
    B = n // T
    ctxs = []
    for pf, ff in zip(torch.split(probs, B, 0), torch.split(feats, B, 0)):
        p = F.softmax(pf.view(B, k, -1), dim=2)
        f = ff.view(B, ff.size(1), -1).permute(0, 2, 1)
        ctxs.append(torch.matmul(p, f).permute(0, 2, 1).unsqueeze(3).unsqueeze(0))
    return torch.mean(torch.cat(ctxs, 0), 0)


def test_temporal_gather(dev):
    from cvpr2021_vspw_implement_amd.models.ocr_modules.spatial_ocr_block import SpatialTemporalGather_Module

    g = torch.Generator().manual_seed(1)
    T, B, C, Kc, h, w = 3, 2, 32, 7, 9, 13
    feats = torch.randn(T * B, C, h, w, generator=g)
    probs = torch.randn(T * B, Kc, h, w, generator=g) * 2
    fr, pr = feats.clone().requires_grad_(True), probs.clone().requires_grad_(True)
    ref = _ref_gather(fr, pr, T)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    fd, pd = feats.to(dev).requires_grad_(True), probs.to(dev).requires_grad_(True)
    out = SpatialTemporalGather_Module(Kc)(fd, pd, T - 1)
    out.backward(gy.to(dev))
    _close(out, ref, 1e-5, "gather fwd")
    _close(fd.grad, fr.grad, 1e-5, "gather dfeats")
    _close(pd.grad, pr.grad, 1e-5, "gather dprobs")


def test_channel_cat(dev):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(2)
    a = torch.randn(2, 12, 5, 7, generator=g)
    b = torch.randn(2, 20, 5, 7, generator=g)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = torch.cat([ar, br], 1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    yd = ops.channel_cat([ad, bd])
    yd.backward(gy.to(dev))
    _close(yd, y, 0.0, "cat")
    _close(ad.grad, ar.grad, 0.0, "cat da")
    _close(bd.grad, br.grad, 0.0, "cat db")


@pytest.mark.parametrize("training,dims", [(False, (2, 64, 7, 9, 11, 32)), (True, (2, 64, 7, 9, 11, 32)),
                                           (True, (2, 512, 124, 9, 9, 256))])
def test_spatial_ocr_module(dev, training, dims):
    """Whole SpatialOCR_Module (attention + conv_bn_dropout) against the reference formulas written with torch ops."""
    from cvpr2021_vspw_implement_amd.models.ocr_modules.spatial_ocr_block import SpatialOCR_Module

    torch.manual_seed(3)
    g = torch.Generator().manual_seed(3)
    B, C, Kc, h, w, key = dims
    mod = SpatialOCR_Module(in_channels=C, key_channels=key, out_channels=C, scale=1, dropout=0.0)
    for m in mod.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    mod.train(training)
    x = torch.randn(B, C, h, w, generator=g)
    proxy = torch.randn(B, C, Kc, 1, generator=g)
    sd = {k: v.clone() for k, v in mod.state_dict().items()}

    def cbr(t, pre, params):
        t = F.conv2d(t, params[pre + ".0.weight"], params[pre + ".0.bias"])
        return F.relu(F.batch_norm(t, sd[pre + ".1.running_mean"].clone(), sd[pre + ".1.running_var"].clone(),
                                   params[pre + ".1.weight"], params[pre + ".1.bias"], training, 0.1, 1e-5))

    def cbr2(t, pre, params):
        t = cbr(t, pre, params)
        t = F.conv2d(t, params[pre + ".3.weight"], params[pre + ".3.bias"])
        return F.relu(F.batch_norm(t, sd[pre + ".4.running_mean"].clone(), sd[pre + ".4.running_var"].clone(),
                                   params[pre + ".4.weight"], params[pre + ".4.bias"], training, 0.1, 1e-5))

    params = {k: v.clone().contiguous().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()
              and "running" not in k}
    xr, pr = x.clone().requires_grad_(True), proxy.clone().requires_grad_(True)
    ob = "object_context_block."
    q = cbr2(xr, ob + "f_pixel", params).view(B, key, -1).permute(0, 2, 1)
    kk = cbr2(pr, ob + "f_object", params).view(B, key, -1)
    v = cbr(pr, ob + "f_down", params).view(B, key, -1).permute(0, 2, 1)
    sim = F.softmax((key ** -0.5) * torch.matmul(q, kk), dim=-1)
    ctx = torch.matmul(sim, v).permute(0, 2, 1).contiguous().view(B, key, h, w)
    ctx = cbr(ctx, ob + "f_up", params)
    ref = cbr(torch.cat([ctx, xr], 1), "conv_bn_dropout", params)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)

    mod.to(dev)
    xd, pd = x.to(dev).requires_grad_(True), proxy.to(dev).requires_grad_(True)
    out = mod(xd, pd)
    out.backward(gy.to(dev))
    # eval mode is exact to fp32 rounding; in train mode the five stacked batch-stat BNs over 162-sample populations
    # amplify fp32 rounding to ~1e-5 in the output, and one flipped ReLU decision moves gradients by ~3e-3 (relative
    # norm) — see tools/diag_ocr2.py: the same error appears against a float64 reference
    gtol = 2e-2 if training else 1e-5
    _close(out, ref, 1e-4 if training else 2e-5, "ocr fwd")

    def rel(a, b):
        return ((a.detach().float().cpu() - b).norm() / b.norm()).item()

    assert rel(xd.grad, xr.grad) < gtol and rel(pd.grad, pr.grad) < gtol
    bad = []
    gmax = max(v.grad.norm().item() for v in params.values())
    for k, p in mod.named_parameters():
        b = params[k].grad
        if b.norm().item() < 1e-4 * gmax:  # conv biases in front of a train-mode BN: exact gradient is zero
            continue
        if rel(p.grad, b) > gtol:
            bad.append((k, rel(p.grad, b)))
    assert not bad, bad
