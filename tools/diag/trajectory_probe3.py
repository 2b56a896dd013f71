"""Where do the data gradients of the r18 + ppm_deepsup model diverge between the Winograd and the direct path?
Gradients w.r.t. every block output (retain_grad), relative L2 difference between the two HIP paths, deepest first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cvpr2021_vspw_implement_amd.models as M
from cvpr2021_vspw_implement_amd import ops
from helpers import K, load_det, zero_dropout
from oracle.det_init import det_input, det_labels

dev = torch.device("cuda:0")
tag = "frame_train_trajectory"


def run():
    enc = M.ModelBuilder.build_encoder(arch="resnet18dilated", fc_dim=512)
    dec = M.ModelBuilder.build_decoder(arch="ppm_deepsup", fc_dim=512, num_class=K)
    mod = M.SegmentationModule(enc, dec, torch.nn.NLLLoss(ignore_index=255), 0.4)
    load_det(mod); zero_dropout(mod); mod.to(dev).train()
    kept = {}

    def hook(name):
        def f(m, i, o):
            t = o[0] if isinstance(o, (tuple, list)) else o
            if torch.is_tensor(t) and t.requires_grad:
                t.retain_grad()
                kept[name] = t
        return f

    hs = []
    for name, m in mod.named_modules():
        if name.count(".") <= 3 and name and not name.endswith(("relu", "relu1", "relu2", "relu3")):
            hs.append(m.register_forward_hook(hook(name)))
    img = torch.from_numpy(det_input("%s:img:0" % tag, (2, 3, 65, 65))).to(dev)
    lab = torch.from_numpy(det_labels("%s:lab:0" % tag, (2, 1, 65, 65), K)).to(dev)
    loss, _ = mod({"img_data": img, "seg_label": lab})
    loss.mean().backward()
    ops.join_side_streams(); torch.cuda.synchronize()
    for h in hs:
        h.remove()
    return {k: (v.detach().float().cpu(), None if v.grad is None else v.grad.detach().float().cpu()) for k, v in kept.items()}, \
        {k: p.grad.detach().float().cpu() for k, p in mod.named_parameters()}


a, pa = run()
ops.set_winograd(False)
b, pb = run()
ops.set_winograd(True)
rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-30))
print("%-36s %12s %12s %8s %10s" % ("module output", "value diff", "grad diff", "flips", "elements"))
for k in a:
    if k in b and a[k][1] is not None and b[k][1] is not None and a[k][1].shape == b[k][1].shape:
        flips = int(((a[k][0] > 0) != (b[k][0] > 0)).sum())
        print("%-36s %12.2e %12.2e %8d %10d" % (k, rel(a[k][0], b[k][0]), rel(a[k][1], b[k][1]), flips, a[k][0].numel()))
print("parameters with the largest wino-vs-direct difference:")
for k, v in sorted(((k, rel(pa[k], pb[k])) for k in pa), key=lambda kv: -kv[1])[:12]:
    print("   %-40s %10.2e" % (k, v))
