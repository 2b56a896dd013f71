import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import *
dev = torch.device("cuda:0")
for kind in sys.argv[1:] or ["clip_ocr", "clip_psp"]:
    tag = "r50_" + kind
    fx = golden(tag)
    mod = build(kind, "resnet50dilated"); load_det(mod, fx=fx); zero_dropout(mod); mod.to(dev).train()
    inp = clip_inputs(tag)
    imgs = [torch.from_numpy(a).to(dev) for a in inp["train_imgs"]]
    labs = [torch.from_numpy(a).to(dev) for a in inp["train_labs"]]
    loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1], "cliplabels_data": labs[:-1]})
    loss.backward()
    names = [str(n) for n in fx["grad_names"]]
    ref = dict(zip(names, fx["grad_norms"])); r64 = dict(zip(names, fx["grad_norms64"]))
    g = {k: p.grad.double().norm().item() for k, p in mod.named_parameters() if p.grad is not None}
    scale = max(ref.values())
    rows = sorted(((abs(g[n] - ref[n]) / max(ref[n], 1e-3 * scale), n, g[n], ref[n], r64[n]) for n in names), reverse=True)
    print(kind, "loss", loss.item(), float(fx["train_loss"]))
    for r in rows[:25]:
        print("  %.3e %-55s gpu %.6e ref32 %.6e ref64 %.6e" % r)
    print("  median err %.3e" % np.median([r[0] for r in rows]))
    by = {r[1]: r for r in rows}
    for n in names:
        if not n.startswith("encoder.") or n.startswith("encoder.layer4.2") or n.startswith("encoder.layer3.5."):
            r = by[n]
            print("  %+.3e %-60s gpu %.5e ref32 %.5e" % ((r[2] - r[3]) / max(r[3], 1e-3 * scale), n, r[2], r[3]))
