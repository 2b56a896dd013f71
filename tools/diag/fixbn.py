import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import build, clip_inputs, golden, load_det
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "clip_psp"
tag = "r50_%s_fixbn" % kind
fx = golden(tag)
mod = build(kind, "resnet50dilated"); load_det(mod, fx=fx); mod.to(dev); mod.eval()
inp = clip_inputs(tag)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
imgs = [t(a) for a in inp["train_imgs"]]; labs = [t(a) for a in inp["train_labs"]]
loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1], "cliplabels_data": labs[:-1]})
loss.backward()
print("loss", loss.item(), float(fx["train_loss"]), float(fx["train_loss64"]))
g = {k: p.grad.detach().double().cpu().numpy() for k, p in mod.named_parameters() if p.grad is not None}
names = [str(n) for n in fx["grad_names"]]
n64 = dict(zip(names, fx["grad_norms64"])); n32 = dict(zip(names, fx["grad_norms"]))
rows = []
for k in names:
    r = n64[k]; rows.append((abs(np.linalg.norm(g[k]) - r) / max(r, 1e-12), abs(n32[k] - r) / max(r, 1e-12), r, k))
rows.sort(reverse=True)
for e, e32, r, k in rows[:25]: print("%.3e  ref32-vs-64 %.3e  norm %.3e  %s" % (e, e32, r, k))
for key in fx.files:
    if key.startswith("grad64:") and ("layer1.0.bn1" in key or "conv1.weight" in key):
        ref = fx[key].astype(np.float64); ref32 = fx["grad:" + key[7:]].astype(np.float64)
        d = g[key[7:]] - ref
        print(key, "max|ref|", np.abs(ref).max(), "hip err", np.abs(d).max(), "ref32 err", np.abs(ref32 - ref).max(), "argmax", np.unravel_index(np.abs(d).argmax(), d.shape))
# distribution of per-tensor max-abs error ratios hip/ref32 (both against ref64)
ratios = []
for key in fx.files:
    if key.startswith("grad64:"):
        ref = fx[key].astype(np.float64); ref32 = fx["grad:" + key[7:]].astype(np.float64)
        eh = np.abs(g[key[7:]] - ref).max() / np.abs(ref).max(); er = np.abs(ref32 - ref).max() / np.abs(ref).max()
        ratios.append((eh / max(er, 1e-12), eh, er, key))
ratios.sort(reverse=True)
print("tensors", len(ratios), "median ratio", np.median([r[0] for r in ratios]), "max hip rel err", max(r[1] for r in ratios), "max ref32 rel err", max(r[2] for r in ratios))
for r in ratios[:8]: print("ratio %.1f hip %.2e ref32 %.2e %s" % r)
eh = np.array([r[0] for r in rows]); er = np.array([r[1] for r in rows])
print("norm err RMS hip %.3e ref32 %.3e ; max hip %.3e ref32 %.3e" % (np.sqrt((eh**2).mean()), np.sqrt((er**2).mean()), eh.max(), er.max()))
