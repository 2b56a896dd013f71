"""R101 T=5 B=2 239^2 train step: HIP vs oracle fp64, and oracle fp32 vs oracle fp64 (the fp32 noise floor)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import K, build, load_det, zero_dropout
from oracle import np_models as NM, np_ops as O
from oracle.det_init import det_input, det_labels
kind = sys.argv[1] if len(sys.argv) > 1 else "clip_psp"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 239
dev = torch.device("cuda:0")
T, B = 5, 2
mod = build(kind, "resnet101dilated", args={"clip_num": T}); sd = load_det(mod); zero_dropout(mod); mod.to(dev).train()
imgs = [det_input("benchval:%s:%d" % (kind, t), (B, 3, S, S), seed=11) for t in range(T)]
labs = [det_labels("benchval:%s:%d" % (kind, t), (B, 1, S, S), K, seed=11) for t in range(T)]
t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ti = [t_(a) for a in imgs]; tl = [t_(a) for a in labs]
loss, acc = mod({"img_data": ti[-1], "seg_label": tl[-1], "clipimgs_data": ti[:-1], "cliplabels_data": tl[:-1]})
loss.backward()
g = {k: p.grad.detach().double().cpu().numpy() for k, p in mod.named_parameters() if p.grad is not None}
fn = NM.clip_psp if kind == "clip_psp" else NM.clip_ocr
res = {}
for dt in (np.float64, np.float32):
    t0 = time.time(); O.set_dtype(dt)
    P = NM.Params({k: v.astype(dt) for k, v in sd.items()}, train_params=True)
    ol, oa = fn(P, "resnet101", [a.astype(dt) for a in imgs], labs, True)
    O.tape().backward(ol)
    res[dt] = (float(np.asarray(ol.v).reshape(())), oa, {k: v.astype(np.float64) for k, v in P.grads().items()})
    print(dt.__name__, "%.1f s" % (time.time() - t0), "loss", res[dt][0])
O.set_dtype(np.float32)
l64, a64, g64 = res[np.float64]; l32, a32, g32 = res[np.float32]
print("loss hip %.7f or64 %.7f or32 %.7f" % (loss.item(), l64, l32))
norms = {k: np.linalg.norm(v) for k, v in g64.items()}; scale = max(norms.values())
rows = []
for k, r in norms.items():
    rows.append((abs(np.linalg.norm(g[k]) - r) / max(r, 1e-3 * scale), abs(np.linalg.norm(g32[k]) - r) / max(r, 1e-3 * scale), k))
eh = np.array([r[0] for r in rows]); er = np.array([r[1] for r in rows])
print("per-param norm err: hip max %.3e rms %.3e | oracle32 max %.3e rms %.3e" % (eh.max(), np.sqrt((eh**2).mean()), er.max(), np.sqrt((er**2).mean())))
print("   percentiles 50/90/99: hip %s | oracle32 %s" % (np.percentile(eh, [50, 90, 99]).round(5), np.percentile(er, [50, 90, 99]).round(5)))
num = sum((np.linalg.norm(g[k]) - r) ** 2 for k, r in norms.items()); den = sum(r ** 2 for r in norms.values())
num32 = sum((np.linalg.norm(g32[k]) - r) ** 2 for k, r in norms.items())
print("   aggregate norm-vector error: hip %.3e oracle32 %.3e" % ((num / den) ** 0.5, (num32 / den) ** 0.5))
rows.sort(reverse=True)
for r in rows[:10]: print("hip %.3e or32 %.3e %s" % r)
for k in ("encoder.conv1.weight", "encoder.layer3.22.conv2.weight", "encoder.layer4.2.conv3.weight"):
    print(k, "rel L2 hip %.3e or32 %.3e" % (np.linalg.norm(g[k] - g64[k]) / np.linalg.norm(g64[k]), np.linalg.norm(g32[k] - g64[k]) / np.linalg.norm(g64[k])))
