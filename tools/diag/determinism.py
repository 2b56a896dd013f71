"""Run-to-run determinism of the eager training step and graph-vs-eager equality (diagnostic, GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import build, clip_inputs, golden, load_det, zero_dropout
from cvpr2021_vspw_implement_amd import optim
from cvpr2021_vspw_implement_amd.graph import GraphedStep

dev = torch.device("cuda:0")
def t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

def run(mode, steps=5, lr=0.01):
    fx = golden("r50_clip_psp")
    mod = build("clip_psp", "resnet50dilated"); load_det(mod, fx=fx); zero_dropout(mod); mod.to(dev).train()
    inp = clip_inputs("r50_clip_psp")
    imgs = [t(a) for a in inp["train_imgs"]]; labs = [t(a) for a in inp["train_labs"]]
    opt = optim.create_optimizers(mod, lr=lr, weight_decay=1e-4, momentum=0.9)
    def body():
        mod.zero_grad()
        loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": list(imgs[:-1]), "cliplabels_data": list(labs[:-1])})
        loss.backward(); opt.step(); return loss
    out = []
    if mode == "graph":
        g = GraphedStep(body, warmup=2)
        for _ in range(steps - 2): out.append(g.replay().item())
    elif mode == "side":
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(steps): out.append(body().item())
    else:
        for _ in range(steps): out.append(body().item())
    torch.cuda.synchronize()
    gn = float(sum(p.double().norm() ** 2 for p in mod.parameters()) ** 0.5)
    return out, gn

for mode in ("eager", "eager", "side", "graph", "graph"):
    o, gn = run(mode)
    print(mode, ["%.8f" % v for v in o], "%.10f" % gn)
