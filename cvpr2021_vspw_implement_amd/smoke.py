"""One small invocation of the hot path on cuda:0 (TCB-PSP, ResNet-50, T=3, 65x65 crops: forward, loss, backward),
checked against the golden vector produced by the reference and against the numpy oracle evaluated live."""
import os
import sys

import numpy as np
import torch


def run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import build, clip_inputs, golden, load_det, zero_dropout  # test plumbing (weights / inputs by seed)
    from oracle import np_models as NM  # the checker
    from oracle import np_ops as O

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs cuda:0 (MI355X)")
    dev = torch.device("cuda:0")
    tag = "r50_clip_psp"
    fx = golden(tag)
    mod = build("clip_psp", "resnet50dilated")
    sd = load_det(mod, fx=fx)
    zero_dropout(mod)
    mod.to(dev).train()
    inp = clip_inputs(tag)
    imgs = [torch.from_numpy(a).to(dev) for a in inp["train_imgs"]]
    labs = [torch.from_numpy(a).to(dev) for a in inp["train_labs"]]
    loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1],
                     "cliplabels_data": labs[:-1]})
    loss.backward()
    torch.cuda.synchronize()
    ref = float(fx["train_loss"])
    assert abs(loss.item() - ref) < 2e-4 * abs(ref), ("loss vs reference fixture", loss.item(), ref)
    O.set_dtype(np.float32)
    P = NM.Params({k: v.copy() for k, v in sd.items()}, train_params=False)
    oloss, oacc = NM.clip_psp(P, "resnet50", inp["train_imgs"], inp["train_labs"], True)
    ol = float(oloss.v.reshape(()))
    assert abs(loss.item() - ol) < 2e-4 * abs(ol), ("loss vs oracle", loss.item(), ol)
    gn = mod.encoder.conv1.weight.grad.norm().item()
    names = [str(n) for n in fx["grad_names"]]
    rn = float(fx["grad_norms"][names.index("encoder.conv1.weight")])
    assert abs(gn - rn) < 5e-2 * rn, ("grad norm", gn, rn)
    print("smoke OK: loss %.6f (reference %.6f, oracle %.6f) acc %.4f |dW conv1| %.4f (reference %.4f)"
          % (loss.item(), ref, ol, acc.item(), gn, rn))
