"""Step-0 gradients of the trajectory fixture: HIP with / without Winograd and the reference's float32, each against the
reference's float64 (64 sampled elements + norm per parameter): which parameters carry the Winograd path's deviation?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cvpr2021_vspw_implement_amd.models as M
from cvpr2021_vspw_implement_amd import ops
from helpers import K, golden, load_det, zero_dropout
from oracle.det_init import det_input, det_labels, det_sample_index

dev = torch.device("cuda:0")
fx = golden("frame_train_trajectory")
tag = "frame_train_trajectory"
names = [str(n) for n in fx["param_names"]]


def grads():
    enc = M.ModelBuilder.build_encoder(arch="resnet18dilated", fc_dim=512)
    dec = M.ModelBuilder.build_decoder(arch="ppm_deepsup", fc_dim=512, num_class=K)
    mod = M.SegmentationModule(enc, dec, torch.nn.NLLLoss(ignore_index=255), 0.4)
    load_det(mod); zero_dropout(mod); mod.to(dev).train()
    img = torch.from_numpy(det_input("%s:img:0" % tag, (2, 3, 65, 65))).to(dev)
    lab = torch.from_numpy(det_labels("%s:lab:0" % tag, (2, 1, 65, 65), K)).to(dev)
    loss, _ = mod({"img_data": img, "seg_label": lab})
    loss.mean().backward()
    ops.join_side_streams(); torch.cuda.synchronize()
    out = {}
    for k, p in mod.named_parameters():
        g = p.grad.detach().double().cpu()  # logical NCHW order
        out[k] = (float(g.norm()), g.contiguous().flatten()[det_sample_index(k, g.numel(), 64)].numpy())
    return out


ref64n, ref64s = fx["f64:grad0_norms"], fx["f64:grad0_samples"]
ref32n, ref32s = fx["f32:grad0_norms"], fx["f32:grad0_samples"]
gw = grads()
ops.set_winograd(False); gd = grads(); ops.set_winograd(True)
rows = []
for i, k in enumerate(names):
    sc = np.abs(ref64s[i]).max() + 1e-30
    rows.append((k, abs(gw[k][0] - ref64n[i]) / ref64n[i], abs(gd[k][0] - ref64n[i]) / ref64n[i], abs(ref32n[i] - ref64n[i]) / ref64n[i],
                 np.abs(gw[k][1] - ref64s[i]).max() / sc, np.abs(gd[k][1] - ref64s[i]).max() / sc, np.abs(ref32s[i] - ref64s[i]).max() / sc))
print("%-40s %10s %10s %10s | %10s %10s %10s" % ("parameter", "norm:wino", "direct", "ref32", "elem:wino", "direct", "ref32"))
for r in rows:
    if r[0].endswith("weight") and ("conv" in r[0] or "downsample.0" in r[0] or "ppm" in r[0] or "cbr" in r[0]):
        print("%-40s %10.2e %10.2e %10.2e | %10.2e %10.2e %10.2e" % r)
a = np.array([[r[1], r[2], r[3], r[4], r[5], r[6]] for r in rows])
print("median", a.__class__.__name__, np.median(a, 0))
print("max   ", a.max(0))
