"""Every data-gradient / weight-gradient call of one r18 + ppm_deepsup training step, evaluated by BOTH the Winograd and
the direct path on the SAME operands (the tensors the model really produces), each against float64 on the CPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import cvpr2021_vspw_implement_amd.models as M
from cvpr2021_vspw_implement_amd import _ops_bn, _ops_conv, ops
from helpers import K, load_det, zero_dropout
from oracle.det_init import det_input, det_labels

dev = torch.device("cuda:0")
tag = "frame_train_trajectory"
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))
orig_dgrad, orig_wgrad = _ops_conv.conv2d_backward_data, _ops_conv.conv2d_backward_weight
busy = [False]


def dgrad(dy, w, d, addend=None, bn_front=None, aff=None):
    out = orig_dgrad(dy, w, d, addend=addend, bn_front=bn_front, aff=aff)
    if not busy[0] and d.kh == 3 and d.stride == 1 and aff is None and addend is None and bn_front is None and _ops_conv._wino_ok(d):
        busy[0] = True
        ops.set_winograd(False)
        ref = orig_dgrad(dy, w, d)
        ops.set_winograd(True)
        wd = w.detach().double().cpu()
        r64 = torch.nn.grad.conv2d_input((d.n, d.c, d.h, d.w), wd, dy.detach().double().cpu(), stride=1, padding=d.pad, dilation=d.dil)
        print("dgrad c%d->k%d %dx%d d%d: wino vs f64 %.1e, direct vs f64 %.1e, |dy| %.2e" % (d.c, d.k, d.h, d.w, d.dil, rel(out, r64), rel(ref, r64), float(dy.norm())))
        busy[0] = False
    return out


def wgrad(dy, x, d, aff=None, wino_v=None):
    out = orig_wgrad(dy, x, d, aff=aff, wino_v=wino_v)
    if not busy[0] and d.kh == 3 and d.stride == 1 and aff is None and _ops_conv._wino_ok(d):
        busy[0] = True
        ops.join_side_streams()
        ops.set_winograd(False)
        ref = orig_wgrad(dy, x, d)
        ops.join_side_streams()
        ops.set_winograd(True)
        r64 = torch.nn.grad.conv2d_weight(x.detach().double().cpu(), (d.k, d.c, 3, 3), dy.detach().double().cpu(), stride=1, padding=d.pad, dilation=d.dil)
        fresh = orig_wgrad(dy, x, d)  # Winograd again, V recomputed from x instead of the forward's kept copy
        ops.join_side_streams()
        print("wgrad c%d->k%d %dx%d d%d: wino(kept V) vs f64 %.1e, wino(fresh V) %.1e, direct vs f64 %.1e" % (
            d.c, d.k, d.h, d.w, d.dil, rel(out, r64), rel(fresh, r64), rel(ref, r64)))
        busy[0] = False
    return out


_ops_bn.conv2d_backward_data = dgrad
_ops_bn.conv2d_backward_weight = wgrad
ops.set_wgrad_side_stream(False)
enc = M.ModelBuilder.build_encoder(arch="resnet18dilated", fc_dim=512)
dec = M.ModelBuilder.build_decoder(arch="ppm_deepsup", fc_dim=512, num_class=K)
mod = M.SegmentationModule(enc, dec, torch.nn.NLLLoss(ignore_index=255), 0.4)
load_det(mod); zero_dropout(mod); mod.to(dev).train()
img = torch.from_numpy(det_input("%s:img:0" % tag, (2, 3, 65, 65))).to(dev)
lab = torch.from_numpy(det_labels("%s:lab:0" % tag, (2, 1, 65, 65), K)).to(dev)
loss, _ = mod({"img_data": img, "seg_label": lab})
loss.mean().backward()
torch.cuda.synchronize()
