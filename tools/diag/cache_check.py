"""After a --hip_graph TCB-PSP run, is every cached Winograd weight transform of a NEW model what a fresh transform gives?"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_drivers_gpu as TD
from cvpr2021_vspw_implement_amd import ops, _C
from cvpr2021_vspw_implement_amd import _ops_conv as OC
from cvpr2021_vspw_implement_amd._opbase import _p, _stream
dev = torch.device("cuda:0")
if "nograph" not in sys.argv:
    TD.test_tcb_training_trajectory_follows_the_reference(dev, pathlib.Path(tempfile.mkdtemp()), "clip_psp", True)
for m in (3, 4):
    c = OC._wu3_copies[m]
    print("cache F%d: %d entries, table %s (n %d), alive %d" % (m, len(c.entries), c.table is not None, c.table_n, sum(e["ref"]() is not None for e in c.entries.values())))
import cvpr2021_vspw_implement_amd.train_clip2 as T
from helpers import K, load_det, zero_dropout
from oracle.det_init import damp_residual_gammas, det_input, det_labels
args = T.build_parser().parse_args(["--method", "clip_ocr", "--dataroot", "/tmp", "--saveroot", "/tmp/t", "--batchsize", "2", "--cropsize", "97", "--clip_num", "3", "--dilation2", "3,6", "--totalepoch", "2", "--lr", "0.004", "--workers", "0", "--gpus", "0"])
cfg = TD._cfg("ppm_deepsup_clip")
args.cfg = os.path.join(os.path.dirname(os.path.abspath(T.__file__)), "config", "vsp-resnet101dilated-ppm_deepsup_clip.yaml")
T.prepare(args, cfg); cfg.MODEL.arch_encoder = "resnet50dilated"
mod = T.build_module(cfg, args, K, training=True); load_det(mod); zero_dropout(mod); mod.to(dev).train()
opt = T.create_optimizers(mod, cfg, args)
tag = "tcb_train_trajectory_clip_ocr"
for it in range(2):
    imgs = [torch.from_numpy(det_input("%s:img:%d:%d" % (tag, it, t), (2, 3, 97, 97))).to(dev) for t in range(3)]
    labs = [torch.from_numpy(det_labels("%s:lab:%d:%d" % (tag, it, t), (2, 1, 97, 97), K)).to(dev) for t in range(3)]
    mod.zero_grad()
    loss, acc = mod(T.make_batch(args, imgs, labs, it))
    # verify the caches right after the forward (every U of this step's weights has been used)
    named = {p.data_ptr(): n for n, p in mod.named_parameters()}
    for m in (3, 4):
        c = OC._wu3_copies[m]
        bad = 0
        for ident, e in c.entries.items():
            w = e["ref"]()
            if w is None or w.data_ptr() not in named:
                continue
            k, cc = w.shape[0], w.shape[1]
            tmp = torch.empty_like(e["buf"])
            _C.call("vspw_wino%d_weights" % m, _p(w), _p(tmp[0]), k, cc, 0, _stream())
            _C.call("vspw_wino%d_weights" % m, _p(w), _p(tmp[1]), k, cc, 1, _stream())
            torch.cuda.synchronize()
            d0, d1 = float((tmp[0] - e["buf"][0]).abs().max()), float((tmp[1] - e["buf"][1]).abs().max())
            if d0 > 0 or d1 > 0:
                bad += 1
                print("  step %d F%d STALE %s: fwd %.3e dgrad %.3e (|U| %.3e) key %s vs %s" % (it, m, named[w.data_ptr()], d0, d1, float(tmp[0].abs().max()), e["key"], OC._wt_key(w)))
        print("step %d cache F%d: %d entries, %d stale" % (it, m, len(c.entries), bad))
    loss.mean().backward(); opt.step()
    print("step", it, "loss %.8f" % loss.item())
