"""Worker of tests/test_drivers_gpu.py::test_two_rank_training_driver (launched by torch.distributed.run, 2 ranks):
train_clip2.main on the tiny VSPW tree, then every rank writes a digest of its replica."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(tree, save):
    from cvpr2021_vspw_implement_amd import watchdog

    wd = watchdog.make(True, 100.0)  # a rank without progress for 100 s dumps its stacks and exits 3
    wd.phase("import + train_clip2.main (2 epochs on the tiny tree)", 150)
    import cvpr2021_vspw_implement_amd.train_clip2 as T
    from cvpr2021_vspw_implement_amd.config import cfg as base_cfg

    args = T.build_parser().parse_args([
        "--method", "clip_psp", "--dataroot", tree, "--saveroot", save, "--batchsize", "2", "--cropsize", "40",
        "--clip_num", "4", "--dilation2", "3,6,9", "--totalepoch", "2", "--ckpt_every", "2", "--lr", "0.01",
        "--workers", "0", "--gpus", "0,1"])
    cfg = base_cfg.clone()
    here = os.path.dirname(os.path.abspath(T.__file__))
    args.cfg = os.path.join(here, "config", "vsp-resnet101dilated-ppm_deepsup_clip.yaml")
    T.prepare(args, cfg)
    cfg.MODEL.arch_encoder = "resnet50dilated"
    captured = {}
    build = T.build_module

    def build_and_keep(*a, **k):
        captured["module"] = build(*a, **k)
        return captured["module"]

    T.build_module = build_and_keep
    hist = T.main(cfg, [0, 1], args)
    wd.phase("digest")
    rank = int(os.environ["RANK"])
    sd = captured["module"].state_dict()
    digest = np.array([float(v.double().sum().item()) for k, v in sorted(sd.items())]
                      + [float(v.double().abs().sum().item()) for k, v in sorted(sd.items())])
    np.save(os.path.join(save, "rank%d_digest.npy" % rank), digest)
    np.save(os.path.join(save, "rank%d_loss.npy" % rank), np.array(hist["train"]["loss"]))
    wd.stop()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
