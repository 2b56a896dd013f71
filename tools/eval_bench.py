"""Inference timing of the clip heads as test_clip2.py runs them: one 480x853 target frame + 3 context frames, segSize =
the frame size, softmax probabilities out (Clip_PSP / ClipOCRNet on ResNet-101 dilated).  One JSON line per method."""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr2021_vspw_implement_amd import models as M  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(304)
    H, W, T = 480, 853, 4
    for method in ("clip_psp", "clip_ocr"):
        args = types.SimpleNamespace(num_class=124, psp_weight=False, use_memory=False, memory_num=0, clipocr_all=False,
                                     clip_num=T)
        enc = M.ModelBuilder.build_encoder(arch="resnet101dilated", fc_dim=2048)
        cls = M.Clip_PSP if method == "clip_psp" else M.ClipOCRNet
        net = cls(enc, torch.nn.NLLLoss(ignore_index=-1), args).to(dev).eval()
        frames = [torch.randn(1, 3, H, W, device=dev) for _ in range(T)]
        lab = torch.zeros(1, 1, H, W, device=dev)

        def run():
            with torch.no_grad():
                return net({"img_data": frames[0], "seg_label": lab, "clipimgs_data": list(frames[1:])}, segSize=(H, W))

        for _ in range(2):
            out = run()
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            out = run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({"workload": "%s R101 eval, 480x853, %d frames per prediction" % (method, T),
                          "ms_per_prediction": round(ms, 2), "predictions_per_s": round(1e3 / ms, 2),
                          "out_shape": list(out.shape), "finite": bool(torch.isfinite(out).all().item())}))


if __name__ == "__main__":
    main()
