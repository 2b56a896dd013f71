"""Training-step timing of NetWarp (cfg 5c: ResNet-101 dilated + ppm_deepsup_clip, clip_num 2, B = 2 frame pairs of
479x479, frozen RAFT with 20 iterations on the HIP kernels, FlowCNN, two flow-warps + blends, loss, backward, SGD)."""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr2021_vspw_implement_amd import models as M  # noqa: E402
from cvpr2021_vspw_implement_amd import optim  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(304)
    B, S = 2, 479
    args = types.SimpleNamespace(num_class=124, clip_num=2, raft_weights=None)
    enc = M.ModelBuilder.build_encoder(arch="resnet101dilated", fc_dim=2048)
    dec = M.ModelBuilder.build_decoder(arch="ppm_deepsup_clip", fc_dim=2048, num_class=124)
    net = M.NetWarp(enc, dec, torch.nn.NLLLoss(ignore_index=255), args, 0.4).to(dev).train()
    opt = optim.create_optimizers(net, lr=0.002)
    cur, prev = torch.randn(B, 3, S, S, device=dev), torch.randn(B, 3, S, S, device=dev)
    lab = torch.randint(0, 124, (B, 1, S, S), device=dev).float()

    def step():
        net.zero_grad()
        loss, acc = net({"img_data": cur, "seg_label": lab, "clipimgs_data": [prev], "cliplabels_data": []})
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    reps = 8
    t0 = time.perf_counter()
    for _ in range(reps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(json.dumps({"workload": "NetWarp R101 train step, B=2 pairs, 479x479, RAFT 20 iters on HIP",
                      "ms_per_step": round(ms, 2), "pairs_per_s": round(B / ms * 1e3, 2),
                      "loss_finite": bool(torch.isfinite(loss).item())}))


if __name__ == "__main__":
    main()
