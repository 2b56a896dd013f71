"""Training-step timing of the per-frame PSPNet of BASELINE.json cfg 2 (SegmentationModule(resnet101dilated,
ppm_deepsup), B = 2 frames of 479x479, two fused SGDs as train.py builds them): launch by launch against the captured
hipGraph (`train.py --hip_graph`) - a 2-frame step is shorter than the time Python needs to issue its launches."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr2021_vspw_implement_amd import train as T  # noqa: E402
from cvpr2021_vspw_implement_amd.config import cfg as base_cfg  # noqa: E402
from cvpr2021_vspw_implement_amd.graph import GraphedStep  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(304)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mode = sys.argv[2] if len(sys.argv) > 2 else "both"
    yaml = sys.argv[3] if len(sys.argv) > 3 else "vsp-resnet101dilated-ppm_deepsup.yaml"  # e.g. ...-nonlocal2d.yaml (cfg 5a)
    if mode == "both":  # one process per mode: a capture after eager steps would meet their AccumulateGrad nodes
        import subprocess

        outs = [json.loads(subprocess.run([sys.executable, os.path.abspath(__file__), str(B), m, yaml], capture_output=True,
                                          text=True, check=True).stdout.strip().splitlines()[-1]) for m in ("eager", "graph")]
        print(json.dumps({"workload": outs[0]["workload"], "eager_ms_per_step": outs[0]["ms_per_step"],
                          "hip_graph_ms_per_step": outs[1]["ms_per_step"],
                          "frames_per_s_graph": round(B / outs[1]["ms_per_step"] * 1e3, 1),
                          "finite": outs[0]["finite"] and outs[1]["finite"]}))
        return
    S, K = 479, 124
    here = os.path.dirname(os.path.abspath(T.__file__))
    args = T.build_parser().parse_args(["--cfg", os.path.join(here, "config", yaml),
                                        "--predir", "", "--lr", "0.002"])
    cfg = base_cfg.clone()
    T.prepare(args, cfg)
    net, nets = T.build_module(cfg, args)
    net.to(dev).train()
    opts = T.create_optimizers(nets, cfg)
    img = torch.randn(B, 3, S, S, device=dev)
    lab = torch.randint(0, K, (B, 1, S, S), device=dev).float()

    def step():
        net.zero_grad()
        loss, _ = net({"img_data": img, "seg_label": lab})
        loss = loss.mean()
        loss.backward()
        for o in opts:
            o.step()
        return loss

    def timeit(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    if mode == "eager":
        ms, loss = timeit(step, 10)
    else:
        graph = GraphedStep(step, warmup=2)
        ms, out = timeit(graph.replay, 20)
        loss = out
    print(json.dumps({"workload": "per-frame %s + %s train step (train.py's model and optimizers), B=%d, 479x479"
                      % (cfg.MODEL.arch_encoder, cfg.MODEL.arch_decoder, B),
                      "ms_per_step": round(ms, 2), "finite": bool(torch.isfinite(loss).item())}))


if __name__ == "__main__":
    main()
