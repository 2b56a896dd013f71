#!/usr/bin/env python
"""bench.py — the BASELINE.json metric on MI355X: "480p clips/s (T=5, B=2/GPU) train fwd+bwd".

One step = one training step of TCB-PSP (Clip_PSP, ResNet-101 dilated; reference config
vsp-resnet101dilated-ppm_deepsup_clip.yaml = BASELINE.json configs[2], the configuration the metric is quoted on) over
B=2 clips of T=5 frames, 479x479 crops, 124 classes per GPU: forward, fused loss, backward, gradient all-reduce
(N>1) and the SGD update — every FLOP in the hand-written HIP kernels of libvspw_hip.so.  Synthetic data (seed 304)
is resident in HBM before the timed region; weights are random-init (no network for checkpoints).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events recorded on the launch stream around every
launch of the dominant kernel (igemm_nt_kernel: all convolution forward and data-gradient GEMMs, ~2/3 of the step's
FLOPs) inside the timed region; `cpu_baseline` times the numpy oracle (a port of the reference's arithmetic, test
infrastructure) on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
K_CLASSES, T_FRAMES, B_CLIPS, CROP = 124, 5, 2, 479
GFLOP_PER_CLIP = 5785.0  # cfg 3 forward+backward, conv/bmm FLOPs


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/*_pmc_summary.json,
    produced by tools/gpu_profile.sh + tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE passes of this same
    bench command).  PMC counters cannot be read from inside the process, so this is the committed measurement, not
    a live one; None when no summary is present."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"][kernel]
        return k.get("hbm_bytes_per_launch"), os.path.basename(files[-1])
    except Exception:
        return None, None


def make_inputs(dev, seed):
    g = torch.Generator().manual_seed(seed)
    imgs = [torch.randn(B_CLIPS, 3, CROP, CROP, generator=g).to(dev) for _ in range(T_FRAMES)]
    labs = []
    for _ in range(T_FRAMES):
        lab = torch.randint(0, K_CLASSES, (B_CLIPS, 1, CROP, CROP), generator=g).float()
        lab[torch.rand(B_CLIPS, 1, CROP, CROP, generator=g) < 0.05] = 255.0
        labs.append(lab.to(dev))
    return imgs, labs


def cpu_baseline(budget_note=True):
    """Numpy-oracle port timed on the host cores: TCB-PSP R101 forward+backward on a bounded sample
    (B=2 clips x 1 of the 5 frames at 239x239, i.e. 1/5 of the frames at 1/4.02 of the pixels); cost is linear in
    frames and (to first order) in pixels, so clips/s = 2 / (t * 5 * (479/239)^2)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from helpers import build, det_numpy_state
    from oracle import np_models as NM
    from oracle import np_ops as O
    from oracle.det_init import det_input, det_labels

    O.set_dtype(np.float32)
    S = 239
    mod = build("clip_psp", "resnet101dilated")
    sd = det_numpy_state(mod)
    imgs = [det_input("bench:0", (B_CLIPS, 3, S, S))]
    labs = [det_labels("bench:0", (B_CLIPS, 1, S, S), K_CLASSES)]
    t0 = time.time()
    P = NM.Params(sd, train_params=True)
    loss, _ = NM.clip_psp(P, "resnet101", imgs, labs, True)
    O.tape().backward(loss)
    dt = time.time() - t0
    scale = T_FRAMES * (CROP / float(S)) ** 2
    cores = os.cpu_count() or 1
    try:  # threads the BLAS behind numpy.matmul actually used
        from threadpoolctl import threadpool_info

        blas = [i["num_threads"] for i in threadpool_info() if i.get("user_api") == "blas"]
        if blas:
            cores = max(blas)
    except Exception:
        pass
    return {"value": B_CLIPS / (dt * scale), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "numpy oracle (oracle/np_models.clip_psp, R101) fwd+bwd on B=2 clips x 1 frame at %dx%d: %.1f s; "
                      "scaled x%.1f (5 frames, (479/%d)^2 pixels) to one B=2,T=5,479^2 step" % (S, S, dt, scale, S)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-report", default="", help="write a per-shape GEMM efficiency table to this file")
    ap.add_argument("--method", default="clip_psp", choices=["clip_psp", "clip_ocr"])
    args = ap.parse_args()

    from cvpr2021_vspw_implement_amd import distributed as vdist
    from cvpr2021_vspw_implement_amd import models as M
    from cvpr2021_vspw_implement_amd import ops, optim

    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the hot path has no CPU fallback")
    rank, local_rank, world = vdist.init_from_env()
    if world != args.gpus:
        raise RuntimeError("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    torch.manual_seed(304)
    margs = types.SimpleNamespace(num_class=K_CLASSES, psp_weight=False, use_memory=False, memory_num=0,
                                  clipocr_all=False, clip_num=T_FRAMES)
    enc = M.ModelBuilder.build_encoder(arch="resnet101dilated", fc_dim=2048)
    crit = torch.nn.NLLLoss(ignore_index=255)
    cls = M.Clip_PSP if args.method == "clip_psp" else M.ClipOCRNet
    net = cls(enc, crit, margs, deep_sup_scale=0.4).to(dev)
    net.train()
    # param broadcast, bucketed grad all-reduce, SyncBN over RCCL (N>1); VSPW_FORCE_COLLECTIVES=1 runs the same
    # collectives in a 1-rank RCCL group (the only way to exercise them on a single-GPU box)
    force = os.environ.get("VSPW_FORCE_COLLECTIVES") == "1"
    model = vdist.DataParallelOverRCCL(net, force_collectives=force)
    opt = optim.create_optimizers(net, lr=0.002, weight_decay=1e-4, momentum=0.9)
    imgs, labs = make_inputs(dev, 304 + rank)
    max_iters = 1000

    def step(it):
        net.zero_grad()
        optim.adjust_learning_rate(opt, it, max_iters, 0.002)
        feed = {"img_data": imgs[0], "seg_label": labs[0], "clipimgs_data": list(imgs[1:]),
                "cliplabels_data": list(labs[1:]), "step": it}
        loss, acc = model(feed)
        loss = loss.mean()
        loss.backward()
        model.finish_gradients()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        loss = step(i)
    barrier()
    # per-launch HIP events (two per GEMM launch) cost ~2 % of the step: record them on a sample of the timed steps
    # (first, middle, last third) unless a full per-shape report was asked for
    if args.no_kernel_timing:
        timed_steps = set()
    elif args.kernel_report or args.steps <= 3:
        timed_steps = set(range(args.steps))
    else:
        timed_steps = {0, args.steps // 2, args.steps - 1}
    ops.kernel_timer(False)
    ops.kernel_timer_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.kernel_timer(i in timed_steps, reset=False)
        loss = step(args.warmup + i)
    host_enqueue = time.perf_counter() - t0  # host time to enqueue all K steps (GPU still running)
    barrier()
    elapsed = time.perf_counter() - t0
    ops.kernel_timer(False, reset=False)
    last_loss = float(loss.item())
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    roofline = None
    if not args.no_kernel_timing:
        allrecs = ops.kernel_timer_records()
        if args.kernel_report and rank == 0:
            agg = {}
            for name, fl, ms, tag in allrecs:
                a = agg.setdefault(tag, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += fl
                a[2] += ms
            with open(args.kernel_report, "w") as f:
                f.write("shape,launches_per_step,gflop_per_launch,avg_ms,tflops,ms_per_step\n")
                for tag, (cnt, fl, ms) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
                    f.write("%s,%.1f,%.2f,%.4f,%.1f,%.3f\n" % (tag, cnt / len(timed_steps), fl / cnt / 1e9, ms / cnt,
                                                             fl / ms / 1e9, ms / len(timed_steps)))
        recs = [r for r in allrecs if r[0] == "igemm_nt_kernel"]
        if recs:
            flops = sum(r[1] for r in recs)
            ms = sum(r[2] for r in recs)
            achieved = flops / (ms * 1e-3) / 1e12
            traffic, traffic_src = measured_traffic("igemm_nt_kernel")
            roofline = {"bound": "mfma", "kernel": "igemm_nt_kernel", "achieved": round(achieved, 2),
                        "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": traffic, "traffic_unit": "HBM bytes/launch (rocprofv3 PMC, %s)" % traffic_src,
                        "launches_per_step": len(recs) // max(len(timed_steps), 1),
                        "timed_steps": sorted(timed_steps),
                        "avg_launch_ms": round(ms / len(recs), 4),
                        "gflop_per_launch": round(flops / len(recs) / 1e9, 3),
                        "share_of_step_time": round(ms * 1e-3 / len(timed_steps) / (elapsed / args.steps), 3)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        clips_per_s = world * B_CLIPS * args.steps / elapsed
        out = {
            "metric": "480p clips/s (T=5, B=2/GPU) train fwd+bwd",
            "value": round(clips_per_s, 4),
            "unit": "clips/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seed 304), random-init weights",
            "config": {"workload": "TCB-PSP (Clip_PSP, resnet101dilated) train step: T=5 frames, B=2 clips/GPU, "
                                   "479x479 crop, 124 classes, fwd+loss+bwd+SGD (vsp-resnet101dilated-ppm_deepsup_clip)"
                       if args.method == "clip_psp" else
                       "TCB-OCR (ClipOCRNet, resnet101dilated) train step: T=5, B=2/GPU, 479x479, 124 classes",
                       "global_batch_clips": world * B_CLIPS, "frames_per_step_per_gpu": T_FRAMES * B_CLIPS,
                       "parallelism": "dp%d" % world, "sync_bn": world > 1 or force},
            "e2e_mfma_frac": round(GFLOP_PER_CLIP * 1e9 * clips_per_s / world / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4),
            "last_loss": round(last_loss, 5),
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 2),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
