"""End-to-end throughput of the training drivers' real input pipeline (reference dataset2.py:852-1048):
a synthetic VSPW-format tree of 480x853 JPEG frames + PNG masks on local disk -> BaseDataset_longclip in W DataLoader
worker processes (PIL decode) -> DeviceTransform (HIP: flip / multi-scale resize / pad / crop / normalise / NHWC / label
remap) -> one TCB-PSP R101 training step (B = 2 clips x T = 5 frames, 479x479 crop), next to bench.py's resident-data
number.  Three measurements:
  loader      : clips/s the W workers can decode (no GPU work)                 - where decode stops being the limiter
  transform   : HIP-event time of DeviceTransform per batch + bytes moved      - data.hip against the HBM roofline
  end to end  : clips/s of the eager training loop fed by the loader           - vs the resident-data step

usage: python tools/pipeline_bench.py [--videos 24] [--frames 16] [--workers 0,4,8,16,32] [--steps 12] [--root DIR]
       --no-train: skip the end-to-end part (loader + transform only)."""
import argparse
import json
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

H, W, CROP, T, B, K = 480, 853, 479, 5, 2, 124


def make_tree(root, videos, frames):
    """VSPW on-disk layout (dataset2.py:866-884): <root>/train.txt, data/<video>/origin/*.jpg, data/<video>/mask/*.png.
    Frames are smooth random fields + noise (JPEG-compressible like natural video: ~90-130 KB at quality 92)."""
    from PIL import Image

    os.makedirs(root, exist_ok=True)
    names = ["v%03d" % i for i in range(videos)]
    with open(os.path.join(root, "train.txt"), "w") as f:
        for v in names:
            f.write(v + "\n")
    rs = np.random.RandomState(7)
    nbytes = 0
    for v in names:
        od, md = os.path.join(root, "data", v, "origin"), os.path.join(root, "data", v, "mask")
        os.makedirs(od, exist_ok=True)
        os.makedirs(md, exist_ok=True)
        base = rs.randint(0, 256, size=(H // 16 + 2, W // 16 + 2, 3)).astype(np.float32)
        lab = rs.randint(0, 125, size=(H // 32 + 2, W // 32 + 2))
        big = np.kron(base, np.ones((16, 16, 1), np.float32))
        seg0 = np.kron(lab, np.ones((32, 32), np.int64)).astype(np.uint8)
        for t in range(frames):
            img = np.clip(big[t % 8:t % 8 + H, t % 8:t % 8 + W] + rs.randn(H, W, 3) * 5.0, 0, 255).astype(np.uint8)
            stem = "%08d" % (3 * t + 1)
            p = os.path.join(od, stem + ".jpg")
            Image.fromarray(img, "RGB").save(p, quality=92)
            nbytes += os.path.getsize(p)
            Image.fromarray(seg0[t % 8:t % 8 + H, t % 8:t % 8 + W], "L").save(os.path.join(md, stem + ".png"))
    return nbytes / float(videos * frames)


def loader(args_ns, workers, batch=B):
    from cvpr2021_vspw_implement_amd.dataset2 import BaseDataset_longclip, collate_raw

    ds = BaseDataset_longclip(args_ns, "train")
    return torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=True, num_workers=workers, drop_last=True,
                                       pin_memory=False, collate_fn=collate_raw, persistent_workers=workers > 0,
                                       prefetch_factor=4 if workers > 0 else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=24)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--workers", default="0,4,8,16,32")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--root", default="")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--multi-scale", action="store_true", help="draw the reference's scale in {0.8,1,1.5,2} per clip")
    a = ap.parse_args()
    root = a.root or os.path.join(tempfile.gettempdir(), "vspw_synth_%dx%d" % (a.videos, a.frames))
    if not os.path.exists(os.path.join(root, "train.txt")):
        t0 = time.time()
        avg = make_tree(root, a.videos, a.frames)
        print("# synthetic VSPW tree: %d videos x %d frames at %dx%d under %s (%.0f KB/jpg, %.1f s to write)"
              % (a.videos, a.frames, H, W, root, avg / 1024, time.time() - t0))
    ns = types.SimpleNamespace(cropsize=CROP, dataroot=root, trainfps=1, clip_num=T, dilation2="3,6,9,12",
                               multi_scale=a.multi_scale, lesslabel=False)
    out = {"tree": {"videos": a.videos, "frames": a.frames, "hw": [H, W]}, "host_cpus": os.cpu_count(), "loader": [],
           "multi_scale": bool(a.multi_scale)}
    # ---- 1. loader only -------------------------------------------------------------------------------------------
    for wk in [int(x) for x in a.workers.split(",")]:
        dl = loader(ns, wk)
        it = iter(dl)
        n_batches = max(2, min(len(dl), 6 if wk == 0 else 4 * max(wk, 4) // B))
        got = 0
        next(it)  # workers started, first batch decoded
        t0 = time.time()
        for _ in range(n_batches):
            try:
                next(it)
            except StopIteration:
                it = iter(dl)
                next(it)
            got += 1
        dt = time.time() - t0
        out["loader"].append({"workers": wk, "clips_per_s": round(got * B / dt, 2), "frames_per_s": round(got * B * T / dt, 1)})
        print("# loader W=%-2d: %.2f clips/s (%.1f frames/s decoded)" % (wk, got * B / dt, got * B * T / dt))
        del it, dl
    if not torch.cuda.is_available():
        print(json.dumps(out))
        return
    # ---- 2. DeviceTransform ------------------------------------------------------------------------------------------
    from cvpr2021_vspw_implement_amd.dataset2 import DeviceTransform

    dev = torch.device("cuda:0")
    tf = DeviceTransform(dev)
    dl = loader(ns, 8)
    batches = []
    for i, bt in enumerate(dl):
        batches.append(bt)
        if i >= 3:
            break
    tf(batches[0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    for bt in batches:
        tf(bt)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.time() - t0) / len(batches)
    gpu_ms = e0.elapsed_time(e1) / len(batches)
    # bytes per batch: uint8 frame + mask in (PCIe + read), fp32 image + label out
    b_in = B * T * (H * W * 3 + H * W)
    b_out = B * T * (CROP * CROP * 3 * 4 + CROP * CROP * 4)
    out["transform"] = {"ms_per_batch_stream": round(gpu_ms, 3), "ms_per_batch_wall": round(wall * 1e3, 3),
                        "bytes_in": b_in, "bytes_out": b_out,
                        "hbm_gbps_if_kernels_only": round((b_in + b_out) / (gpu_ms * 1e-3) / 1e9, 1),
                        "note": "stream time includes the pageable host->device copies of the uint8 frames (PCIe); the "
                                "kernels' own time is in the rocprofv3 trace of this tool"}
    print("# DeviceTransform: %.2f ms/batch on the stream (%.2f ms wall), %.1f MB in, %.1f MB out"
          % (gpu_ms, wall * 1e3, b_in / 1e6, b_out / 1e6))
    if a.no_train:
        print(json.dumps(out))
        return
    # ---- 3. end to end -----------------------------------------------------------------------------------------------
    from cvpr2021_vspw_implement_amd import models as M
    from cvpr2021_vspw_implement_amd import optim
    from cvpr2021_vspw_implement_amd.train_clip2 import make_batch

    margs = types.SimpleNamespace(num_class=K, psp_weight=False, use_memory=False, memory_num=0, clipocr_all=False,
                                  clip_num=T, method="clip_psp")
    enc = M.ModelBuilder.build_encoder(arch="resnet101dilated", fc_dim=2048)
    net = M.Clip_PSP(enc, torch.nn.NLLLoss(ignore_index=255), margs, deep_sup_scale=0.4).to(dev).train()
    opt = optim.create_optimizers(net, lr=0.002, weight_decay=1e-4, momentum=0.9)

    def step(imgs, labs):
        net.zero_grad()
        loss, acc = net(make_batch(margs, imgs, labs, 0))
        loss.backward()
        opt.step()
        return loss

    imgs, labs = tf(batches[0])
    for _ in range(3):
        step(imgs, labs)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        step(imgs, labs)
    torch.cuda.synchronize()
    resident = (time.time() - t0) / a.steps
    out["resident_ms_per_step"] = round(resident * 1e3, 2)
    out["end_to_end"] = []
    print("# resident data, eager launches: %.2f ms/step = %.2f clips/s" % (resident * 1e3, B / resident))
    for wk in [int(x) for x in a.workers.split(",") if int(x) > 0]:
        dl = loader(ns, wk)
        it = iter(dl)
        imgs, labs = tf(next(it))
        step(imgs, labs)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 0
        while n < a.steps:
            try:
                bt = next(it)
            except StopIteration:
                it = iter(dl)
                bt = next(it)
            imgs, labs = tf(bt)
            step(imgs, labs)
            n += 1
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        out["end_to_end"].append({"workers": wk, "ms_per_step": round(dt * 1e3, 2), "clips_per_s": round(B / dt, 2),
                                  "vs_resident": round(resident / dt, 3)})
        print("# end to end W=%-2d: %.2f ms/step = %.2f clips/s (%.0f %% of resident)" % (wk, dt * 1e3, B / dt,
                                                                                         100 * resident / dt))
        del it, dl
    print(json.dumps(out))


if __name__ == "__main__":
    main()
