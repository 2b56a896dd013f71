#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported read-only from /root/reference) on small,
seeded cases.  Runs only in the build container (the GPU box has no /root/reference); the fixtures it writes contain
arrays only: inputs are regenerated from seeds (oracle/det_init.py), weights likewise, so a fixture holds the expected
outputs (logits, probabilities, arg-max, loss, accuracy, gradients / gradient norms).

    python tests/golden/make_golden.py            # (re)writes every fixture

Nothing from the reference is copied: it is imported, executed and discarded.
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle.det_init import det_input, det_labels, det_sample_index, det_tensor  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
K = 124


class _Normalize(object):
    """torchvision.transforms.Normalize (functional.normalize: sub_(mean[:,None,None]).div_(std[:,None,None]) in the
    tensor's dtype) - torchvision itself is not installed here."""

    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        t = t.clone()
        m = torch.as_tensor(self.mean, dtype=t.dtype)
        s = torch.as_tensor(self.std, dtype=t.dtype)
        return t.sub_(m[:, None, None]).div_(s[:, None, None])


def import_reference():
    tv = sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    if not hasattr(tv, "transforms"):
        tv.transforms = types.ModuleType("torchvision.transforms")
        tv.transforms.Normalize = _Normalize
        sys.modules["torchvision.transforms"] = tv.transforms
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "RAFT_core"))
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import models as ref_models  # noqa
    finally:
        os.chdir(cwd)
    return ref_models


def load_det(module, seed=304):
    sd = module.state_dict()
    new = {k: torch.from_numpy(det_tensor(k, v.shape, seed)).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)


def calibrate_bn(module, run_train_forward):
    """Make eval-mode BatchNorm meaningful for random weights: one training-mode forward with momentum 1 sets every
    running statistic to the batch statistic of the calibration input.  Returns them for storage in the fixture."""
    bns = [m for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    for m in bns:
        m.momentum = 1.0
    module.train()
    with torch.no_grad():
        run_train_forward()
    for m in bns:
        m.momentum = 0.1
    return {"bnstat:" + k: v.numpy().copy() for k, v in module.state_dict().items()
            if k.endswith("running_mean") or k.endswith("running_var")}


def double_pass(module, reload_sd, run_train, hook_mod=None, run_eval=None):
    """Same case in float64 (reference cast with .double()): pins semantics free of fp32 rounding noise."""
    module.double()
    module.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in reload_sd.items()}, strict=False)
    out = {}
    if run_eval is not None:
        module.eval()
        store = {}
        h = hook_output(hook_mod, store, "logits")
        with torch.no_grad():
            run_eval()
        h.remove()
        out["eval_logits64"] = store["logits"].numpy().copy()
    module.train()
    module.zero_grad()
    loss, acc = run_train()
    loss.backward()
    out["train_loss64"] = np.float64(loss.item())
    names, norms = [], []
    for k, p in module.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(float(p.grad.norm()))
    out["grad_norms64"] = np.array(norms, dtype=np.float64)
    module.float()
    return out


def zero_dropout(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0


def grads_summary(module, full=()):
    out = {}
    names, norms = [], []
    for k, p in module.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if k in full:
            out["grad:" + k] = p.grad.detach().numpy().copy()
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, dtype=np.float64)
    return out


def top2_margin(probs):
    """log p(top1) - log p(top2) = gap between the two largest up-sampled LOGITS (softmax is monotone)."""
    s = np.sort(probs.astype(np.float64), axis=1)
    return (np.log(s[:, -1]) - np.log(np.maximum(s[:, -2], 1e-300))).astype(np.float32)


def eval_pack(probs, logits):
    probs = probs.detach().numpy()
    out = {"probs_sub": probs[:, :, ::4, ::4].copy(), "argmax": probs.argmax(1).astype(np.uint8),
           "margin": top2_margin(probs)}
    if logits is not None:
        out["logits"] = logits.detach().numpy().copy()
    return out


def hook_output(mod, store, key):
    return mod.register_forward_hook(lambda m, i, o: store.__setitem__(key, o.detach().clone()))


def args_ns(**kw):
    base = dict(num_class=K, psp_weight=False, use_memory=False, memory_num=0, clipocr_all=False, clip_num=3)
    base.update(kw)
    return types.SimpleNamespace(**base)


def case_segmodule(M, arch, decoder, tag, train_shape=(2, 3, 65, 65), eval_shape=(1, 3, 64, 96), fc_dim=512):
    torch.manual_seed(0)
    enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=fc_dim)
    dec = M.ModelBuilder.build_decoder(arch=decoder, fc_dim=fc_dim, num_class=K)
    crit = torch.nn.NLLLoss(ignore_index=255)
    mod = M.SegmentationModule(enc, dec, crit, 0.4 if decoder != "nonlocal2d" else None)
    load_det(mod)
    zero_dropout(mod)
    res = {}
    cal_img = torch.from_numpy(det_input(tag + ":train", train_shape))
    cal_lab = torch.from_numpy(det_labels(tag + ":train", (train_shape[0], 1) + train_shape[2:], K))
    res.update(calibrate_bn(mod, lambda: mod({"img_data": cal_img, "seg_label": cal_lab})))
    calibrated_sd = {k: v.clone() for k, v in mod.state_dict().items()}
    # eval
    mod.eval()
    store = {}
    last = {"ppm_deepsup": "conv_last_", "ocrnet_deepsup": "head", "nonlocal2d": "last_layer"}[decoder]
    h = hook_output(getattr(mod.decoder, last), store, "logits")
    img = torch.from_numpy(det_input(tag + ":eval", eval_shape))
    with torch.no_grad():
        probs = mod({"img_data": img, "seg_label": torch.zeros(eval_shape[0], 1, *eval_shape[2:])},
                    segSize=eval_shape[2:])
    h.remove()
    for k, v in eval_pack(probs, store["logits"]).items():
        res["eval_" + k] = v
    # train
    mod.train()
    img = torch.from_numpy(det_input(tag + ":train", train_shape))
    lab = torch.from_numpy(det_labels(tag + ":train", (train_shape[0], 1) + train_shape[2:], K))
    mod.zero_grad()
    loss, acc = mod({"img_data": img, "seg_label": lab})
    loss.backward()
    res["train_loss"] = np.float64(loss.item())
    res["train_acc"] = np.float64(acc.item())
    res.update(grads_summary(mod, full=("encoder.conv1.weight", "encoder.layer1.0.conv2.weight",
                                        "encoder.layer4.1.bn2.weight", "encoder.layer2.0.downsample.0.weight")))
    res["bn_running_mean:encoder.bn1"] = mod.encoder.bn1.running_mean.numpy().copy()
    res["bn_running_var:encoder.bn1"] = mod.encoder.bn1.running_var.numpy().copy()
    eimg = torch.from_numpy(det_input(tag + ":eval", eval_shape)).double()
    res.update(double_pass(
        mod, calibrated_sd, lambda: mod({"img_data": img.double(), "seg_label": lab.double()}),
        getattr(mod.decoder, last),
        lambda: mod({"img_data": eimg, "seg_label": torch.zeros(eval_shape[0], 1, *eval_shape[2:])},
                    segSize=eval_shape[2:])))
    res["meta"] = np.array([arch, decoder, str(train_shape), str(eval_shape), str(fc_dim)])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "loss %.6f acc %.4f" % (loss.item(), acc.item()))


def case_clip(M, method, arch, tag, T=3, train_shape=(2, 3, 65, 65), eval_shape=(1, 3, 64, 96), **argkw):
    torch.manual_seed(0)
    enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=2048)
    crit = torch.nn.NLLLoss(ignore_index=255)
    cls = {"clip_psp": M.Clip_PSP, "clip_ocr": M.ClipOCRNet}[method]
    mod = cls(enc, crit, args_ns(**argkw), deep_sup_scale=0.4)
    load_det(mod)
    zero_dropout(mod)
    res = {}
    cimgs = [torch.from_numpy(det_input("%s:train:%d" % (tag, t), train_shape)) for t in range(T)]
    clabs = [torch.from_numpy(det_labels("%s:train:%d" % (tag, t), (train_shape[0], 1) + train_shape[2:], K))
             for t in range(T)]
    res.update(calibrate_bn(mod, lambda: mod({"img_data": cimgs[-1], "seg_label": clabs[-1],
                                              "clipimgs_data": list(cimgs[:-1]), "cliplabels_data": list(clabs[:-1])})))
    calibrated_sd = {k: v.clone() for k, v in mod.state_dict().items()}
    mod.eval()
    store = {}
    last = mod.ppm_conv.conv_last_ if method == "clip_psp" else mod.head
    h = hook_output(last, store, "logits")
    imgs = [torch.from_numpy(det_input("%s:eval:%d" % (tag, t), eval_shape)) for t in range(T)]
    with torch.no_grad():
        probs = mod({"img_data": imgs[-1], "clipimgs_data": imgs[:-1],
                     "seg_label": torch.zeros(eval_shape[0], 1, *eval_shape[2:])}, segSize=eval_shape[2:])
    h.remove()
    for k, v in eval_pack(probs, store["logits"]).items():
        res["eval_" + k] = v
    mod.train()
    imgs = [torch.from_numpy(det_input("%s:train:%d" % (tag, t), train_shape)) for t in range(T)]
    labs = [torch.from_numpy(det_labels("%s:train:%d" % (tag, t), (train_shape[0], 1) + train_shape[2:], K))
            for t in range(T)]
    mod.zero_grad()
    loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1],
                     "cliplabels_data": labs[:-1]})
    loss.backward()
    res["train_loss"] = np.float64(loss.item())
    res["train_acc"] = np.float64(acc.item())
    full = ("encoder.conv1.weight", "encoder.layer1.0.conv2.weight", "encoder.layer4.2.bn3.weight")
    res.update(grads_summary(mod, full=full))
    eimgs = [torch.from_numpy(det_input("%s:eval:%d" % (tag, t), eval_shape)).double() for t in range(T)]
    res.update(double_pass(
        mod, calibrated_sd,
        lambda: mod({"img_data": imgs[-1].double(), "seg_label": labs[-1].double(),
                     "clipimgs_data": [i.double() for i in imgs[:-1]],
                     "cliplabels_data": [l.double() for l in labs[:-1]]}),
        last,
        lambda: mod({"img_data": eimgs[-1], "clipimgs_data": list(eimgs[:-1]),
                     "seg_label": torch.zeros(eval_shape[0], 1, *eval_shape[2:])}, segSize=eval_shape[2:])))
    res["meta"] = np.array([arch, method, str(T), str(train_shape), str(eval_shape)])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "loss %.6f acc %.4f" % (loss.item(), acc.item()))


def case_ocr_memory(M, arch, tag, T=3, shape=(1, 3, 64, 96), calls=3, memory_num=4):
    """ClipOCRNet inference with the context memory bank (use_memory): three consecutive frames of one video; pins
    the bank's (quirky) persistence rule of spatial_ocr_block.py:110-125."""
    torch.manual_seed(0)
    enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=2048)
    mod = M.ClipOCRNet(enc, torch.nn.NLLLoss(ignore_index=255), args_ns(use_memory=True, memory_num=memory_num),
                       deep_sup_scale=0.4)
    load_det(mod)
    cimgs = [torch.from_numpy(det_input("%s:cal:%d" % (tag, t), (2, 3, 65, 65))) for t in range(T)]
    clabs = [torch.from_numpy(det_labels("%s:cal:%d" % (tag, t), (2, 1, 65, 65), K)) for t in range(T)]
    res = calibrate_bn(mod, lambda: mod({"img_data": cimgs[-1], "seg_label": clabs[-1],
                                         "clipimgs_data": list(cimgs[:-1]), "cliplabels_data": list(clabs[:-1])}))
    mod.eval()
    for c in range(calls):
        imgs = [torch.from_numpy(det_input("%s:call%d:%d" % (tag, c, t), shape)) for t in range(T)]
        with torch.no_grad():
            probs = mod({"img_data": imgs[-1], "clipimgs_data": imgs[:-1], "is_clean_memory": c == 0,
                         "seg_label": torch.zeros(shape[0], 1, *shape[2:])}, segSize=shape[2:])
        res["call%d_probs_sub" % c] = probs.numpy()[:, :, ::4, ::4].copy()
        res["call%d_memlen" % c] = np.int64(len(mod.memory))
    res["meta"] = np.array([arch, "clip_ocr_memory", str(T), str(shape), str(calls), str(memory_num)])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "memory lens", [int(res["call%d_memlen" % c]) for c in range(calls)])


def case_netwarp_ocr(M, arch, tag, shape=(2, 3, 65, 65)):
    import models.netwarp_ocr as ref_nw

    class FakeRaft(torch.nn.Module):
        def forward(self, a, b, iters=20, test_mode=True):
            n, _, h, w = a.shape
            f = torch.from_numpy(det_input(tag + ":flow", (n, 2, h, w), scale=1.9)) - 0.7
            return None, f.clamp(-10, 10)

    orig_raft, orig_load = ref_nw.RAFT, torch.load
    ref_nw.RAFT = lambda: torch.nn.Identity()
    torch.load = lambda *a, **k: {}
    torch.manual_seed(0)
    try:
        enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=2048)
        mod = M.NetWarp_ocr(enc, torch.nn.NLLLoss(ignore_index=255), args_ns(clip_num=2), deep_sup_scale=0.4)
    finally:
        ref_nw.RAFT, torch.load = orig_raft, orig_load
    mod.raft = FakeRaft()
    sd = {k: v for k, v in mod.state_dict().items() if not k.startswith("raft.")}
    mod.load_state_dict({k: torch.from_numpy(det_tensor(k, v.shape)).to(v.dtype) for k, v in sd.items()}, strict=False)
    zero_dropout(mod)
    mod.train()
    cur = torch.from_numpy(det_input(tag + ":cur", shape))
    prev = torch.from_numpy(det_input(tag + ":prev", shape))
    lab = torch.from_numpy(det_labels(tag + ":lab", (shape[0], 1) + shape[2:], K))
    plab = torch.from_numpy(det_labels(tag + ":plab", (shape[0], 1) + shape[2:], K))
    loss, acc = mod({"img_data": cur, "seg_label": lab, "clipimgs_data": [prev], "cliplabels_data": [plab]})
    loss.backward()
    res = {"train_loss": np.float64(loss.item()), "train_acc": np.float64(acc.item())}
    res.update(grads_summary(mod, full=("w0_1", "w1_0", "flowcnn.conv4.0.weight")))
    res["meta"] = np.array([arch, "netwarp_ocr", str(shape)])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "loss %.6f acc %.4f" % (loss.item(), acc.item()))


def case_nonlocal3d(M, arch, tag, T=3, train_shape=(1, 3, 49, 49)):
    torch.manual_seed(0)
    enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=2048)
    crit = torch.nn.NLLLoss(ignore_index=255)
    mod = M.Non_local3d(args_ns(), enc, crit)
    load_det(mod)
    imgs = [torch.from_numpy(det_input("%s:train:%d" % (tag, t), train_shape)) for t in range(T)]
    labs = [torch.from_numpy(det_labels("%s:train:%d" % (tag, t), (train_shape[0], 1) + train_shape[2:], K))
            for t in range(T)]
    res = calibrate_bn(mod, lambda: mod({"clipimgs_data": imgs, "cliplabels_data": labs}))
    mod.train()
    loss, acc = mod({"clipimgs_data": imgs, "cliplabels_data": labs})
    loss.backward()
    res.update({"train_loss": np.float64(loss.item()), "train_acc": np.float64(acc.item())})
    res.update(grads_summary(mod, full=("nonlocalblock.theta.weight", "nonlocalblock.W_z.1.weight")))
    mod.eval()
    with torch.no_grad():
        preds = mod({"clipimgs_data": imgs, "cliplabels_data": labs}, segSize=(49, 49))
    res["eval_argmax"] = np.stack([p.numpy().argmax(1) for p in preds]).astype(np.uint8)
    res["eval_margin"] = np.stack([top2_margin(p.numpy()) for p in preds])
    res["eval_probs_sub"] = np.stack([p.numpy()[:, :, ::4, ::4] for p in preds])
    res["meta"] = np.array([arch, "nonlocal3d", str(T), str(train_shape)])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "loss %.6f acc %.4f" % (loss.item(), acc.item()))


def case_nonlocal_downsample(M, arch, tag, T=3, shape=(2, 3, 73, 73)):
    """The avg-pool `downsample` switch of both non-local decoders (reference models/non_local_models.py:30-32,43-44,
    135-138): Non_local3d(downsample=True) over T frames and SegmentationModule(encoder, Non_local2d(downsample=True)).
    73x73 crops -> 10x10 embedding... (the 2x2 pool halves it to 5x5, bilinear back up)."""
    from models.non_local_models import Non_local2d  # reference

    res = {}
    crit = torch.nn.NLLLoss(ignore_index=255)
    for which in ("3d", "2d"):
        torch.manual_seed(0)
        enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=2048)
        if which == "3d":
            mod = M.Non_local3d(args_ns(), enc, crit, downsample=True)
            imgs = [torch.from_numpy(det_input("%s:3d:%d" % (tag, t), shape)) for t in range(T)]
            labs = [torch.from_numpy(det_labels("%s:3d:%d" % (tag, t), (shape[0], 1) + shape[2:], K)) for t in range(T)]
            feed = lambda: {"clipimgs_data": list(imgs), "cliplabels_data": list(labs)}  # noqa: E731
        else:
            mod = M.SegmentationModule(enc, Non_local2d(num_class=K, downsample=True), crit, None)
            img = torch.from_numpy(det_input("%s:2d" % tag, shape))
            lab = torch.from_numpy(det_labels("%s:2d" % tag, (shape[0], 1) + shape[2:], K))
            feed = lambda: {"img_data": img, "seg_label": lab}  # noqa: E731
        load_det(mod)
        # W_z's BatchNorm gamma is zero-initialised by the reference's constructor; load_det gave it U(0.5,1.5), so the
        # block contributes
        for k, v in calibrate_bn(mod, lambda: mod(feed())).items():
            res["%s:%s" % (which, k)] = v
        mod.train()
        mod.zero_grad()
        loss, acc = mod(feed())
        loss.backward()
        res[which + ":train_loss"] = np.float64(loss.item())
        res[which + ":train_acc"] = np.float64(acc.item())
        for k, v in grads_summary(mod, full=("encoder.layer4.2.conv3.weight",)).items():
            res["%s:%s" % (which, k)] = v
        mod.eval()
        with torch.no_grad():
            out = mod(feed(), segSize=shape[2:])
        preds = out if isinstance(out, (list, tuple)) else [out]
        res[which + ":eval_probs_sub"] = np.stack([p.numpy()[:, :, ::2, ::2] for p in preds])
        res[which + ":eval_argmax"] = np.stack([p.numpy().argmax(1) for p in preds]).astype(np.uint8)
        res[which + ":eval_margin"] = np.stack([top2_margin(p.numpy()) for p in preds])
        print(tag, which, "loss %.6f acc %.4f" % (loss.item(), acc.item()))
    res["meta"] = np.array([arch, str(T), str(shape)])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)


def case_netwarp(M, arch, tag, shape=(2, 3, 65, 65)):
    """NetWarp with the flow network replaced by a fixed synthetic field (the RAFT checkpoint is not in the tree):
    pins FlowCNN, the nearest-resized un-rescaled flow, both flow-warps, the per-channel blends and the loss."""
    import models.netwarp as ref_nw

    class FakeRaft(torch.nn.Module):
        def forward(self, a, b, iters=20, test_mode=True):
            n, _, h, w = a.shape
            f = torch.from_numpy(det_input(tag + ":flow", (n, 2, h, w), scale=1.9)) - 0.7
            return None, f.clamp(-10, 10)

    orig_raft, orig_load = ref_nw.RAFT, torch.load
    ref_nw.RAFT = lambda: torch.nn.Identity()
    torch.load = lambda *a, **k: {}
    torch.manual_seed(0)
    try:
        enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=2048)
        dec = M.ModelBuilder.build_decoder(arch="ppm_deepsup_clip", fc_dim=2048, num_class=K)
        mod = M.NetWarp(enc, dec, torch.nn.NLLLoss(ignore_index=255), args_ns(clip_num=2), deep_sup_scale=0.4)
    finally:
        ref_nw.RAFT, torch.load = orig_raft, orig_load
    mod.raft = FakeRaft()
    sd = {k: v for k, v in mod.state_dict().items() if not k.startswith("raft.")}
    mod.load_state_dict({k: torch.from_numpy(det_tensor(k, v.shape)).to(v.dtype) for k, v in sd.items()}, strict=False)
    zero_dropout(mod)
    mod.train()
    cur = torch.from_numpy(det_input(tag + ":cur", shape))
    prev = torch.from_numpy(det_input(tag + ":prev", shape))
    lab = torch.from_numpy(det_labels(tag + ":lab", (shape[0], 1) + shape[2:], K))
    loss, acc = mod({"img_data": cur, "seg_label": lab, "clipimgs_data": [prev], "cliplabels_data": []})
    loss.backward()
    res = {"train_loss": np.float64(loss.item()), "train_acc": np.float64(acc.item())}
    res.update(grads_summary(mod, full=("w0_1", "w1_0", "flowcnn.conv4.0.weight")))
    res["meta"] = np.array([arch, "netwarp", str(shape)])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "loss %.6f acc %.4f" % (loss.item(), acc.item()))


def raft_images(tag, shape):
    """Two frames in [0, 255]: smooth-ish random texture and a shifted, perturbed copy."""
    a = np.clip(det_input(tag + ":img1", shape) * 60.0 + 120.0, 0.0, 255.0).astype(np.float32)
    b = np.roll(a, (2, -3), axis=(2, 3)) + det_input(tag + ":noise", shape) * 4.0
    return a, np.clip(b, 0.0, 255.0).astype(np.float32)


def case_raft(tag="raft_basic", shape=(1, 3, 128, 192), iters=4):
    """The frozen flow network of NetWarp (models/netwarp.py:71-77,170-176): RAFT_core.raft.RAFT in eval mode with
    deterministic weights, test_mode=True (fp32; the reference's forward casts to float32 internally, raft.py:92-93,
    so there is no float64 re-run here - the oracle's own fp64-vs-fp32 gap is the noise estimate in the tests)."""
    from RAFT_core.corr import CorrBlock
    from RAFT_core.raft import RAFT

    torch.manual_seed(0)
    raft = RAFT()
    load_det(raft)
    raft.eval()
    a, b = raft_images(tag, shape)
    res = {}
    with torch.no_grad():
        ta, tb = torch.from_numpy(a), torch.from_numpy(b)
        for it in (1, iters):
            low, up = raft(ta, tb, iters=it, test_mode=True)
            res["flow_low_it%d" % it] = low.numpy()
            res["flow_up_it%d" % it] = up.numpy()
        i1, i2 = 2 * (ta / 255.0) - 1.0, 2 * (tb / 255.0) - 1.0
        f1, f2 = raft.fnet([i1, i2])
        res["fmap1"] = f1.numpy()
        res["cnet"] = raft.cnet(i1).numpy()
        n, _, h, w = f1.shape
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        coords = torch.stack([xs, ys], 0).float()[None].repeat(n, 1, 1, 1)
        res["corr0"] = CorrBlock(f1.float(), f2.float(), radius=4)(coords + 0.37).numpy()
    sd = raft.state_dict()
    res["sd_keys"] = np.array(list(sd.keys()))
    res["sd_shapes"] = np.array([",".join(str(int(d)) for d in v.shape) for v in sd.values()])
    res["meta"] = np.array([str(shape), str(iters)])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "flow_low |max| %.3f" % np.abs(res["flow_low_it%d" % iters]).max())


def case_datasets(tag="vspw_datasets"):
    """The reference's dataset classes (dataset2.py) on the tiny deterministic VSPW tree of oracle/det_data.py: which
    frames / flip / scale / crop the seeded draws select, and the tensors the model would be fed."""
    import random
    import tempfile

    import dataset2 as D
    from oracle.det_data import make_tiny_vspw

    root = tempfile.mkdtemp(prefix="vspw_tiny_")
    make_tiny_vspw(root)
    res = {}

    def put(key, out):
        imgs, labs = out[0], out[1]
        res[key + ":imgs"] = np.stack([t.numpy() for t in imgs])
        res[key + ":labs"] = np.stack([t.numpy() for t in labs])

    for ms in (False, True):
        a = args_ns(cropsize=40, dataroot=root, trainfps=1, clip_num=4, dilation2="3,6,9", multi_scale=ms,
                    lesslabel=False, dilation_num=0, method="clip_psp")
        ds = D.BaseDataset_longclip(a, "train")
        for seed in (0, 1, 2, 3, 4):
            np.random.seed(100 + seed)
            random.seed(200 + seed)
            idx = seed % len(ds)
            put("longclip:ms%d:seed%d" % (ms, seed), ds[idx])
    a = args_ns(cropsize=40, dataroot=root, trainfps=1, clip_num=2, dilation_num=0, multi_scale=True, lesslabel=False,
                method="netwarp")
    ds = D.BaseDataset_clip(a, "train")
    for seed in (0, 1, 2):
        np.random.seed(300 + seed)
        random.seed(400 + seed)
        put("clip:seed%d" % seed, ds[seed % len(ds)])
    a = args_ns(clip_num=4, dilation2="3,6,9", lesslabel=False, method="clip_psp")
    ts = D.TestDataset_longclip(root, "v_b", a, is_train=False)
    for index in (0, 7):
        img, tgt, cimgs, ctgts, name = ts[index]
        put("test_longclip:%d" % index, ([img] + cimgs, [tgt] + ctgts))
        res["test_longclip:%d:name" % index] = np.array(name)
    res["test_longclip:len"] = np.int64(len(ts))
    a = args_ns(clip_num=3, dilation_num=1, lesslabel=False, method="netwarp")
    tc = D.TestDataset_clip(root, "v_c", a, is_train=False)
    for index in (0, 9, 19):
        img, tgt, cimgs, ctgts, name = tc[index]
        put("test_clip:%d" % index, ([img] + cimgs, [tgt] + ctgts))
    a = args_ns(clip_num=3, dilation_num=1, lesslabel=False, method="nonlocal3d")
    tn = D.TestDataset_clip(root, "v_c", a, is_train=True)
    img, tgt, cimgs, ctgts, names = tn[1]
    put("test_clip_nl3d:1", ([img] + cimgs, [tgt] + ctgts))
    res["test_clip_nl3d:1:names"] = np.array(names)
    res["test_clip_nl3d:len"] = np.int64(len(tn))
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, len(res), "arrays")


def case_frame_dataset(tag="vspw_dataset_frame"):
    """The reference's PER-FRAME dataset (dataset2.py:494-650, BaseDataset: what train.py feeds the per-frame PSPNet /
    OCRNet of cfg 1-2) on the tiny VSPW tree: the frame list at `trainfps`, the seeded flip / scale / crop draws of the
    train split and the un-augmented val split."""
    import random
    import tempfile

    import dataset2 as D
    from oracle.det_data import make_tiny_vspw

    root = tempfile.mkdtemp(prefix="vspw_tiny_")
    make_tiny_vspw(root)
    res = {}
    for ms in (False, True):
        a = args_ns(cropsize=40, dataroot=root, trainfps=5, multi_scale=ms, lesslabel=False, train_filter=False)
        ds = D.BaseDataset(a, "train")
        res["train:len"] = np.int64(len(ds))
        res["train:list"] = np.array(["%s/%s" % vi for vi in ds.imglist])
        for seed in (0, 1, 2, 3, 4, 5):
            np.random.seed(500 + seed)
            random.seed(600 + seed)
            img, seg = ds[(7 * seed + 1) % len(ds)]
            res["train:ms%d:seed%d:img" % (ms, seed)] = img.numpy()
            res["train:ms%d:seed%d:seg" % (ms, seed)] = seg.numpy()
    a = args_ns(cropsize=40, dataroot=root, trainfps=5, multi_scale=True, lesslabel=False, train_filter=False)
    dv = D.BaseDataset(a, "val")
    res["val:len"] = np.int64(len(dv))
    res["val:list"] = np.array(["%s/%s" % vi for vi in dv.imglist])
    for index in (0, len(dv) - 1):
        img, seg = dv[index]
        res["val:%d:img" % index] = img.numpy()
        res["val:%d:seg" % index] = seg.numpy()
    # per-frame TEST dataset (dataset2.py:34-141, test.py's feed): every frame of one video, optional 720p resize
    a = args_ns(lesslabel=False, use_720p=False)
    ts = D.TestDataset(root, "v_b", a)
    res["test:len"] = np.int64(len(ts))
    for index in (0, len(ts) - 1):
        img, gt, name = ts[index]
        res["test:%d:img" % index], res["test:%d:seg" % index] = img.numpy(), gt.numpy()
        res["test:%d:name" % index] = np.array(name)
    a = args_ns(lesslabel=False, use_720p=True)
    img, gt, name = D.TestDataset(root, "v_b", a)[3]
    res["test720:3:shape"] = np.array(img.shape)
    res["test720:3:img_sub"] = img.numpy()[:, ::8, ::8].copy()   # (the full 3 x 720 x 1080 frame is 9 MB)
    res["test720:3:seg_sub"] = gt.numpy()[:, ::8, ::8].copy()
    res["test720:3:img_sum"] = np.float64(img.double().sum().item())
    res["test720:3:seg_sum"] = np.float64(gt.double().sum().item())
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, len(res), "arrays; train frames", int(res["train:len"]), "val frames", int(res["val:len"]))


def case_helpers(tag="helpers_reference"):
    """Host-side helpers the reference's test driver imports (test_clip2.py:16-18): utils.colorEncode / accuracy /
    intersectionAndUnion / unique / find_recursive and lib.utils.as_numpy, captured on seeded inputs."""
    import tempfile

    import collections
    import collections.abc
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_utils_py", os.path.join(REF, "utils.py"))  # (RAFT_core/utils
    U = importlib.util.module_from_spec(spec)                                                      # shadows the name)
    spec.loader.exec_module(U)
    for n in ("Sequence", "Mapping"):  # lib/utils/th.py predates their move to collections.abc
        if not hasattr(collections, n):
            setattr(collections, n, getattr(collections.abc, n))
    from lib.utils import as_numpy

    rs = np.random.RandomState(11)
    res = {}
    lab = rs.randint(-1, 9, size=(13, 17))
    colors = rs.randint(0, 256, size=(9, 3)).astype(np.uint8)
    res["labelmap"], res["colors"] = lab, colors
    res["colorEncode:RGB"] = U.colorEncode(lab.copy(), colors)
    res["colorEncode:BGR"] = np.ascontiguousarray(U.colorEncode(lab.copy(), colors, mode="BGR"))
    pred = rs.randint(0, 9, size=(13, 17))
    res["pred"] = pred
    acc, n = U.accuracy(pred, lab)
    res["accuracy"] = np.array([acc, float(n)])
    inter, union = U.intersectionAndUnion(pred, lab, 9)
    res["intersection"], res["union"] = inter, union
    u, idx, inv, cnt = U.unique(lab.copy(), True, True, True)
    res["unique"], res["unique_index"], res["unique_inverse"], res["unique_counts"] = u, idx, inv, cnt
    root = tempfile.mkdtemp(prefix="find_")
    for rel in ("a/x.jpg", "a/.hidden.jpg", "a/b/y.jpg", "a/b/z.png", "c/w.jpg"):
        os.makedirs(os.path.join(root, os.path.dirname(rel)), exist_ok=True)
        open(os.path.join(root, rel), "w").close()
    res["find_recursive:jpg"] = np.array(sorted(os.path.relpath(f, root) for f in U.find_recursive(root)))
    res["find_recursive:png"] = np.array(sorted(os.path.relpath(f, root) for f in U.find_recursive(root, ext=".png")))
    nested = as_numpy({"a": [torch.arange(3), (torch.ones(2, 2), 5)], "b": torch.tensor(2.5)})
    res["as_numpy:a0"], res["as_numpy:a1_0"], res["as_numpy:a1_1"], res["as_numpy:b"] = (
        nested["a"][0], nested["a"][1][0], nested["a"][1][1], nested["b"])
    res["as_numpy:types"] = np.array([type(nested).__name__, type(nested["a"]).__name__, type(nested["a"][1]).__name__])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, len(res), "arrays")


def case_ops(M, tag="ops_reference"):
    """Op-level vectors straight from the reference's own helper functions."""
    import models.netwarp as ref_nw
    from models.ocr_modules.spatial_ocr_block import SpatialGather_Module

    res = {}
    x = torch.from_numpy(det_input(tag + ":warp_x", (2, 6, 9, 11)))
    flo = torch.from_numpy(det_input(tag + ":warp_f", (2, 2, 9, 11), scale=1.9)) - 0.7
    flo[0, :, 0, 0] = torch.tensor([-30.0, 25.0])
    res["flowwarp_flow"] = flo.numpy().copy()
    res["flowwarp"] = ref_nw.flowwarp(x, flo).numpy()
    feats = torch.from_numpy(det_input(tag + ":g_feats", (2, 32, 9, 13)))
    probs = torch.from_numpy(det_input(tag + ":g_probs", (2, 7, 9, 13), scale=2.0))
    res["ocr_gather"] = SpatialGather_Module(7)(feats, probs).numpy()
    pa_pred = torch.from_numpy(det_input(tag + ":pa", (2, 5, 6, 7)))
    pa_lab = torch.from_numpy(det_labels(tag + ":pa", (2, 1, 6, 7), 5)).squeeze(1).long()
    res["pixel_acc"] = np.float64(M.models.SegmentationModuleBase().pixel_acc(pa_pred, pa_lab).item())
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "done")


def case_keys(M, tag="state_keys"):
    """state_dict key names / shapes and the 4 SGD parameter-group listings of the reference's modules
    (the drop-in contract of SURVEY.md 8b), for the CPU host-logic tests."""
    import models.netwarp as ref_nw

    res = {}
    crit = torch.nn.NLLLoss(ignore_index=255)

    def record(name, mod, groups=True):
        sd = mod.state_dict()
        res[name + ":keys"] = np.array(list(sd.keys()))
        res[name + ":shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        if groups:
            ids = {id(p): k for k, p in mod.named_parameters()}
            for g in ("get_1x_lr_params", "get_10x_lr_params", "get_1x_lr_params_bias", "get_10x_lr_params_bias"):
                res["%s:%s" % (name, g)] = np.array([ids[id(p)] for p in getattr(mod, g)()])

    for arch in ("resnet50dilated", "resnet101dilated"):
        enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=2048)
        record("clip_psp:" + arch, M.Clip_PSP(enc, crit, args_ns(), deep_sup_scale=0.4))
        enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=2048)
        record("clip_ocr:" + arch, M.ClipOCRNet(enc, crit, args_ns(), deep_sup_scale=0.4))
    enc = M.ModelBuilder.build_encoder(arch="resnet50dilated", fc_dim=2048)
    record("clip_psp_pspw:resnet50dilated", M.Clip_PSP(enc, crit, args_ns(psp_weight=True), deep_sup_scale=0.4))
    enc = M.ModelBuilder.build_encoder(arch="resnet50dilated", fc_dim=2048)
    record("nonlocal3d:resnet50dilated", M.Non_local3d(args_ns(), enc, crit))
    for arch, dec, fc in (("resnet18dilated", "ppm_deepsup", 512), ("resnet101dilated", "ppm_deepsup", 2048),
                          ("resnet50dilated", "ocrnet_deepsup", 2048), ("resnet50dilated", "nonlocal2d", 2048),
                          ("resnet50dilated", "ppm_deepsup_clip", 2048), ("resnet50dilated", "ppm", 2048),
                          ("resnet50", "ppm_clip", 2048)):
        enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=fc)
        d = M.ModelBuilder.build_decoder(arch=dec, fc_dim=fc, num_class=K)
        record("seg:%s:%s" % (arch, dec), M.SegmentationModule(enc, d, crit, 0.4), groups=False)
    orig_raft, orig_load = ref_nw.RAFT, torch.load
    ref_nw.RAFT = lambda: torch.nn.Identity()
    torch.load = lambda *a, **k: {}
    try:
        enc = M.ModelBuilder.build_encoder(arch="resnet50dilated", fc_dim=2048)
        d = M.ModelBuilder.build_decoder(arch="ppm_deepsup_clip", fc_dim=2048, num_class=K)
        nw = M.NetWarp(enc, d, crit, args_ns(clip_num=2), deep_sup_scale=0.4)
    finally:
        ref_nw.RAFT, torch.load = orig_raft, orig_load
    import models.netwarp_ocr as ref_nwo

    orig_raft2 = ref_nwo.RAFT
    ref_nwo.RAFT = lambda: torch.nn.Identity()
    torch.load = lambda *a, **k: {}
    try:
        enc = M.ModelBuilder.build_encoder(arch="resnet50dilated", fc_dim=2048)
        nwo = M.NetWarp_ocr(enc, crit, args_ns(clip_num=2), deep_sup_scale=0.4)
    finally:
        ref_nwo.RAFT, torch.load = orig_raft2, orig_load
    for name, m_ in (("netwarp:resnet50dilated", nw), ("netwarp_ocr:resnet50dilated", nwo)):
        sd = m_.state_dict()
        res[name + ":keys"] = np.array([k for k in sd if not k.startswith("raft.")])
        res[name + ":shapes"] = np.array([str(tuple(v.shape)) for k, v in sd.items() if not k.startswith("raft.")])
        ids = {id(p): k for k, p in m_.named_parameters()}
        for g in ("get_1x_lr_params", "get_10x_lr_params", "get_1x_lr_params_bias", "get_10x_lr_params_bias"):
            res["%s:%s" % (name, g)] = np.array([ids[id(p)] for p in getattr(m_, g)()])
    # BN init of build_decoder(weights_init) and default conv geometry after _nostride_dilate
    enc = M.ModelBuilder.build_encoder(arch="resnet101dilated", fc_dim=2048)
    geo = []
    for k, m in enc.named_modules():
        if isinstance(m, torch.nn.Conv2d):
            geo.append("%s %s %s %s %s" % (k, m.kernel_size, m.stride, m.padding, m.dilation))
    res["geometry:resnet101dilated"] = np.array(geo)
    enc = M.ModelBuilder.build_encoder(arch="resnet18dilated", fc_dim=512)
    geo = []
    for k, m in enc.named_modules():
        if isinstance(m, torch.nn.Conv2d):
            geo.append("%s %s %s %s %s" % (k, m.kernel_size, m.stride, m.padding, m.dilation))
    res["geometry:resnet18dilated"] = np.array(geo)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "done")


class _YacsNode(dict):
    """Minimal stand-in for yacs.config.CfgNode (yacs is not installed here): attribute access + the two merge calls the
    reference's drivers make.  Only used to EXECUTE the reference's config/defaults.py and driver files."""

    def __init__(self, init=None, **kw):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def merge_from_file(self, f):
        import yaml

        def rec(dst, src):
            for k, v in src.items():
                if isinstance(v, dict):
                    rec(dst[k], v)
                else:
                    dst[k] = tuple(v) if isinstance(v, list) else v

        rec(self, yaml.safe_load(open(f)))

    def merge_from_list(self, lst):
        pass

    def clone(self):
        import copy

        return copy.deepcopy(self)


def import_reference_drivers():
    """train_clip2 / test_clip2 of the reference as modules (functions only; their __main__ blocks are not run)."""
    import importlib.util

    y, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")
    yc.CfgNode = _YacsNode
    y.config = yc
    sys.modules["yacs"], sys.modules["yacs.config"] = y, yc
    # `utils` already resolves to RAFT_core/utils (a package) for the RAFT imports: bind the reference's utils.py
    spec = importlib.util.spec_from_file_location("utils", os.path.join(REF, "utils.py"))
    um = importlib.util.module_from_spec(spec)
    sys.modules["utils"] = um
    spec.loader.exec_module(um)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import test_clip2 as ref_test  # noqa
        import train_clip2 as ref_train  # noqa
    finally:
        os.chdir(cwd)
    return ref_train, ref_test, um


def _parser_defaults(path):
    """Run the `__main__` block of a reference driver up to parse_args() and return {flag: (default, type, choices)}.
    parse_args is replaced by a hook that records the parser and aborts, so nothing after it executes."""
    import argparse
    import runpy

    class _Stop(Exception):
        pass

    got = {}
    orig = argparse.ArgumentParser.parse_args

    def hook(self, *a, **k):
        for act in self._actions:
            if act.dest == "help":
                continue
            tname = getattr(act.type, "__name__", str(act.type)) if act.type is not None else "None"
            got[act.dest] = (repr(act.default), tname, repr(list(act.choices)) if act.choices else "None",
                             "opt" if act.option_strings else "pos")
        raise _Stop()

    argparse.ArgumentParser.parse_args = hook
    cwd, argv = os.getcwd(), sys.argv
    os.chdir(REF)
    sys.argv = [path]
    try:
        runpy.run_path(os.path.join(REF, path), run_name="__main__")
    except _Stop:
        pass
    finally:
        argparse.ArgumentParser.parse_args = orig
        os.chdir(cwd)
        sys.argv = argv
    return got


def case_frame_drivers(M, tag="frame_drivers_reference"):
    """Host-side definitions of the PER-FRAME drivers (train.py / test.py: the entry points of cfg 1-2,
    scripts/run_psp.sh) captured from the reference: argparse flags and defaults, group_weight's decay / no-decay
    partition (train.py:191-211), the two SGD optimizers (:214-226) and the poly schedule (:229-238)."""
    import_reference_drivers()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import train as ref_train  # noqa
    finally:
        os.chdir(cwd)
    res = {}
    for drv in ("train.py", "test.py"):
        d = _parser_defaults(drv)
        keys = sorted(d)
        res["argparse:%s:dest" % drv] = np.array(keys)
        res["argparse:%s:default" % drv] = np.array([d[k][0] for k in keys])
        res["argparse:%s:type" % drv] = np.array([d[k][1] for k in keys])
        res["argparse:%s:kind" % drv] = np.array([d[k][3] for k in keys])
    enc = M.ModelBuilder.build_encoder(arch="resnet18dilated", fc_dim=512)
    dec = M.ModelBuilder.build_decoder(arch="ppm_deepsup", fc_dim=512, num_class=K)
    for name, net in (("encoder", enc), ("decoder", dec)):
        names = {id(p): n for n, p in net.named_parameters()}
        groups = ref_train.group_weight(net)
        res["group_weight:%s:decay" % name] = np.array([names[id(p)] for p in groups[0]["params"]])
        res["group_weight:%s:no_decay" % name] = np.array([names[id(p)] for p in groups[1]["params"]])
        res["group_weight:%s:no_decay_wd" % name] = np.float64(groups[1]["weight_decay"])
    cfg = ref_train.cfg
    cfg.TRAIN.lr_encoder, cfg.TRAIN.lr_decoder, cfg.TRAIN.weight_decay = 0.002, 0.004, 1e-4
    opts = ref_train.create_optimizers((enc, dec, None), cfg)
    for name, opt in zip(("encoder", "decoder"), opts):
        res["opt:%s:group_sizes" % name] = np.array([len(g["params"]) for g in opt.param_groups])
        res["opt:%s:group_wd" % name] = np.array([g["weight_decay"] for g in opt.param_groups], dtype=np.float64)
        res["opt:%s:group_lr0" % name] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
        res["opt:%s:momentum" % name] = np.float64(opt.param_groups[0]["momentum"])
    max_iters, iters = 900, [0, 1, 13, 450, 899]
    trace = []
    for it in iters:
        ref_train.adjust_learning_rate(opts, it, cfg, max_iters)
        trace.append([opts[0].param_groups[0]["lr"], opts[0].param_groups[1]["lr"], opts[1].param_groups[0]["lr"],
                      opts[1].param_groups[1]["lr"], cfg.TRAIN.running_lr_encoder, cfg.TRAIN.running_lr_decoder])
    res["lr:iters"], res["lr:max_iters"] = np.array(iters), np.int64(max_iters)
    res["lr:trace"] = np.array(trace, dtype=np.float64)
    res["lr:lr_pow"] = np.float64(cfg.TRAIN.lr_pow)
    res["cfg:beta1"] = np.float64(cfg.TRAIN.beta1)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, len(res), "arrays")


def _function_from_script(path, name):
    """One top-level function of a reference SCRIPT (a file whose module body runs a whole job, so it cannot be imported):
    its definition is compiled from the parsed file and executed in a fresh namespace, nothing else of the file runs."""
    import ast

    tree = ast.parse(open(os.path.join(REF, path)).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(fn) == 1, (path, name)
    ns = {"torch": torch, "nn": torch.nn, "np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns[name]


def case_metric_tools(tag="metric_tools_reference"):
    """The evaluation-side scripts of the reference: TC_cal.py's nearest flow-warp (TC_cal.py:12-38) on label maps with
    sub-pixel, half-integer (ties) and out-of-image displacements; VC_perclip.py's get_common (:7-24); the 480p target
    size of change2_480p.py:17."""
    res = {}
    warp = _function_from_script("TC_cal.py", "flowwarp")
    rs = np.random.RandomState(21)
    for name, (h, w) in (("a", (23, 31)), ("b", (48, 64)), ("c", (17, 16))):
        lab = rs.randint(0, 124, size=(2, 1, h, w)).astype(np.float32)
        flo = (rs.randn(2, 2, h, w) * 3.0).astype(np.float32)
        flo[0, :, :4] = np.round(flo[0, :, :4] * 2.0) / 2.0     # exact halves: nearbyint ties
        flo[1, 0, -3:] += w                                     # far outside the image
        flo[1, 1, :, :2] -= h
        res["warp:%s:lab" % name], res["warp:%s:flow" % name] = lab, flo
        res["warp:%s:out" % name] = warp(torch.from_numpy(lab), torch.from_numpy(flo)).numpy()
    vc = _function_from_script("VC_perclip.py", "get_common")
    h, w = 13, 17
    gts = [rs.randint(0, 4, (h, w)) for _ in range(12)]
    for t in range(1, 12):
        keep = rs.rand(h, w) < 0.8
        gts[t][keep] = gts[t - 1][keep]
    preds = [np.where(rs.rand(h, w) < 0.85, g, rs.randint(0, 4, (h, w))) for g in gts]
    res["vc:gt"], res["vc:pred"] = np.stack(gts), np.stack(preds)
    for cn in (2, 5):
        res["vc:accs%d" % cn] = np.array(vc(gts, preds, cn, h, w), dtype=np.float64)
    sizes = [(1920, 1080), (1280, 720), (853, 480), (640, 360), (1000, 563), (720, 1280)]
    res["size480:in"] = np.array(sizes)
    res["size480:out"] = np.array([int(480 * ww / hh) for ww, hh in sizes])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, len(res), "arrays")


def case_frame_trajectory(M, tag="frame_train_trajectory"):
    """Five optimisation steps of the reference's per-frame training loop (train.py:58-92 with its own create_optimizers /
    adjust_learning_rate / SegmentationModule, resnet18dilated + ppm_deepsup, B = 2 frames of 65 x 65, Dropout2d off) in
    float32 and float64: loss / accuracy per step, every parameter's norm and the momentum-buffer norms at the end.  Pins
    model + loss + two SGDs (momentum, decay / no-decay groups) + poly schedule JOINTLY over several updates."""
    import_reference_drivers()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import train as ref_train  # noqa
    finally:
        os.chdir(cwd)
    res = {}
    steps, max_iters = 5, 12
    # "f32" / "f64": the run itself; "p0".."p5": float32 runs whose first image carries a relative perturbation of 1e-7
    # (one float32 ulp): how far rounding-sized differences carry a float32 trajectory of THIS loop - the yardstick
    runs = [(torch.float32, "f32", None), (torch.float64, "f64", None)] + [(torch.float32, "p%d" % i, i) for i in range(6)]
    for dt, name, pert in runs:
        torch.manual_seed(0)
        enc = M.ModelBuilder.build_encoder(arch="resnet18dilated", fc_dim=512)
        dec = M.ModelBuilder.build_decoder(arch="ppm_deepsup", fc_dim=512, num_class=K)
        mod = M.SegmentationModule(enc, dec, torch.nn.NLLLoss(ignore_index=255), 0.4)
        load_det(mod)
        zero_dropout(mod)
        mod.to(dt).train()
        cfg = ref_train.cfg
        cfg.TRAIN.lr_encoder = cfg.TRAIN.lr_decoder = 0.002
        cfg.TRAIN.weight_decay = 1e-4
        opts = ref_train.create_optimizers((enc, dec, None), cfg)
        losses, accs = [], []
        for it in range(steps):
            x = det_input("%s:img:%d" % (tag, it), (2, 3, 65, 65))
            if pert is not None and it == 0:
                x = x * (1.0 + 1e-7 * np.random.RandomState(900 + pert).randn(*x.shape)).astype(np.float32)
            img = torch.from_numpy(x).to(dt)
            lab = torch.from_numpy(det_labels("%s:lab:%d" % (tag, it), (2, 1, 65, 65), K))
            mod.zero_grad()
            ref_train.adjust_learning_rate(opts, it, cfg, max_iters)
            loss, acc = mod({"img_data": img, "seg_label": lab})
            loss = loss.mean()
            loss.backward()
            if it == 0 and pert is None:  # the first step's gradients: norms, and the values at 64 fixed positions each
                g0 = [(k, p.grad.detach().double()) for k, p in mod.named_parameters()]
                res[name + ":grad0_norms"] = np.array([float(g.norm()) for _, g in g0])
                res[name + ":grad0_samples"] = np.stack([g.flatten()[det_sample_index(k, g.numel(), 64)].numpy()
                                                         for k, g in g0])
            for o in opts:
                o.step()
            losses.append(float(loss.detach()))
            accs.append(float(acc.mean()))
        res[name + ":loss"], res[name + ":acc"] = np.array(losses), np.array(accs)
        names = [k for k, _ in mod.named_parameters()]
        res["param_names"] = np.array(names)
        res[name + ":param_norms"] = np.array([float(p.detach().double().norm()) for _, p in mod.named_parameters()])
        res[name + ":momentum_norms"] = np.array([float(o.state[p]["momentum_buffer"].double().norm())
                                                  for o in opts for g in o.param_groups for p in g["params"]])
        if pert is None:
            bn = mod.encoder.layer4[1].bn2
            res[name + ":running_mean"] = bn.running_mean.double().numpy()
            res[name + ":running_var"] = bn.running_var.double().numpy()
    res["meta"] = np.array([steps, max_iters])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    dev = np.stack([np.abs(res["p%d:loss" % i] - res["f64:loss"]) for i in range(6)])
    print(tag, "loss f64", res["f64:loss"], "|f32 - f64|", np.abs(res["f32:loss"] - res["f64:loss"]),
          "perturbed float32 runs: max |. - f64| per step", dev.max(0))


def case_drivers(M, tag="drivers_reference"):
    """The host-side row of SURVEY.md 8(f)-2 pinned on the reference itself: Evaluator (utils.py:55-107), get_common
    (utils.py:37-53), parse_devices, create_optimizers / adjust_learning_rate (train_clip2.py:215-252), the argparse
    flags and defaults of both drivers (train_clip2.py:404-490, test_clip2.py:349-406), config/defaults.py and the
    palette of test_clip2.py:25."""
    ref_train, ref_test, ref_utils = import_reference_drivers()
    res = {}
    rng = np.random.RandomState(304)
    # ---- config defaults (captured first: the optimizer section below writes schedule fields into cfg)
    # ---- config defaults (config/defaults.py executed with the CfgNode stand-in) and the 18 yaml files
    flat = []

    def rec(prefix, node):
        for k in node:
            if isinstance(node[k], dict):
                rec(prefix + k + ".", node[k])
            else:
                flat.append((prefix + k, repr(node[k])))

    import importlib

    cfgmod = importlib.import_module("config.defaults")
    rec("", cfgmod._C)
    res["cfg:keys"] = np.array([k for k, _ in flat])
    res["cfg:values"] = np.array([v for _, v in flat])

    # ---- Evaluator: 124 classes, several absent from the ground truth, 255 = ignore, predictions over all classes
    for name, ncls, present in (("ev124", 124, [0, 1, 2, 5, 8, 13, 21, 34, 55, 89, 123]), ("ev7", 7, [0, 2, 3, 6])):
        ev = ref_utils.Evaluator(ncls)
        gts, prs = [], []
        for b in range(3):
            gt = rng.choice(present + [255], size=(2, 37, 41)).astype(np.float32)
            pr = np.where(rng.rand(2, 37, 41) < 0.6, np.minimum(gt, ncls - 1), rng.randint(0, ncls, (2, 37, 41)))
            pr = pr.astype(np.int64)
            gts.append(gt)
            prs.append(pr)
            ev.add_batch(gt, pr)
        res[name + ":gt"] = np.stack(gts)
        res[name + ":pred"] = np.stack(prs)
        res[name + ":cm"] = ev.confusion_matrix.copy()
        with np.errstate(divide="ignore", invalid="ignore"):
            res[name + ":metrics"] = np.array([ev.Pixel_Accuracy(), ev.Pixel_Accuracy_Class(),
                                               ev.Mean_Intersection_over_Union(),
                                               ev.Frequency_Weighted_Intersection_over_Union()], dtype=np.float64)
            ev.beforeval()
            res[name + ":cm_beforeval"] = ev.confusion_matrix.copy()
            res[name + ":metrics_beforeval"] = np.array([ev.Pixel_Accuracy(), ev.Pixel_Accuracy_Class(),
                                                         ev.Mean_Intersection_over_Union(),
                                                         ev.Frequency_Weighted_Intersection_over_Union()],
                                                        dtype=np.float64)
        ev.reset()
        res[name + ":cm_reset_sum"] = np.float64(ev.confusion_matrix.sum())
    # ---- video consistency
    h, w = 19, 23
    base_gt = rng.randint(0, 5, (h, w))
    gl, pl = [], []
    for t in range(14):
        g = base_gt.copy()
        flip = rng.rand(h, w) < 0.05 * (t % 4)
        g[flip] = rng.randint(0, 5, flip.sum())
        p = g.copy()
        noise = rng.rand(h, w) < 0.15
        p[noise] = rng.randint(0, 5, noise.sum())
        gl.append(g)
        pl.append(p)
    res["vc:gt"] = np.stack(gl)
    res["vc:pred"] = np.stack(pl)
    for cn in (2, 4, 8):
        res["vc:accs%d" % cn] = np.array(ref_utils.get_common(gl, pl, cn, h, w), dtype=np.float64)
    # ---- parse_devices
    devs = ["0-3", "0,1,2,3", "gpu0-gpu2", "2", "0,2-3", "3-1"]
    res["parse_devices:in"] = np.array(devs)
    res["parse_devices:out"] = np.array([",".join(ref_utils.parse_devices(d)) for d in devs])
    # ---- optimizer groups + poly schedule on the reference's Clip_PSP
    enc = M.ModelBuilder.build_encoder(arch="resnet50dilated", fc_dim=2048)
    mod = M.Clip_PSP(enc, torch.nn.NLLLoss(ignore_index=255), args_ns(), deep_sup_scale=0.4)
    cfg = ref_train.cfg
    for fix in (False, True):
        a = types.SimpleNamespace(lr=0.002, fix=fix)
        cfg.TRAIN.weight_decay = 1e-4
        opt = ref_train.create_optimizers(mod, cfg, a)
        res["opt:fix%d:group_sizes" % fix] = np.array([len(g["params"]) for g in opt.param_groups])
        res["opt:fix%d:group_wd" % fix] = np.array([g["weight_decay"] for g in opt.param_groups], dtype=np.float64)
        res["opt:fix%d:group_lr0" % fix] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
        res["opt:fix%d:momentum" % fix] = np.float64(opt.param_groups[0]["momentum"])
        max_iters = 1200
        iters = [0, 1, 7, 300, 599, 1100, 1199]
        trace, running = [], []
        for it in iters:
            ref_train.adjust_learning_rate(opt, it, cfg, max_iters, a)
            trace.append([g["lr"] for g in opt.param_groups])
            running.append(cfg.TRAIN.running_lr_encoder)
        res["lr:fix%d:iters" % fix] = np.array(iters)
        res["lr:fix%d:max_iters" % fix] = np.int64(max_iters)
        res["lr:fix%d:trace" % fix] = np.array(trace, dtype=np.float64)
        res["lr:fix%d:running_lr_encoder" % fix] = np.array(running, dtype=np.float64)
    # ---- argparse surfaces
    for drv in ("train_clip2.py", "test_clip2.py"):
        d = _parser_defaults(drv)
        keys = sorted(d)
        res["argparse:%s:dest" % drv] = np.array(keys)
        res["argparse:%s:default" % drv] = np.array([d[k][0] for k in keys])
        res["argparse:%s:type" % drv] = np.array([d[k][1] for k in keys])
        res["argparse:%s:choices" % drv] = np.array([d[k][2] for k in keys])
        res["argparse:%s:kind" % drv] = np.array([d[k][3] for k in keys])
    import yaml

    ynames, yflat = [], []
    cdir = os.path.join(REF, "config")
    for f in sorted(os.listdir(cdir)):
        if f.endswith(".yaml"):
            y = yaml.safe_load(open(os.path.join(cdir, f)))
            ynames.append(f)
            yflat.append(repr(sorted((sec + "." + k, repr(v)) for sec, d in y.items() if isinstance(d, dict)
                                     for k, v in d.items()) + sorted((k, repr(v)) for k, v in y.items()
                                                                     if not isinstance(v, dict))))
    res["yaml:names"] = np.array(ynames)
    res["yaml:flat"] = np.array(yflat)
    res["palette"] = np.array(ref_test._palette, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, len(res), "arrays")


FIXBN_FULL = ("encoder.conv1.weight", "encoder.layer1.0.conv2.weight", "encoder.layer2.0.downsample.0.weight",
              "encoder.layer3.5.conv2.bias")


def case_fixbn(M, method, arch, tag, T=3, train_shape=(2, 3, 65, 65)):
    """Training step with frozen BatchNorm (cfg.TRAIN.fix_bn: train_clip2.py `segmentation_module.train(not
    cfg.TRAIN.fix_bn)`, config/defaults.py:71): the module is in eval mode - running statistics, no dropout - while the
    loss and its gradients are computed.  Without batch statistics the gradients are smooth functions of the weights,
    so they can be gated element-wise; stored in fp32 and from the float64 re-run."""
    torch.manual_seed(0)
    enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=2048)
    crit = torch.nn.NLLLoss(ignore_index=255)
    cls = {"clip_psp": M.Clip_PSP, "clip_ocr": M.ClipOCRNet}[method]
    mod = cls(enc, crit, args_ns(), deep_sup_scale=0.4)
    load_det(mod)
    imgs = [torch.from_numpy(det_input("%s:train:%d" % (tag, t), train_shape)) for t in range(T)]
    labs = [torch.from_numpy(det_labels("%s:train:%d" % (tag, t), (train_shape[0], 1) + train_shape[2:], K))
            for t in range(T)]

    def feed(cast=lambda x: x):
        return {"img_data": cast(imgs[-1]), "seg_label": cast(labs[-1]), "clipimgs_data": [cast(i) for i in imgs[:-1]],
                "cliplabels_data": [cast(l) for l in labs[:-1]]}

    res = calibrate_bn(mod, lambda: mod(feed()))
    calibrated_sd = {k: v.clone() for k, v in mod.state_dict().items()}
    bn_params = [k for k, _ in mod.named_parameters() if ".bn" in k or ".downsample.1." in k]
    full = tuple(FIXBN_FULL) + tuple(bn_params)

    def run(store_suffix):
        mod.eval()  # = segmentation_module.train(not fix_bn) with fix_bn True
        mod.zero_grad()
        loss, acc = mod(feed((lambda x: x.double()) if store_suffix else (lambda x: x)))
        loss.backward()
        res["train_loss" + store_suffix] = np.float64(loss.item())
        res["train_acc" + store_suffix] = np.float64(acc.item())
        names, norms = [], []
        for k, p in mod.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            norms.append(float(p.grad.double().norm()))
            if k in full:
                res["grad%s:%s" % (store_suffix, k)] = p.grad.detach().float().numpy().copy()  # fp64 run stored as fp32
        res["grad_names"] = np.array(names)
        res["grad_norms" + store_suffix] = np.array(norms, dtype=np.float64)

    run("")
    rm_before = {k: v.clone() for k, v in mod.state_dict().items() if k.endswith("running_mean")}
    assert all(torch.equal(v, calibrated_sd[k]) for k, v in rm_before.items()), "frozen BN must not update statistics"
    mod.double()
    mod.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in calibrated_sd.items()})
    run("64")
    mod.float()
    res["meta"] = np.array([arch, method, str(T), str(train_shape), "fix_bn"])
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **res)
    print(tag, "loss %.6f (fp64 %.6f)" % (float(res["train_loss"]), float(res["train_loss64"])))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    M = import_reference()
    only = set(sys.argv[1:])

    def want(t):
        return not only or t in only

    if want("state_keys"):
        case_keys(M)
    if want("ops_reference"):
        case_ops(M)
    if want("r18_ppm_deepsup"):
        case_segmodule(M, "resnet18dilated", "ppm_deepsup", "r18_ppm_deepsup", fc_dim=512)
    if want("r50_clip_psp"):
        case_clip(M, "clip_psp", "resnet50dilated", "r50_clip_psp")
    if want("r50_clip_ocr"):
        case_clip(M, "clip_ocr", "resnet50dilated", "r50_clip_ocr")
    if want("r50_ocrnet_deepsup"):
        case_segmodule(M, "resnet50dilated", "ocrnet_deepsup", "r50_ocrnet_deepsup", fc_dim=2048)
    if want("r50_nonlocal2d"):
        case_segmodule(M, "resnet50dilated", "nonlocal2d", "r50_nonlocal2d", fc_dim=2048)
    if want("r50_nonlocal3d"):
        case_nonlocal3d(M, "resnet50dilated", "r50_nonlocal3d")
    if want("r50_nonlocal_downsample"):
        case_nonlocal_downsample(M, "resnet50dilated", "r50_nonlocal_downsample")
    if want("r50_netwarp"):
        case_netwarp(M, "resnet50dilated", "r50_netwarp")
    if want("r50_netwarp_ocr"):
        case_netwarp_ocr(M, "resnet50dilated", "r50_netwarp_ocr")
    if want("r50_clip_psp_pspw"):
        case_clip(M, "clip_psp", "resnet50dilated", "r50_clip_psp_pspw", psp_weight=True)
    if want("r50_clip_ocr_memory"):
        case_ocr_memory(M, "resnet50dilated", "r50_clip_ocr_memory")
    if want("raft_basic"):
        case_raft()
    if want("vspw_datasets"):
        case_datasets()
    if want("vspw_dataset_frame"):
        case_frame_dataset()
    if want("helpers_reference"):
        case_helpers()
    if want("r50_clip_psp_fixbn"):
        case_fixbn(M, "clip_psp", "resnet50dilated", "r50_clip_psp_fixbn")
    if want("r50_clip_ocr_fixbn"):
        case_fixbn(M, "clip_ocr", "resnet50dilated", "r50_clip_ocr_fixbn")
    if want("drivers_reference"):
        case_drivers(M)
    if want("frame_drivers_reference"):
        case_frame_drivers(M)
    if want("metric_tools_reference"):
        case_metric_tools()
    if want("frame_train_trajectory"):
        case_frame_trajectory(M)


if __name__ == "__main__":
    main()
