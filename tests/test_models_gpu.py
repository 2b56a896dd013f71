"""End-to-end parity of the HIP-backed modules (through the C ABI) against
  (1) the golden vectors produced by the reference itself (tests/golden/*.npz), and
  (2) the numpy oracle run live on the same seeded inputs.
Tolerances (north_star): eval logits within 1e-3 (where the reference's own fp32 result is further than that from its
float64 re-run: within 2x that measured error of the float64 logits — helpers.logit_tol / logit_error), arg-max identical wherever the reference's top-2 logit gap exceeds 2*tol; training loss
within 2e-4 relative, per-parameter gradient norms within 8e-2 relative and 4e-2 in aggregate (fp32 gradients through ~50 train-mode BN
layers on 9x9 maps are rounding-noisy: the oracle itself agrees with the reference only to that level in fp32 while
agreeing to 1e-6 in float64, see tests/test_oracle_golden.py; on the GPU the fp32 MFMA k-sequential accumulation
gives ~1e-5 forward differences after a few train-mode BN layers over 162-sample populations, enough to flip a
single ReLU decision among 83k activations, which alone moves every upstream gradient norm coherently by ~2%).
Those loose gates against the reference's float32 fixtures are kept as they were; what they cannot discriminate is
checked by test_training_gradients_with_pinned_decisions below: with the HIP forward's ReLU / max-pool decisions
injected into the float64 oracle every parameter's gradient is compared in relative L2, held to 1.5x the float32
oracle's own error under the same procedure (tests/test_fullsize_gpu.py explains the method)."""
import numpy as np
import pytest
import torch

from helpers import (K, build, check_argmax, check_grad_norms, clip_inputs, golden, hip_decision_store, load_det,
                     logit_error, logit_tol, pinned_gradient_errors, seg_inputs, zero_dropout)
from oracle.det_init import det_input, det_labels

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _grads(mod):
    return {k: p.grad.detach().float().cpu().numpy() for k, p in mod.named_parameters() if p.grad is not None}


def _hook(mod, store):
    return mod.register_forward_hook(lambda m, i, o: store.__setitem__("logits", o.detach().float().cpu().numpy()))


def _check_eval(fx, probs, store):
    tol = logit_tol(fx)
    probs = probs.detach().float().cpu().numpy()
    if "eval_logits" in fx.files:
        err = logit_error(fx, store["logits"])
        assert err < tol, "logits err %.3e (tol %.1e)" % (err, tol)
    assert np.abs(probs[:, :, ::4, ::4] - fx["eval_probs_sub"]).max() < tol
    flips, near = check_argmax(probs.argmax(1), fx, tol)
    assert flips <= near


def _check_train(fx, mod, loss, acc, tag):
    assert abs(loss.item() - float(fx["train_loss"])) < 2e-4 * abs(float(fx["train_loss"])), (loss.item(),
                                                                                             float(fx["train_loss"]))
    assert abs(acc.item() - float(fx["train_acc"])) < 2e-3
    g = _grads(mod)
    check_grad_norms(g, fx, 8e-2, tag, agg_rtol=4e-2)
    for key in fx.files:
        if key.startswith("grad:"):
            ref = fx[key]
            assert np.linalg.norm(g[key[5:]] - ref) <= 0.1 * np.linalg.norm(ref), key


@pytest.mark.parametrize("tag,arch,decoder,fc_dim", [
    ("r18_ppm_deepsup", "resnet18dilated", "ppm_deepsup", 512),
    ("r50_ocrnet_deepsup", "resnet50dilated", "ocrnet_deepsup", 2048),
    ("r50_nonlocal2d", "resnet50dilated", "nonlocal2d", 2048),
])
def test_per_frame_segmentation_module(dev, tag, arch, decoder, fc_dim):
    fx = golden(tag)
    mod = build("seg", arch, decoder, fc_dim, deep_sup_scale=None if decoder == "nonlocal2d" else 0.4)
    load_det(mod, fx=fx)
    zero_dropout(mod)
    mod.to(dev)
    inp = seg_inputs(tag)
    mod.eval()
    store = {}
    last = {"ppm_deepsup": "conv_last_", "ocrnet_deepsup": "head", "nonlocal2d": "last_layer"}[decoder]
    h = _hook(getattr(mod.decoder, last), store)
    with torch.no_grad():
        probs = mod({"img_data": _t(inp["eval_img"], dev), "seg_label": torch.zeros(1, 1, 64, 96, device=dev)},
                    segSize=(64, 96))
    h.remove()
    assert tuple(probs.shape) == (1, K, 64, 96)
    _check_eval(fx, probs, store)
    mod.train()
    loss, acc = mod({"img_data": _t(inp["train_img"], dev), "seg_label": _t(inp["train_lab"], dev)})
    loss.backward()
    _check_train(fx, mod, loss, acc, tag)
    assert np.abs(mod.encoder.bn1.running_mean.cpu().numpy() - fx["bn_running_mean:encoder.bn1"]).max() < 1e-5
    assert np.abs(mod.encoder.bn1.running_var.cpu().numpy() - fx["bn_running_var:encoder.bn1"]).max() < 1e-5


@pytest.mark.parametrize("kind", ["clip_psp", "clip_ocr"])
def test_clip_heads(dev, kind):
    tag = "r50_" + kind
    fx = golden(tag)
    mod = build(kind, "resnet50dilated")
    load_det(mod, fx=fx)
    zero_dropout(mod)
    mod.to(dev)
    inp = clip_inputs(tag)
    mod.eval()
    store = {}
    h = _hook(mod.ppm_conv.conv_last_ if kind == "clip_psp" else mod.head, store)
    ev = [_t(a, dev) for a in inp["eval_imgs"]]
    others = ev[:-1]
    with torch.no_grad():
        probs = mod({"img_data": ev[-1], "clipimgs_data": others, "seg_label": torch.zeros(1, 1, 64, 96, device=dev)},
                    segSize=(64, 96))
    h.remove()
    assert len(others) == 3, "forward appends the current frame to the caller's list, like the reference"
    _check_eval(fx, probs, store)
    mod.train()
    imgs = [_t(a, dev) for a in inp["train_imgs"]]
    labs = [_t(a, dev) for a in inp["train_labs"]]
    loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1],
                     "cliplabels_data": labs[:-1]})
    loss.backward()
    _check_train(fx, mod, loss, acc, tag)


def test_bn_backward_fusion_matches_unfused(dev):
    """The batch-norm backward reductions produced by the consumer's data-gradient epilogue (ops.BNLink) against the
    separate reduction pass: same loss, same gradient for every parameter, and the fused path is really taken."""
    from cvpr2021_vspw_implement_amd import ops

    tag = "r50_clip_psp"
    inp = clip_inputs(tag)
    grads = {}
    for fused in (True, False):
        ops.set_bn_backward_fusion(fused)
        ops._bn_fusion["fused_nodes"] = 0
        try:
            mod = build("clip_psp", "resnet50dilated")
            load_det(mod, fx=golden(tag))
            zero_dropout(mod)
            mod.to(dev).train()
            imgs = [_t(a, dev) for a in inp["train_imgs"]]
            labs = [_t(a, dev) for a in inp["train_labs"]]
            loss, _ = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1],
                           "cliplabels_data": labs[:-1]})
            loss.backward()
            grads[fused] = (loss.item(), {k: p.grad.double().cpu() for k, p in mod.named_parameters()
                                          if p.grad is not None}, ops._bn_fusion["fused_nodes"])
        finally:
            ops.set_bn_backward_fusion(True)
    assert grads[True][2] >= 40 and grads[False][2] == 0   # R50: 2 + 16*2 + 12 fusable nodes
    assert abs(grads[True][0] - grads[False][0]) < 1e-6 * abs(grads[False][0])
    for k, g in grads[False][1].items():
        err = (grads[True][1][k] - g).norm().item()
        assert err <= 2e-4 * max(g.norm().item(), 1e-6), (k, err, g.norm().item())


def test_clipocr_all_fails_like_the_reference(dev):
    """--clipocr_all pairs B*T pixel frames with B object contexts; the reference's view() raises RuntimeError
    (tests/golden/make_golden.py could not produce a vector for it) and so does the mirror."""
    mod = build("clip_ocr", "resnet50dilated", args={"clipocr_all": True}).to(dev)
    inp = clip_inputs("r50_clip_ocr")
    imgs = [_t(a, dev) for a in inp["train_imgs"]]
    labs = [_t(a, dev) for a in inp["train_labs"]]
    with pytest.raises(RuntimeError, match="invalid"):
        mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1], "cliplabels_data": labs[:-1]})


def test_clip_psp_against_live_oracle(dev):
    """Same comparison against the numpy oracle evaluated here (not a stored vector), on a different seed / shape
    (ragged 57x71 frames, T=2) than any fixture."""
    from oracle import np_models as NM
    from oracle import np_ops as O

    O.set_dtype(np.float32)
    mod = build("clip_psp", "resnet50dilated")
    sd = load_det(mod)
    zero_dropout(mod)
    mod.to(dev)
    T, shape = 2, (2, 3, 57, 71)
    imgs = [det_input("live:%d" % t, shape, seed=99) for t in range(T)]
    labs = [det_labels("live:%d" % t, (2, 1, 57, 71), K, seed=99) for t in range(T)]
    mod.train()
    ti = [_t(a, dev) for a in imgs]
    tl = [_t(a, dev) for a in labs]
    loss, acc = mod({"img_data": ti[-1], "seg_label": tl[-1], "clipimgs_data": ti[:-1], "cliplabels_data": tl[:-1]})
    loss.backward()
    P = NM.Params({k: v.copy() for k, v in sd.items()}, train_params=True)
    oloss, oacc = NM.clip_psp(P, "resnet50", imgs, labs, True)
    O.tape().backward(oloss)
    assert abs(loss.item() - float(oloss.v.reshape(()))) < 2e-4 * abs(loss.item())
    assert abs(acc.item() - oacc) < 2e-3
    g, og = _grads(mod), P.grads()
    worst = 0.0
    for k, v in og.items():
        n = np.linalg.norm(v)
        worst = max(worst, abs(np.linalg.norm(g[k]) - n) / max(n, 1e-3))
    assert worst < 5e-2, worst
    # running statistics follow the reference's momentum / unbiased-variance rule
    rm = mod.encoder.layer4[2].bn3.running_mean.cpu().numpy()
    assert np.abs(rm - P.sd["encoder.layer4.2.bn3.running_mean"]).max() < 1e-4


def test_nonlocal3d(dev):
    tag = "r50_nonlocal3d"
    fx = golden(tag)
    mod = build("nonlocal3d", "resnet50dilated")
    load_det(mod, fx=fx)
    mod.to(dev)
    T, shape = 3, (1, 3, 49, 49)
    imgs = [_t(det_input("%s:train:%d" % (tag, t), shape), dev) for t in range(T)]
    labs = [_t(det_labels("%s:train:%d" % (tag, t), (1, 1, 49, 49), K), dev) for t in range(T)]
    mod.train()
    loss, acc = mod({"clipimgs_data": imgs, "cliplabels_data": labs})
    loss.backward()
    _check_train(fx, mod, loss, acc, tag)
    mod.eval()
    with torch.no_grad():
        preds = mod({"clipimgs_data": imgs, "cliplabels_data": labs}, segSize=(49, 49))
    probs = np.stack([p.float().cpu().numpy() for p in preds])
    assert np.abs(probs[:, :, :, ::4, ::4] - fx["eval_probs_sub"]).max() < 1e-3


def test_nonlocal_decoders_with_the_downsample_switch(dev):
    """Non_local3d(downsample=True) and Non_local2d(downsample=True) (reference models/non_local_models.py:30-32,43-44,
    135-138: affinity on the 2x2-average-pooled embedding, bilinear back up) against vectors from the reference: 73x73
    crops give a 10x10 embedding, pooled to 5x5."""
    import cvpr2021_vspw_implement_amd.models as M
    from cvpr2021_vspw_implement_amd.models.non_local_models import Non_local2d
    from helpers import args_ns

    tag = "r50_nonlocal_downsample"
    fx = golden(tag)
    T, shape = 3, (2, 3, 73, 73)
    crit = torch.nn.NLLLoss(ignore_index=255)
    for which in ("3d", "2d"):
        sub = type("Fx", (), {})()
        pre = which + ":"
        sub.files = [k[len(pre):] for k in fx.files if k.startswith(pre)]
        sub.__class__.__getitem__ = lambda self, k, pre=pre: fx[pre + k]
        enc = M.ModelBuilder.build_encoder(arch="resnet50dilated", fc_dim=2048)
        if which == "3d":
            mod = M.Non_local3d(args_ns(), enc, crit, downsample=True)
            imgs = [_t(det_input("%s:3d:%d" % (tag, t), shape), dev) for t in range(T)]
            labs = [_t(det_labels("%s:3d:%d" % (tag, t), (shape[0], 1) + shape[2:], K), dev) for t in range(T)]
            feed = lambda: {"clipimgs_data": list(imgs), "cliplabels_data": list(labs)}  # noqa: E731
        else:
            mod = M.SegmentationModule(enc, Non_local2d(num_class=K, downsample=True), crit, None)
            img = _t(det_input("%s:2d" % tag, shape), dev)
            lab = _t(det_labels("%s:2d" % tag, (shape[0], 1) + shape[2:], K), dev)
            feed = lambda: {"img_data": img, "seg_label": lab}  # noqa: E731
        load_det(mod, fx=sub)
        mod.to(dev).train()
        loss, acc = mod(feed())
        loss.backward()
        _check_train(sub, mod, loss, acc, tag + ":" + which)
        mod.eval()
        with torch.no_grad():
            out = mod(feed(), segSize=shape[2:])
        preds = out if isinstance(out, (list, tuple)) else [out]
        probs = np.stack([p.float().cpu().numpy() for p in preds])
        assert np.abs(probs[:, :, :, ::2, ::2] - sub["eval_probs_sub"]).max() < 1e-3
        decisive = sub["eval_margin"] > 2e-3
        assert ((probs.argmax(2) != sub["eval_argmax"]) & decisive).sum() == 0


def test_netwarp(dev):
    tag = "r50_netwarp"
    fx = golden(tag)
    shape = (2, 3, 65, 65)

    class FakeRaft(torch.nn.Module):
        def forward(self, a, b, iters=20, test_mode=True):
            n, _, h, w = a.shape
            f = torch.from_numpy(det_input(tag + ":flow", (n, 2, h, w), scale=1.9)) - 0.7
            return None, f.clamp(-10, 10).to(a.device)

    mod = build("netwarp", "resnet50dilated", flow_net=FakeRaft())
    load_det(mod, skip_prefix=("raft.",))
    zero_dropout(mod)
    mod.to(dev)
    mod.train()
    cur = _t(det_input(tag + ":cur", shape), dev)
    prev = _t(det_input(tag + ":prev", shape), dev)
    lab = _t(det_labels(tag + ":lab", (2, 1, 65, 65), K), dev)
    loss, acc = mod({"img_data": cur, "seg_label": lab, "clipimgs_data": [prev], "cliplabels_data": []})
    loss.backward()
    _check_train(fx, mod, loss, acc, tag)


def test_netwarp_with_hip_raft(dev):
    """NetWarp with its own (HIP) RAFT: the flow handed to FlowCNN is the oracle's RAFT flow of the zero-padded,
    un-normalised frames, cropped back (models/netwarp.py:160-176), and the train step runs end to end."""
    from oracle import np_raft
    from helpers import args_ns, raft_state
    import cvpr2021_vspw_implement_amd.models as M

    shape = (1, 3, 131, 150)
    enc = M.ModelBuilder.build_encoder(arch="resnet50dilated", fc_dim=2048)  # the blends are 2048/4096 wide
    dec = M.ModelBuilder.build_decoder(arch="ppm_deepsup_clip", fc_dim=2048, num_class=K)
    mod = M.NetWarp(enc, dec, torch.nn.NLLLoss(ignore_index=255), args_ns(clip_num=2, raft_weights=None), 0.4)
    rsd = raft_state(golden("raft_basic"))
    mod.raft.load_state_dict({k: torch.from_numpy(v) for k, v in rsd.items()})
    zero_dropout(mod)
    mod.to(dev).train()
    cur = det_input("nwraft:cur", shape, scale=0.8)
    prev = det_input("nwraft:prev", shape, scale=0.8)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    a, b = (cur * std + mean) * 255.0, (prev * std + mean) * 255.0
    pad = ((0, 0), (0, 0), (2, 3), (1, 1))  # 131 -> 136 (2,3), 150 -> 152 (1,1): InputPadder 'sintel'
    _, up = np_raft.raft_forward(rsd, np.pad(a, pad), np.pad(b, pad), iters=20)
    ref = up[:, :, 2:-3, 1:-1]
    got = mod._flow(_t(a, dev), _t(b, dev)).cpu().numpy()
    assert got.shape == ref.shape == (1, 2, 131, 150)
    assert np.abs(got - ref).max() < 2e-2 * max(1.0, np.abs(ref).max() / 50)
    lab = _t(det_labels("nwraft:lab", (1, 1) + shape[2:], K), dev)
    loss, acc = mod({"img_data": _t(cur, dev), "seg_label": lab, "clipimgs_data": [_t(prev, dev)],
                     "cliplabels_data": []})
    loss.backward()
    assert np.isfinite(loss.item()) and 0.0 <= acc.item() <= 1.0
    assert all(p.grad is None for p in mod.raft.parameters())
    # w*_1 start at zero (netwarp.py:92-95): the warped branch has no weight yet, so only its blend vectors see gradient
    assert mod.w0_1.grad.abs().sum().item() > 0 and mod.w1_1.grad.abs().sum().item() > 0


def test_netwarp_ocr(dev):
    tag = "r50_netwarp_ocr"
    fx = golden(tag)
    shape = (2, 3, 65, 65)

    class FakeRaft(torch.nn.Module):
        def forward(self, a, b, iters=20, test_mode=True):
            n, _, h, w = a.shape
            f = torch.from_numpy(det_input(tag + ":flow", (n, 2, h, w), scale=1.9)) - 0.7
            return None, f.clamp(-10, 10).to(a.device)

    mod = build("netwarp_ocr", "resnet50dilated", flow_net=FakeRaft())
    load_det(mod, skip_prefix=("raft.",))
    zero_dropout(mod)
    mod.to(dev)
    mod.train()
    cur = _t(det_input(tag + ":cur", shape), dev)
    prev = _t(det_input(tag + ":prev", shape), dev)
    lab = _t(det_labels(tag + ":lab", (2, 1, 65, 65), K), dev)
    plab = _t(det_labels(tag + ":plab", (2, 1, 65, 65), K), dev)
    loss, acc = mod({"img_data": cur, "seg_label": lab, "clipimgs_data": [prev], "cliplabels_data": [plab]})
    loss.backward()
    _check_train(fx, mod, loss, acc, tag)


def test_clip_psp_temporal_weights(dev):
    """args.psp_weight: softmax-over-T weighting of the pooled features (clip_psp.py:147-152,184-186), forward and
    the gradient of pspweight_conv."""
    tag = "r50_clip_psp_pspw"
    fx = golden(tag)
    mod = build("clip_psp", "resnet50dilated", args={"psp_weight": True})
    load_det(mod, fx=fx)
    zero_dropout(mod)
    mod.to(dev)
    inp = clip_inputs(tag)
    mod.eval()
    store = {}
    h = _hook(mod.ppm_conv.conv_last_, store)
    ev = [_t(a, dev) for a in inp["eval_imgs"]]
    with torch.no_grad():
        probs = mod({"img_data": ev[-1], "clipimgs_data": ev[:-1], "seg_label": torch.zeros(1, 1, 64, 96, device=dev)},
                    segSize=(64, 96))
    h.remove()
    _check_eval(fx, probs, store)
    mod.train()
    imgs = [_t(a, dev) for a in inp["train_imgs"]]
    labs = [_t(a, dev) for a in inp["train_labs"]]
    loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1],
                     "cliplabels_data": labs[:-1]})
    loss.backward()
    _check_train(fx, mod, loss, acc, tag)
    assert mod.pspweight_conv[0].weight.grad is not None


def test_clip_ocr_memory_bank(dev):
    """use_memory inference over three consecutive frames (is_clean_memory on the first)."""
    tag = "r50_clip_ocr_memory"
    fx = golden(tag)
    mod = build("clip_ocr", "resnet50dilated", args={"use_memory": True, "memory_num": 4})
    load_det(mod, fx=fx)
    mod.to(dev).eval()
    shape = (1, 3, 64, 96)
    for c in range(3):
        imgs = [_t(det_input("%s:call%d:%d" % (tag, c, t), shape), dev) for t in range(3)]
        with torch.no_grad():
            probs = mod({"img_data": imgs[-1], "clipimgs_data": imgs[:-1], "is_clean_memory": c == 0,
                         "seg_label": torch.zeros(1, 1, 64, 96, device=dev)}, segSize=(64, 96))
        assert len(mod.memory) == int(fx["call%d_memlen" % c])
        err = np.abs(probs.float().cpu().numpy()[:, :, ::4, ::4] - fx["call%d_probs_sub" % c]).max()
        assert err < 2e-3, (c, err)


def test_bench_shape_properties(dev):
    """BASELINE-size (T=5, B=2, 479x479, R101 TCB-PSP) size-independent properties: the step runs, loss is finite and
    ~log(K) at init, every parameter receives a finite gradient, and the loss is linear in the incoming gradient
    (backward(2*loss) = 2*backward(loss))."""
    import math

    mod = build("clip_psp", "resnet101dilated").to(dev)
    mod.train()
    zero_dropout(mod)
    g = torch.Generator().manual_seed(304)
    T, B, S = 5, 2, 479
    imgs = [torch.randn(B, 3, S, S, generator=g).to(dev) for _ in range(T)]
    labs = [torch.randint(0, K, (B, 1, S, S), generator=g).float().to(dev) for _ in range(T)]

    def step(scale):
        mod.zero_grad()
        loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": list(imgs[:-1]),
                         "cliplabels_data": list(labs[:-1])})
        (loss * scale).backward()
        return loss.item(), {k: p.grad.clone() for k, p in mod.named_parameters()}

    # freeze running-stat drift between the two runs: it does not affect train-mode outputs
    l1, g1 = step(1.0)
    l2, g2 = step(2.0)
    assert math.isfinite(l1) and abs(l1 - l2) < 1e-5 * abs(l1)
    assert 0.5 * 1.4 * math.log(K) < l1 < 3 * 1.4 * math.log(K)
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        n1, n2 = g1[k].norm().item(), g2[k].norm().item()
        assert abs(n2 - 2 * n1) <= 1e-3 * max(n2, 1e-12), k


@pytest.mark.parametrize("kind", ["clip_psp", "clip_ocr"])
def test_frozen_bn_training_step_elementwise(dev, kind):
    """cfg.TRAIN.fix_bn: loss + gradients with the module in eval mode (running statistics, no dropout), gated
    ELEMENT-WISE on 113 stored gradient tensors (stem / layer1 / layer2 convolutions, every BatchNorm weight and bias)
    against the reference's FLOAT64 run.

    Measured (tools/diag/fixbn.py): even without batch statistics these random-weight gradients are ill-conditioned -
    the reference's OWN fp32 gradients differ from its float64 ones by up to 4 % of a tensor's largest entry (0.7 % on
    a parameter's norm; 65x65 crops, 9x9 maps, ReLU / max-pool decisions amplify fp32 rounding ~1e5-fold).  A flat 1e-3
    gate is therefore not attainable by ANY fp32 implementation, the reference included; the HIP path is held to the
    reference's own measured fp32 error instead: per tensor  |hip - ref64| <= max(1e-3, 6 x |ref32 - ref64|)  (largest
    observed ratio 4.2), median ratio <= 2 (observed 1.3-1.4), per-parameter norm error RMS <= 2 x the reference's own
    (observed 1.1-1.4 x) and <= 2e-2 anywhere; loss within 3e-5 (observed 1.3e-6 PSP, 1.1e-5 OCR)."""
    tag = "r50_%s_fixbn" % kind
    fx = golden(tag)
    mod = build(kind, "resnet50dilated")
    load_det(mod, fx=fx)
    mod.to(dev)
    inp = clip_inputs(tag)
    mod.eval()  # train_clip2.py: segmentation_module.train(not cfg.TRAIN.fix_bn)
    imgs = [_t(a, dev) for a in inp["train_imgs"]]
    labs = [_t(a, dev) for a in inp["train_labs"]]
    rm0 = mod.encoder.layer3[2].bn2.running_mean.clone()
    loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1],
                     "cliplabels_data": labs[:-1]})
    loss.backward()
    assert torch.equal(rm0, mod.encoder.layer3[2].bn2.running_mean), "frozen BN must not update its statistics"
    ref_loss = float(fx["train_loss64"])
    assert abs(loss.item() - ref_loss) < 3e-5 * abs(ref_loss), (loss.item(), ref_loss)  # the reference's fp32: 7e-6
    assert abs(acc.item() - float(fx["train_acc"])) < 2e-3
    g = {k: v.astype(np.float64) for k, v in _grads(mod).items()}
    names = [str(n) for n in fx["grad_names"]]
    e_hip, e_ref = [], []
    for n, r64, r32 in zip(names, fx["grad_norms64"], fx["grad_norms"]):
        e_hip.append(abs(float(np.linalg.norm(g[n])) - r64) / r64)
        e_ref.append(abs(r32 - r64) / r64)
    e_hip, e_ref = np.array(e_hip), np.array(e_ref)
    rms_hip, rms_ref = float(np.sqrt((e_hip ** 2).mean())), float(np.sqrt((e_ref ** 2).mean()))
    assert rms_hip <= 2.0 * rms_ref, (rms_hip, rms_ref)
    assert e_hip.max() < 2e-2, (e_hip.max(), names[int(e_hip.argmax())])
    ratios = []
    for key in fx.files:
        if key.startswith("grad64:"):
            ref = fx[key].astype(np.float64)
            top = np.abs(ref).max()
            eh = np.abs(g[key[7:]] - ref).max() / top
            er = np.abs(fx["grad:" + key[7:]].astype(np.float64) - ref).max() / top
            assert eh <= max(1e-3, 6.0 * er), (key, eh, er)
            ratios.append(eh / max(er, 1e-12))
    assert len(ratios) > 100
    assert float(np.median(ratios)) <= 2.0, np.median(ratios)


def test_bn_backward_affine_operand_matches_materialised_dy(dev):
    """BatchNorm's backward apply folded into the pointwise convs' gradient GEMMs (ops: vspw_conv2d_bwd_data_aff /
    _weight_aff) against the path that writes dy and feeds it to the plain GEMMs: same loss, every gradient within
    fp32 rounding of the other formulation, and the folded path is really taken."""
    from cvpr2021_vspw_implement_amd import ops

    tag = "r50_clip_psp"
    inp = clip_inputs(tag, train_shape=(2, 3, 57, 57))  # 8x8 maps: 384 pixel rows, a multiple of the 32-pixel K-tile
    results = []
    for affine in (True, False):
        ops._bn_fusion["affine"] = affine
        ops._bn_fusion["affine_nodes"] = 0
        try:
            mod = build("clip_psp", "resnet50dilated")
            load_det(mod)
            zero_dropout(mod)
            mod.to(dev).train()
            imgs = [_t(a, dev) for a in inp["train_imgs"]]
            labs = [_t(a, dev) for a in inp["train_labs"]]
            loss, _ = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1],
                           "cliplabels_data": labs[:-1]})
            loss.backward()
            torch.cuda.synchronize()
            results.append((loss.item(), _grads(mod), ops._bn_fusion["affine_nodes"]))
        finally:
            ops._bn_fusion["affine"] = True
    (l1, g1, n1), (l0, g0, n0) = results
    assert n1 >= 10 and n0 == 0, (n1, n0)
    assert l1 == l0
    for k in g0:
        den = np.abs(g0[k]).max() + 1e-12
        assert np.abs(g1[k] - g0[k]).max() <= 2e-4 * den, (k, np.abs(g1[k] - g0[k]).max() / den)


def test_deferred_forward_apply_is_bit_identical(dev):
    """A bottleneck's final BatchNorm apply + skip + ReLU (reference models/resnet.py:83-90) evaluated inside the next
    block's conv1 GEMM, and bn2 + ReLU evaluated inside conv3's (vspw_conv2d_fwd_apply), against the separate apply
    passes: the same bits everywhere - loss, every gradient, the running statistics - and the fused path is really taken."""
    from cvpr2021_vspw_implement_amd import ops

    from cvpr2021_vspw_implement_amd.models import resnet

    tag = "r50_clip_psp"
    inp = clip_inputs(tag, train_shape=(2, 3, 57, 57))
    results = []
    for fused in (True, False):
        ops._fwd_apply["enabled"] = fused
        ops._fwd_apply["nodes"] = 0
        defer_conv2, resnet._DEFER_CONV2 = resnet._DEFER_CONV2, True  # exercise the optional conv2 -> conv3 hand-over too
        try:
            mod = build("clip_psp", "resnet50dilated")
            load_det(mod)
            zero_dropout(mod)
            mod.to(dev).train()
            imgs = [_t(a, dev) for a in inp["train_imgs"]]
            labs = [_t(a, dev) for a in inp["train_labs"]]
            loss, _ = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1],
                           "cliplabels_data": labs[:-1]})
            loss.backward()
            ops.join_side_streams()
            torch.cuda.synchronize()
            stats = {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items() if "running_" in k}
            results.append((loss.item(), _grads(mod), stats, ops._fwd_apply["nodes"]))
        finally:
            ops._fwd_apply["enabled"] = True
            resnet._DEFER_CONV2 = defer_conv2
    (l1, g1, s1, n1), (l0, g0, s0, n0) = results
    # conv2 -> conv3 inside each of the 16 bottlenecks, block output -> next conv1 for every non-final block of a layer
    assert n1 == 16 + (3 - 1) + (4 - 1) + (6 - 1) + (3 - 1) and n0 == 0, (n1, n0)
    assert l1 == l0
    for k in g0:
        assert np.array_equal(g1[k], g0[k]), k
    for k in s0:
        assert np.array_equal(s1[k], s0[k]), k


@pytest.mark.parametrize("tag,kind,arch,decoder,fc_dim", [
    ("r18_ppm_deepsup", "seg", "resnet18", "ppm_deepsup", 512),
    ("r50_ocrnet_deepsup", "seg", "resnet50", "ocrnet_deepsup", 2048),
    ("r50_nonlocal2d", "seg", "resnet50", "nonlocal2d", 2048),
    ("r50_clip_psp", "clip_psp", "resnet50", None, 2048),
    ("r50_clip_ocr", "clip_ocr", "resnet50", None, 2048),
])
def test_training_gradients_with_pinned_decisions(dev, tag, kind, arch, decoder, fc_dim):
    """Every fixture configuration's training step, gradient by gradient: HIP against the float64 oracle with the HIP
    forward's decisions injected; median / 99th percentile / maximum of the per-parameter relative L2 error no more than
    max(1e-3, 1.5x the float32 oracle's) (the oracle's GEMMs in matrix-core accumulation order, its own decisions)."""
    from cvpr2021_vspw_implement_amd import ops
    from oracle import np_models as NM
    from oracle import np_ops as O

    fx = golden(tag)
    if kind == "seg":
        mod = build("seg", arch + "dilated", decoder, fc_dim, deep_sup_scale=None if decoder == "nonlocal2d" else 0.4)
    else:
        mod = build(kind, arch + "dilated")
    sd = load_det(mod, fx=fx)
    zero_dropout(mod)
    mod.to(dev).train()
    if kind == "seg":
        inp = seg_inputs(tag)
        feed = {"img_data": _t(inp["train_img"], dev), "seg_label": _t(inp["train_lab"], dev)}

        def fn(P, dt):
            return NM.segmentation_module(P, arch, inp["train_img"].astype(dt), inp["train_lab"], True,
                                          None if decoder == "nonlocal2d" else 0.4, decoder=decoder)
    else:
        inp = clip_inputs(tag)
        imgs = [_t(a, dev) for a in inp["train_imgs"]]
        labs = [_t(a, dev) for a in inp["train_labs"]]
        feed = {"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": imgs[:-1], "cliplabels_data": labs[:-1]}

        def fn(P, dt):
            f = NM.clip_psp if kind == "clip_psp" else NM.clip_ocr
            return f(P, arch, [a.astype(dt) for a in inp["train_imgs"]], inp["train_labs"], True)
    taps = []
    ops.record_decisions(taps)
    try:
        loss, acc = mod(feed)
    finally:
        ops.record_decisions(None)
    loss.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().double().cpu().numpy() for k, p in mod.named_parameters() if p.grad is not None}
    names, e_hip, e_or = pinned_gradient_errors(fn, sd, hip_decision_store(mod, taps), g)
    assert O.F32 is np.float32
    w = int(e_hip.argmax())
    print("%s pinned: HIP median %.2e p99 %.2e max %.2e (%s) | float32 oracle median %.2e p99 %.2e max %.2e"
          % (tag, np.median(e_hip), np.percentile(e_hip, 99), e_hip.max(), names[w], np.median(e_or),
             np.percentile(e_or, 99), e_or.max()))
    for what, f in (("median", np.median), ("p99", lambda v: np.percentile(v, 99)), ("max", np.max)):
        assert f(e_hip) <= max(1e-3, 1.5 * f(e_or)), (what, f(e_hip), f(e_or))
