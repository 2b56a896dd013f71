"""Every BASELINE.json configuration at ITS OWN size against vectors produced by the REFERENCE itself
(tests/golden/make_golden_fullsize.py: /root/reference imported in the build container, float32 and float64):
inference at 480x853 for cfg 1-4, one training step at R101 / B=2 / 479x479 for cfg 2, 3, 4 (T=5), 5a, 5b (Non_local3d,
T=5: 18 000 positions) and 5c (NetWarp, synthetic flow).  No oracle is
evaluated here: stored arrays only, so the whole file takes seconds per case.

Two weight variants per case (see make_golden_fullsize.py / det_init.damp_residual_gammas):
  damped  well-conditioned (residual-closing BatchNorm gammas x0.25): the reference's fp32 logits are < 3e-4 from its
          float64 re-run, and north_star's rule applies AS WRITTEN: |hip - ref32| <= 1e-3 on the logits, arg-max
          identical wherever the reference's top-2 log-probability gap exceeds 2e-3.
  raw     He-normal weights, gamma~U(0.5,1.5): 33 random BatchNorm'd residual blocks amplify float32 rounding ~1e4-fold;
          the reference's own fp32 logits are 1e-3..4e-3 from float64.  Both numbers are printed: the direct
          |hip - ref32|, and |hip - ref64| held to max(1e-3, RAW_GATE x |ref32 - ref64|), RAW_GATE = 1.5.
"""
import numpy as np
import pytest
import torch

from helpers import K, build, golden, load_det, zero_dropout
from oracle.det_init import damp_residual_gammas, det_input, det_labels, det_sample_index

pytestmark = pytest.mark.gpu
H, W, S = 480, 853, 479
# raw-weight variant: |hip - ref64| <= max(1e-3, RAW_GATE x |ref32 - ref64|).  Round 4 needed 2.0: HIP sat 1.3-1.6x farther
# from float64 than the reference's own fp32.  Attributed in round 5 (profiles/r05_parity_attrib_*.log): not Winograd, not
# the long pointwise reductions - the 576-term k-sequential chains of the direct 3x3 kernels in the stem / layer1, whose
# error every later block amplifies.  With those chains folded every 96 terms (conv_igemm.hip FOLD) the measured ratio is
# 0.80-1.20 on cfg 2/3/4, inference and training; the max over ~10^5 logits of one rounding realisation moves by +-0.2.
RAW_GATE = 1.5
RAW_FLIPS = 1.5  # arg-max disagreements with the reference's fp32 masks <= RAW_FLIPS x its own fp32-vs-fp64 count


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _module(kind, T):
    if kind == "r18_ppm":
        return build("seg", "resnet18dilated", "ppm_deepsup", 512), (lambda m: m.decoder.conv_last_)
    if kind == "r101_ppm":
        return build("seg", "resnet101dilated", "ppm_deepsup", 2048), (lambda m: m.decoder.conv_last_)
    if kind == "r101_nonlocal2d":
        return (build("seg", "resnet101dilated", "nonlocal2d", 2048, deep_sup_scale=None),
                (lambda m: m.decoder.last_layer))
    if kind == "nonlocal3d":
        return build("nonlocal3d", "resnet101dilated"), (lambda m: m.last_layer)
    if kind == "netwarp":
        return build("netwarp", "resnet101dilated", flow_net=_FakeRaft()), (lambda m: m.conv_last_)
    if kind == "clip_psp":
        return build(kind, "resnet101dilated", args={"clip_num": T}), (lambda m: m.ppm_conv)
    return build(kind, "resnet101dilated", args={"clip_num": T}), (lambda m: m.head)


class _FakeRaft(torch.nn.Module):
    """the fixed synthetic flow field make_golden_fullsize.FakeRaft hands the reference (RAFT itself: raft_basic.npz)"""

    def forward(self, a, b, iters=20, test_mode=True):
        n, _, h, w = a.shape
        f = torch.from_numpy(det_input("train479:netwarp:flow", (n, 2, h, w), scale=1.9)) - 0.7
        return None, f.clamp(-10, 10).to(a.device)


def _load(mod, variant, fx=None):
    """deterministic name-keyed weights (+ the reference-calibrated running statistics of an eval fixture)"""
    load_det(mod, fx=fx, skip_prefix=("raft.",) if hasattr(mod, "raft") else ())
    if variant == "damped":
        sd = {k: v.clone() for k, v in mod.state_dict().items()}
        assert damp_residual_gammas(sd)
        mod.load_state_dict(sd)
    zero_dropout(mod)


def _feed(frames, labels, clip, kind=None):
    if kind == "nonlocal3d":
        return {"clipimgs_data": list(frames), "cliplabels_data": list(labels)}
    if kind == "netwarp":
        return {"img_data": frames[-1], "seg_label": labels[-1], "clipimgs_data": [frames[0]], "cliplabels_data": []}
    d = {"img_data": frames[-1], "seg_label": labels[-1]}
    if clip:
        d.update(clipimgs_data=list(frames[:-1]), cliplabels_data=list(labels[:-1]))
    return d


EVAL = [("r18_ppm", "cfg1"), ("r101_ppm", "cfg2"), ("clip_psp", "cfg3"), ("clip_ocr", "cfg4")]
TRAIN = [("r101_ppm", "cfg2"), ("clip_psp", "cfg3"), ("clip_ocr", "cfg4"), ("r101_nonlocal2d", "cfg5a"),
         ("nonlocal3d", "cfg5b"), ("netwarp", "cfg5c")]


@pytest.mark.parametrize("variant", ["damped", "raw"])
@pytest.mark.parametrize("kind,cfg", EVAL)
def test_480p_inference_against_reference_vectors(dev, kind, cfg, variant):
    fx = golden("full_eval_%s_%s_%s" % (cfg, kind, variant))
    clip = kind in ("clip_psp", "clip_ocr")
    T = 4 if clip else 1
    mod, tap = _module(kind, T)
    _load(mod, variant, fx)
    mod.to(dev).eval()
    name = "infer480:" + kind
    frames = [_t(det_input("%s:%d" % (name, t), (1, 3, H, W)), dev) for t in range(T)]
    zeros = [torch.zeros(1, 1, H, W, device=dev)] * T
    from cvpr2021_vspw_implement_amd import ops

    # Without autograd the HIP path is the FOLDED one (conv + eval BatchNorm (+ residual) + ReLU in one launch on weights
    # with the BatchNorm scale folded in: ops._conv_bn_folded); the unfolded path (conv, then BatchNorm apply) is run too
    got = {}
    for folded in (True, False):
        store = {}
        hk = tap(mod).register_forward_hook(lambda m, i, o: store.__setitem__("l", o.detach().float().cpu().numpy()))
        hk1 = mod.encoder.layer1.register_forward_hook(lambda m, i, o: store.__setitem__("l1", o.detach().double().cpu()))
        ops.set_inference_folding(folded)
        try:
            with torch.no_grad():
                p_ = mod(_feed(frames, zeros, clip), segSize=(H, W))
        finally:
            ops.set_inference_folding(True)
            hk.remove()
            hk1.remove()
        got[folded] = (store["l"], p_.float().cpu().numpy(), store["l1"])
    logits, probs, _ = got[True]
    unfolded = got[False][0][:, :, ::2, ::2]
    shallow = float((got[True][2] - got[False][2]).norm() / got[False][2].norm())
    assert shallow <= 1e-5, shallow  # folded vs unfolded, 10 convolutions deep
    assert logits.shape == (1, K, 60, 107) and probs.shape == (1, K, H, W)
    sub = logits[:, :, ::2, ::2]
    own = float(fx["ref32_vs_ref64_logits_max"])
    e32 = float(np.abs(sub - fx["logits32_sub"]).max())
    e64 = float(np.abs(sub.astype(np.float64) - fx["logits64_sub"]).max())
    ep = float(np.abs(probs[:, :, ::16, ::16] - fx["probs32_sub"]).max())
    am = probs.argmax(1).astype(np.uint8)
    flips = am != fx["argmax32"]
    margin = fx["margin32"].astype(np.float32)
    tol = 1e-3 if variant == "damped" else max(1e-3, RAW_GATE * own)
    decisive = margin > 2 * tol
    print("%s %s 480x853 vs the reference: |logit| max %.2f; |ref32 - ref64| %.2e; |hip - ref32| %.2e; |hip - ref64| %.2e "
          "(tol %.1e, excess over the reference's own fp32 %.2fx); probs %.2e; arg-max: %d of %d pixels differ (reference fp32 vs its own fp64: %d), %d decisive"
          % (cfg, variant, float(fx["logits_absmax"]), own, e32, e64, tol, e64 / own, ep, flips.sum(), flips.size,
             (fx["argmax32"] != fx["argmax64"]).sum(), (flips & decisive).sum()))
    if variant == "damped":
        assert own < 5e-4, own  # north_star, unwidened
        assert e32 <= tol, (e32, tol)
        assert float(np.abs(unfolded - fx["logits32_sub"]).max()) <= tol
    else:
        assert e64 <= tol, (e64, tol)
        assert float(np.abs(unfolded.astype(np.float64) - fx["logits64_sub"]).max()) <= tol
    assert ep <= 1e-3, ep
    assert np.abs(probs.sum(1) - 1).max() < 1e-5
    assert (flips & decisive).sum() == 0
    own_flips = int((fx["argmax32"] != fx["argmax64"]).sum())
    if variant == "damped":  # no more disagreement with the reference's fp32 masks than 3x its own fp32-vs-fp64 count (measured worst: 2.6x)
        assert flips.sum() <= 3 * max(own_flips, 25), flips.sum()
    else:  # (two fp32 realisations differ ~1.4x more often from each other than either does from float64)
        assert flips.sum() <= RAW_FLIPS * max(own_flips, 25), (flips.sum(), own_flips)


@pytest.mark.parametrize("variant", ["damped", "raw"])
@pytest.mark.parametrize("kind,cfg", TRAIN)
def test_479_training_step_against_reference_vectors(dev, kind, cfg, variant):
    """loss / accuracy / train-mode logits / BatchNorm running statistics / every parameter's gradient (norm + 256 fixed
    elements) of one step.  ReLU and max-pool decisions are free on both sides, so the gradient gate is relative to the
    reference's OWN float32-vs-float64 distance on the same statistic (factor 2 on median / p99: one rounding realisation
    differs from the next by up to 1.5x; factor 3 on the maximum), with an absolute floor of 1e-3."""
    fx = golden("full_train_%s_%s_%s" % (cfg, kind, variant))
    clip = kind in ("clip_psp", "clip_ocr")
    T = {"clip_psp": 5, "clip_ocr": 5, "nonlocal3d": 5, "netwarp": 2}.get(kind, 1)
    B = 2
    mod, tap = _module(kind, T)
    _load(mod, variant)
    mod.to(dev).train()
    name = "train479:" + kind
    frames = [_t(det_input("%s:%d" % (name, t), (B, 3, S, S)), dev) for t in range(T)]
    labels = [_t(det_labels("%s:%d" % (name, t), (B, 1, S, S), K), dev) for t in range(T)]
    store = {}
    hk = tap(mod).register_forward_hook(lambda m, i, o: store.__setitem__("l", o.detach().float().cpu().numpy()))
    loss, acc = mod(_feed(frames, labels, clip, kind))
    hk.remove()
    loss.backward()
    torch.cuda.synchronize()
    l32, l64 = float(fx["loss32"]), float(fx["loss64"])
    lh = loss.item()
    own_loss = abs(l32 - l64) / abs(l64)
    # train-mode logits of the head
    sub = store["l"][: fx["logits32_sub"].shape[0], :, ::2, ::2]  # (the fixture keeps the first 4 images)
    own_l = float(np.abs(fx["logits32_sub"] - fx["logits64_sub"]).max())
    e_l32 = float(np.abs(sub - fx["logits32_sub"]).max())
    e_l64 = float(np.abs(sub - fx["logits64_sub"]).max())
    # running statistics after the step (momentum 0.1, unbiased variance)
    mods = dict(mod.named_modules())
    for key in fx.files:
        if key.startswith("running_mean:") or key.startswith("running_var:"):
            what, bn = key.split(":")
            got = getattr(mods[bn], what).detach().cpu().numpy()
            assert np.abs(got - fx[key]).max() <= 1e-4 * max(1.0, np.abs(fx[key]).max()), key
    # gradients
    names = [str(n) for n in fx["grad_names"]]
    grads = {k: p.grad for k, p in mod.named_parameters() if p.grad is not None}
    assert set(names) == set(grads), sorted(set(names) ^ set(grads))[:5]
    n32, n64 = fx["grad_norms32"], fx["grad_norms64"]
    scale = float(n64.max())
    nh = np.array([float(grads[k].double().norm()) for k in names])
    den = np.maximum(n64, 1e-3 * scale)
    rel_h, rel_o = np.abs(nh - n64) / den, np.abs(n32 - n64) / den
    # sampled elements: per-parameter relative L2 over the 256 fixed positions
    s32, s64 = fx["grad_samples32"].astype(np.float64), fx["grad_samples64"]
    sh, off = [], 0
    es_h, es_o = [], []
    for k in names:
        g = grads[k].detach().contiguous().view(-1)
        idx = det_sample_index(k, g.numel())
        v = g[_t(idx, dev)].double().cpu().numpy()
        r64, r32 = s64[off:off + len(idx)], s32[off:off + len(idx)]
        off += len(idx)
        floor = 1e-3 * scale * (len(idx) / g.numel()) ** 0.5  # the sample's share of a norm at 1e-3 of the largest
        d = max(float(np.linalg.norm(r64)), floor)
        es_h.append(float(np.linalg.norm(v - r64)) / d)
        es_o.append(float(np.linalg.norm(r32 - r64)) / d)
    es_h, es_o = np.array(es_h), np.array(es_o)
    st = lambda v: "median %.2e p99 %.2e max %.2e" % (np.median(v), np.percentile(v, 99), v.max())  # noqa: E731
    print("%s %s 479x479 step vs the reference: loss hip %.7f ref32 %.7f ref64 %.7f; logits |hip-ref32| %.2e |hip-ref64| "
          "%.2e (|ref32-ref64| %.2e, excess %.2fx)\n  grad norms, rel. to ref64:   HIP %s | reference fp32 %s\n  grad samples, rel. L2:       "
          "HIP %s | reference fp32 %s"
          % (cfg, variant, lh, l32, l64, e_l32, e_l64, own_l, e_l64 / own_l, st(rel_h), st(rel_o), st(es_h), st(es_o)))
    assert abs(lh - l64) <= max(2e-5, 2 * own_loss) * abs(l64), (lh, l32, l64)
    assert abs(acc.item() - float(fx["acc64"])) <= max(1e-3, 2 * abs(float(fx["acc32"]) - float(fx["acc64"])))
    assert e_l64 <= max(1e-3, RAW_GATE * own_l), (e_l64, own_l)
    if own_l < 5e-4:
        assert e_l32 <= 1e-3, e_l32  # north_star as written
    # the maximum over ~680 tensors of ONE rounding realisation is a heavy-tailed statistic (measured on the raw cfg 5a
    # case: 2.3x the reference's own): factor 3 there, 2 on the median and the 99th percentile
    for what, f, k in (("median", np.median, 2.0), ("p99", lambda v: np.percentile(v, 99), 2.0), ("max", np.max, 3.0)):
        assert f(rel_h) <= max(1e-3, k * f(rel_o)), ("norms", what, f(rel_h), f(rel_o))
        assert f(es_h) <= max(1e-3, k * f(es_o)), ("samples", what, f(es_h), f(es_o))


@pytest.mark.parametrize("variant", ["damped", "raw"])
def test_479_training_step_cfg5b_at_T7_against_reference_fp32(dev, variant):
    """cfg 5b at BASELINE.json's own clip length: Non_local3d over T=7 frames (25 200 positions, 2 x 2.54 GB affinity that
    the HIP path never materialises), one training step at B=2 / 479x479 against the REFERENCE's float32 run
    (full_train_cfg5b_t7_*: the float64 re-run needs > 64 GB and does not fit the build container, so there is no
    |ref32 - ref64| of its own: the yardstick is the T=5 fixture's, same model, same variant).  Two float32
    realisations are compared, each `own` from the truth: |hip - ref32| <= (1 + RAW_GATE) x own, and the damped
    variant's logits meet north_star's flat 1e-3 directly."""
    fx = golden("full_train_cfg5b_t7_nonlocal3d_%s" % variant)
    y5 = golden("full_train_cfg5b_nonlocal3d_%s" % variant)  # yardstick: the reference's own fp32-vs-fp64 at T=5
    T, B = 7, 2
    mod, tap = _module("nonlocal3d", T)
    _load(mod, variant)
    mod.to(dev).train()
    name = "train479:nonlocal3d"
    frames = [_t(det_input("%s:%d" % (name, t), (B, 3, S, S)), dev) for t in range(T)]
    labels = [_t(det_labels("%s:%d" % (name, t), (B, 1, S, S), K), dev) for t in range(T)]
    store = {}
    hk = tap(mod).register_forward_hook(lambda m, i, o: store.__setitem__("l", o.detach().float().cpu().numpy()))
    loss, acc = mod(_feed(frames, labels, False, "nonlocal3d"))
    hk.remove()
    loss.backward()
    torch.cuda.synchronize()
    lh, l32 = loss.item(), float(fx["loss32"])
    own_loss = abs(float(y5["loss32"]) - float(y5["loss64"])) / abs(float(y5["loss64"]))
    sub = store["l"][: fx["logits32_sub"].shape[0], :, ::2, ::2]
    own_l = float(np.abs(y5["logits32_sub"] - y5["logits64_sub"]).max())
    e_l32 = float(np.abs(sub - fx["logits32_sub"]).max())
    names = [str(n) for n in fx["grad_names"]]
    grads = {k: p.grad for k, p in mod.named_parameters() if p.grad is not None}
    assert set(names) == set(grads), sorted(set(names) ^ set(grads))[:5]
    n32 = fx["grad_norms32"]
    scale = float(n32.max())
    nh = np.array([float(grads[k].double().norm()) for k in names])
    rel_h = np.abs(nh - n32) / np.maximum(n32, 1e-3 * scale)
    y_n32, y_n64 = y5["grad_norms32"], y5["grad_norms64"]
    rel_y = np.abs(y_n32 - y_n64) / np.maximum(y_n64, 1e-3 * float(y_n64.max()))
    s32, off, es_h = fx["grad_samples32"].astype(np.float64), 0, []
    for k in names:
        g = grads[k].detach().contiguous().view(-1)
        idx = det_sample_index(k, g.numel())
        v = g[_t(idx, dev)].double().cpu().numpy()
        r32 = s32[off:off + len(idx)]
        off += len(idx)
        floor = 1e-3 * scale * (len(idx) / g.numel()) ** 0.5
        es_h.append(float(np.linalg.norm(v - r32)) / max(float(np.linalg.norm(r32)), floor))
    es_h = np.array(es_h)
    ys32, ys64, off, es_y = y5["grad_samples32"].astype(np.float64), y5["grad_samples64"], 0, []
    y_names = [str(n) for n in y5["grad_names"]]
    assert y_names == names
    y_scale = float(y_n64.max())
    for k in names:
        n_el = grads[k].numel()
        idx = det_sample_index(k, n_el)
        r64, r32 = ys64[off:off + len(idx)], ys32[off:off + len(idx)]
        off += len(idx)
        floor = 1e-3 * y_scale * (len(idx) / n_el) ** 0.5
        es_y.append(float(np.linalg.norm(r32 - r64)) / max(float(np.linalg.norm(r64)), floor))
    es_y = np.array(es_y)
    st = lambda v: "median %.2e p99 %.2e max %.2e" % (np.median(v), np.percentile(v, 99), v.max())  # noqa: E731
    print("cfg5b %s T=7 479x479 step vs the reference's fp32: loss hip %.7f ref32 %.7f; acc hip %.5f ref32 %.5f; logits "
          "|hip-ref32| %.2e (T=5 yardstick |ref32-ref64| %.2e)\n  grad norms, rel. to ref32:   HIP %s | yardstick %s\n"
          "  grad samples, rel. L2:       HIP %s | yardstick %s"
          % (variant, lh, l32, acc.item(), float(fx["acc32"]), e_l32, own_l, st(rel_h), st(rel_y), st(es_h), st(es_y)))
    two = 1.0 + RAW_GATE  # two float32 realisations, each within its own distance of the truth
    assert abs(lh - l32) <= max(2e-5, two * own_loss) * abs(l32), (lh, l32)
    assert abs(acc.item() - float(fx["acc32"])) <= 1e-3
    if variant == "damped":
        assert e_l32 <= 1e-3, e_l32  # north_star as written
    else:
        assert e_l32 <= max(1e-3, two * own_l), (e_l32, own_l)
    for key in fx.files:
        if key.startswith("running_mean:") or key.startswith("running_var:"):
            what, bn = key.split(":")
            got = getattr(dict(mod.named_modules())[bn], what).detach().cpu().numpy()
            assert np.abs(got - fx[key]).max() <= 1e-4 * max(1.0, np.abs(fx[key]).max()), key
    for what, f, k in (("median", np.median, 3.0), ("p99", lambda v: np.percentile(v, 99), 3.0), ("max", np.max, 4.0)):
        assert f(rel_h) <= max(1e-3, k * f(rel_y)), ("norms", what, f(rel_h), f(rel_y))
        assert f(es_h) <= max(1e-3, k * f(es_y)), ("samples", what, f(es_h), f(es_y))
