"""BASELINE.json configurations at their full sizes on the MI355X.
cfg 1 (resnet18dilated + ppm_deepsup, one 480x853 frame) is small enough for the numpy oracle to be evaluated live and
compared value by value; the larger configurations are checked through size-independent properties (probabilities
normalised, finite loss ~ log K at random init, finite gradients on every parameter, loss linear in the incoming
gradient, eval deterministic)."""
import math

import numpy as np
import pytest
import torch

from helpers import K, build, calibrate_bn_hip, load_det, zero_dropout
from oracle.det_init import det_input, det_labels

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_cfg1_r18_ppm_480p_frame_against_oracle(dev):
    from oracle import np_models as NM
    from oracle import np_ops as O

    O.set_dtype(np.float32)
    mod = build("seg", "resnet18dilated", "ppm_deepsup", 512)
    sd = load_det(mod)
    mod.to(dev).eval()
    img = det_input("cfg1", (1, 3, 480, 853))
    store = {}
    h = mod.decoder.conv_last_.register_forward_hook(lambda m, i, o: store.__setitem__("l", o.float().cpu().numpy()))
    with torch.no_grad():
        probs = mod({"img_data": _t(img, dev), "seg_label": torch.zeros(1, 1, 480, 853, device=dev)},
                    segSize=(480, 853))
    h.remove()
    assert tuple(probs.shape) == (1, K, 480, 853)
    P = NM.Params({k: v.copy() for k, v in sd.items()}, train_params=False)
    feats = NM.resnet_dilated(P, O.Var(img), "resnet18", "encoder.", False)
    conv5 = feats[-1]
    pooled = [O.adaptive_avg_pool2d(conv5, s) for s in (1, 2, 3, 6)]  # 60x107 map: overlapping bins
    logits = NM._head(P, NM._ppm_concat(P, conv5, pooled, "decoder.ppm.", 1, 2, False), "decoder.conv_last_", False)
    assert logits.shape == store["l"].shape == (1, K, 60, 107)
    err = np.abs(store["l"] - logits.v).max()
    assert err < 1e-3, err
    ref = O.softmax(O.interpolate_bilinear(logits, (480, 853)), 1).v
    got = probs.float().cpu().numpy()
    assert np.abs(got - ref).max() < 1e-3
    s = np.sort(ref, axis=1)
    decisive = (np.log(s[:, -1]) - np.log(s[:, -2])) > 2e-3
    assert ((got.argmax(1) != ref.argmax(1)) & decisive).sum() == 0
    assert np.abs(got.sum(1) - 1).max() < 1e-5


def test_cfg2_r101_ppm_480p_frame_properties(dev):
    mod = build("seg", "resnet101dilated", "ppm_deepsup", 2048)
    load_det(mod)
    mod.to(dev)
    img = _t(det_input("cfg2", (1, 3, 480, 853)), dev)
    two = torch.cat([img, img.flip(-1)], 0)
    lab2 = _t(det_labels("cfg2", (2, 1, 480, 853), K), dev)
    calibrate_bn_hip(mod, lambda: mod({"img_data": two, "seg_label": lab2}))
    mod.eval()
    with torch.no_grad():
        p1 = mod({"img_data": img, "seg_label": torch.zeros(1, 1, 480, 853, device=dev)}, segSize=(480, 853))
        p2 = mod({"img_data": img, "seg_label": torch.zeros(1, 1, 480, 853, device=dev)}, segSize=(480, 853))
    assert tuple(p1.shape) == (1, K, 480, 853)
    assert torch.isfinite(p1).all() and (p1 >= 0).all()
    assert (p1.sum(1) - 1).abs().max().item() < 1e-5
    assert torch.equal(p1, p2), "inference is deterministic"
    assert p1.max().item() < 0.999, "calibrated BN statistics keep the random-weight logits in a sane range"


@pytest.mark.parametrize("kind", ["clip_ocr"])
def test_cfg4_tcb_ocr_train_step_properties(dev, kind):
    mod = build(kind, "resnet101dilated").to(dev)
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
        return loss.item(), {k: p.grad.norm().item() for k, p in mod.named_parameters()}

    l1, g1 = step(1.0)
    l2, g2 = step(2.0)
    assert math.isfinite(l1) and abs(l1 - l2) < 1e-5 * abs(l1)
    assert 0.5 * 1.4 * math.log(K) < l1 < 3 * 1.4 * math.log(K)
    gmax = max(g2.values())
    for k in g1:
        assert math.isfinite(g1[k]), k
        # conv biases in front of a train-mode BN have an exactly-zero gradient (pure rounding noise): absolute floor
        assert abs(g2[k] - 2 * g1[k]) <= 1e-3 * max(g2[k], 1e-6 * gmax), k


def test_cfg5_nonlocal3d_t7_and_netwarp_fullsize_properties(dev):
    """cfg 5 pieces: Non_local3d over T=7 frames (N = 7*60*60 = 25 200 positions, a 2.5 GB affinity per sample) and
    NetWarp (R101, T=2) with a synthetic flow field, at 479x479."""
    S = 479
    g = torch.Generator().manual_seed(5)
    mod = build("nonlocal3d", "resnet101dilated").to(dev)
    mod.train()
    imgs = [torch.randn(2, 3, S, S, generator=g).to(dev) for _ in range(7)]   # B = 2 clips, as BASELINE.json cfg 5b
    labs = [torch.randint(0, K, (2, 1, S, S), generator=g).float().to(dev) for _ in range(7)]
    loss, acc = mod({"clipimgs_data": imgs, "cliplabels_data": labs})
    loss.backward()
    assert math.isfinite(loss.item())
    assert all(torch.isfinite(p.grad).all() for p in mod.parameters() if p.grad is not None)
    assert mod.nonlocalblock.theta.weight.grad is not None
    del mod, loss
    torch.cuda.empty_cache()

    class Flow(torch.nn.Module):
        def forward(self, a, b, iters=20, test_mode=True):
            n, _, h, w = a.shape
            return None, (torch.randn(n, 2, h, w, device=a.device) * 1.9 - 0.7).clamp(-10, 10)

    nw = build("netwarp", "resnet101dilated", flow_net=Flow()).to(dev)
    nw.train()
    cur, prev = torch.randn(2, 3, S, S, generator=g).to(dev), torch.randn(2, 3, S, S, generator=g).to(dev)
    lab = torch.randint(0, K, (2, 1, S, S), generator=g).float().to(dev)
    loss, acc = nw({"img_data": cur, "seg_label": lab, "clipimgs_data": [prev], "cliplabels_data": []})
    loss.backward()
    assert math.isfinite(loss.item())
    assert torch.isfinite(nw.w0_1.grad).all() and torch.isfinite(nw.flowcnn.conv1[0].weight.grad).all()


def test_cfg5a_nonlocal2d_train_step_fullsize_properties(dev):
    """cfg 5a: per-frame SegmentationModule with the Non_local2d decoder, R101, B = 2 at 479x479 (3 600 positions)."""
    S = 479
    g = torch.Generator().manual_seed(6)
    mod = build("seg", "resnet101dilated", decoder="nonlocal2d", deep_sup_scale=None).to(dev)
    mod.train()
    img = torch.randn(2, 3, S, S, generator=g).to(dev)
    lab = torch.randint(0, K, (2, 1, S, S), generator=g).float().to(dev)
    loss, acc = mod({"img_data": img, "seg_label": lab})
    loss.backward()
    assert math.isfinite(loss.item()) and 0.0 <= float(acc) <= 1.0
    grads = [p.grad for p in mod.parameters() if p.grad is not None]
    assert len(grads) > 300 and all(torch.isfinite(gr).all() for gr in grads)


def test_non_local_dot_values_at_cfg5b_size(dev):
    """The fused affinity kernel at the T = 7 size (B = 2, N = 25 200 positions, C = 128; reference
    models/non_local.py:105-143 would hold a 2.5 GB N x N tensor per sample): sampled output rows against float64
    theta_i . phi^T . g / N, and the three gradients - which run through the same kernel with permuted operands -
    against their float64 rows."""
    from cvpr2021_vspw_implement_amd import ops

    B, N, C = 2, 25200, 128
    g = torch.Generator().manual_seed(8)
    q, k, v, dy = (torch.randn(B, N, C, generator=g).to(dev).requires_grad_(i < 3) for i in range(4))
    out = ops.non_local_dot(q, k, v, 1.0 / N)
    out.backward(dy)
    rows = torch.randint(0, N, (48,), generator=g).to(dev)
    q64, k64, v64, d64 = (t.detach().double() for t in (q, k, v, dy))
    for b in range(B):
        want = (q64[b, rows] @ k64[b].T) @ v64[b] / N
        err = (out[b, rows].double() - want).abs().max().item()
        assert err <= 2e-5 * max(want.abs().max().item(), 1e-3), ("y", b, err)
        # d theta_i = (dy_i . g^T) phi / N ; d phi_j = (g_j . dy^T) theta / N ; d g_j = (phi_j . theta^T) dy / N
        for name, got, a, m1, m2 in (("dq", q.grad, d64, v64, k64), ("dk", k.grad, v64, d64, q64),
                                     ("dv", v.grad, k64, q64, d64)):
            want = (a[b, rows] @ m1[b].T) @ m2[b] / N
            err = (got[b, rows].double() - want).abs().max().item()
            assert err <= 2e-5 * max(want.abs().max().item(), 1e-3), (name, b, err)


@pytest.mark.parametrize("kind", ["clip_psp", "clip_ocr"])
def test_bench_workload_values_against_live_oracle(dev, kind):
    """VALUE-level parity on the metric's own configuration (BASELINE.json configs[2] / [3]): ResNet-101 dilated TCB-PSP
    / TCB-OCR, T=5 frames, B=2 clips, train step (loss, pixel accuracy, gradient of every parameter) against the numpy
    oracle evaluated live in FLOAT64 - the oracle that tests/test_oracle_golden.py pins on the reference to 1e-9 - at
    239x239 crops (30x30 feature maps, BatchNorm populations of 9 000).

    What fp32 allows here was measured, not assumed (tools/diag/benchval.py): the SAME oracle run in float32 - i.e. the
    reference's arithmetic type - misses its own float64 gradients by up to 1.3e-2 on a parameter's norm (RMS 2.3e-3)
    and by 6 % in relative L2 on the stem weights: 100 random-weight layers amplify fp32 rounding ~1e5-fold.  The HIP
    path is therefore gated against that measured floor, evaluated in the same test: loss 2e-4 and accuracy 2e-3
    absolute; per-parameter gradient-norm error RMS and maximum within a factor of the float32 oracle's, aggregate
    norm-vector error bounded, full-tensor relative L2 error <= 2 x the float32 oracle's.
    TCB-PSP (factors 2 / 3, aggregate 5e-3): observed max 2.1e-2, RMS 2.9e-3, aggregate 1.3e-3.
    TCB-OCR (factors 5 / 3.5, aggregate 3e-2): its 124 object-context vectors per clip pass through BatchNorm/ReLU
    stacks with populations of 248 and are attended to by every pixel, so ONE rounding-level ReLU decision there moves
    every upstream gradient norm coherently by 1-2 %: the same kernels with the 3x3 K loop in tap-outer order give RMS
    6.5e-3 / aggregate 6.0e-3, in tap-inner order RMS 1.25e-2 / aggregate 1.7e-2 (median 8.7e-3 vs 2.5e-3; float32
    oracle 3.3e-3 / 2.0e-3) - a different realisation of the same noise, not a different accuracy (kernel-level tests
    hold both orders to 1e-4).  The gate keeps what it can discriminate: a plumbing error (a missing term, a wrong
    scale) moves norms by tens of percent."""
    import time

    from oracle import np_models as NM
    from oracle import np_ops as O

    T, B, S = 5, 2, 239
    mod = build(kind, "resnet101dilated", args={"clip_num": T})
    sd = load_det(mod)
    zero_dropout(mod)
    mod.to(dev).train()
    imgs = [det_input("benchval:%s:%d" % (kind, t), (B, 3, S, S), seed=11) for t in range(T)]
    labs = [det_labels("benchval:%s:%d" % (kind, t), (B, 1, S, S), K, seed=11) for t in range(T)]
    ti = [_t(a, dev) for a in imgs]
    tl = [_t(a, dev) for a in labs]
    loss, acc = mod({"img_data": ti[-1], "seg_label": tl[-1], "clipimgs_data": ti[:-1], "cliplabels_data": tl[:-1]})
    loss.backward()
    g = {k: p.grad.detach().double().cpu().numpy() for k, p in mod.named_parameters() if p.grad is not None}
    fn = NM.clip_psp if kind == "clip_psp" else NM.clip_ocr
    res = {}
    try:
        for dt in (np.float64, np.float32):
            t0 = time.time()
            O.set_dtype(dt)
            P = NM.Params({k: v.astype(dt) for k, v in sd.items()}, train_params=True)
            ol, oa = fn(P, "resnet101", [a.astype(dt) for a in imgs], labs, True)
            O.tape().backward(ol)
            res[dt] = (float(np.asarray(ol.v).reshape(())), oa, {k: v.astype(np.float64) for k, v in P.grads().items()})
            print("oracle %s: %.1f s" % (dt.__name__, time.time() - t0))
    finally:
        O.set_dtype(np.float32)
    l64, a64, g64 = res[np.float64]
    _, _, g32 = res[np.float32]
    assert abs(loss.item() - l64) < 2e-4 * abs(l64), (loss.item(), l64)
    assert abs(acc.item() - a64) < 2e-3
    norms = {k: float(np.linalg.norm(v)) for k, v in g64.items()}
    scale = max(norms.values())
    e_hip, e_or, num, den = [], [], 0.0, 0.0
    for k, r in norms.items():
        assert k in g, k
        n = float(np.linalg.norm(g[k]))
        num += (n - r) ** 2
        den += r ** 2
        e_hip.append(abs(n - r) / max(r, 1e-3 * scale))
        e_or.append(abs(float(np.linalg.norm(g32[k])) - r) / max(r, 1e-3 * scale))
    e_hip, e_or = np.array(e_hip), np.array(e_or)
    rms = lambda e: float(np.sqrt((e ** 2).mean()))  # noqa: E731
    print("per-parameter gradient-norm error: hip max %.3e rms %.3e | float32 oracle max %.3e rms %.3e; aggregate %.3e"
          % (e_hip.max(), rms(e_hip), e_or.max(), rms(e_or), (num / den) ** 0.5))
    f_rms, f_max, agg = {"clip_psp": (2.0, 3.0, 5e-3), "clip_ocr": (5.0, 3.5, 3e-2)}[kind]
    assert rms(e_hip) <= f_rms * rms(e_or), (rms(e_hip), rms(e_or))
    assert e_hip.max() <= f_max * e_or.max(), (e_hip.max(), e_or.max())
    assert (num / den) ** 0.5 < agg
    for k in ("encoder.conv1.weight", "encoder.layer3.22.conv2.weight", "encoder.layer4.2.conv3.weight"):
        rel = np.linalg.norm(g[k] - g64[k]) / np.linalg.norm(g64[k])
        rel32 = np.linalg.norm(g32[k] - g64[k]) / np.linalg.norm(g64[k])
        assert rel <= 2.0 * rel32 + 1e-3, (k, rel, rel32)
