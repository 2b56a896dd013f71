"""BASELINE.json configurations at their full sizes on the MI355X.
cfg 1 (resnet18dilated + ppm_deepsup, one 480x853 frame) is small enough for the numpy oracle to be evaluated live and
compared value by value; the larger configurations are checked through size-independent properties (probabilities
normalised, finite loss ~ log K at random init, finite gradients on every parameter, loss linear in the incoming
gradient, eval deterministic)."""
import math
import os

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


# ------------------------------------------------------------------------------------------------------------------
# VALUE-level parity on the metric's own configuration (BASELINE.json configs[2] / [3]): ResNet-101 dilated TCB-PSP /
# TCB-OCR, T = 5 frames, B = 2 clips, one training step at 239x239 crops (30x30 feature maps, BatchNorm populations of
# 9 000) against the numpy oracle that tests/test_oracle_golden.py pins on the reference to 1e-9.
#
# Two questions, two tests; every gate is "no worse than 1.5x the float32 oracle put through the same procedure":
#  (1) Is the HIP backward the gradient of the same function?  ReLU and max-pool are the only non-smooth steps; a unit
#      whose pre-activation is a rounding error away from 0 flips between two implementations and moves EVERY upstream
#      gradient (this made round 2's free comparison noisy: 3-7 % relative L2 between the reference's own float32 and
#      float64 runs).  So the HIP forward's decisions (ops.record_decisions: every ReLU mask, the max-pool taps) are
#      injected into the float64 oracle (oracle.np_ops.set_decisions): both then differentiate the same
#      piecewise-linear branch and differ by smooth rounding only (tests/test_decisions_cpu.py pins the mechanism).
#      What is left is NOT small - 100 random-weight layers amplify float32 rounding 1e4-fold in the forward pass
#      (measured with tools/diag/pinned.py: 1e-7 after the first convolution, 6e-4 at the encoder output, identical for
#      HIP and the float32 oracle) - so it is split in two:
#        * the total error of a step must not exceed 1.5x that of the float32 ORACLE put through the same procedure
#          (its own decisions, same inputs), with the oracle's GEMMs in the accumulation order of a matrix-core k-loop
#          (np_ops.set_gemm("sequential"): OpenBLAS, like the ATen kernels behind the reference, blocks the k loop,
#          which is worth a factor 3 in rounding noise per GEMM at K = 4 608 ... 36 864; with it the oracle's forward
#          error tracks HIP's to 2 % at every depth).  Two realisations each (step A on the inputs, step B on inputs
#          perturbed by 1e-7 relative), the larger one counts on both sides: one oracle realisation differs from the
#          next by up to 1.5x on its own;
#        * the part of the error that REPRODUCES over the two realisations, sqrt(<e_A, e_B>) / |g| - what a wrong
#          formula, a missing term or a mis-scaled factor would produce, but also every legitimate deterministic
#          difference in float32 evaluation order, amplified like the rest (the oracle's own A/B errors correlate at
#          0.24-0.31, HIP's at 0.40) - must stay below max(1e-3, 1.5x the oracle's reproducible part): the rule of
#          helpers.logit_tol, 1e-3 unless the reference's arithmetic type is itself further than that from exact.
#      Measured (R101, T=5, B=2, 239^2; per-parameter relative L2, median / p99 / max over the 340 / 358 tensors):
#        TCB-PSP total HIP 1.6e-3 / 2.9e-3 / 3.7e-3, oracle 1.3e-3 / 2.3e-3 / 2.9e-3; reproducible 8.5e-4 / 1.5e-3 /
#        2.0e-3 vs 5.3e-4 / 1.1e-3 / 1.5e-3.  TCB-OCR total 3.4e-3 / 7.2e-3 / 7.8e-3 vs 3.4e-3 / 7.6e-3 / 8.6e-3;
#        reproducible 2.0e-3 / 4.4e-3 / 5.2e-3 vs 1.7e-3 / 4.0e-3 / 5.5e-3.  (Free comparison, for scale: 3e-2 ... 7e-2.)
#  (2) Is the HIP step, compared freely against float64, one more realisation of float32 noise?  An ENSEMBLE of nine
#      float32 oracle runs (the same sequential-order arithmetic; inputs perturbed by 1e-7 relative = one-ulp flips)
#      measures that noise; HIP must lie within 1.5x of its worst member on every statistic.
@pytest.fixture(scope="module", params=["clip_psp", "clip_ocr"])
def bench_case(request, dev, tmp_path_factory):
    """Two HIP training steps (inputs A, and B = A perturbed by 1e-7 relative) with their decisions recorded + every
    oracle evaluation the two tests below need, run as parallel worker processes (module scope: pytest runs both tests
    of a kind back to back, then the ~3 GB of gradients of a case are dropped)."""
    import time

    from cvpr2021_vspw_implement_amd import ops
    from helpers import run_oracle_jobs
    from oracle_worker import pack_decisions, perturbed

    kind = request.param
    tmp = tmp_path_factory.mktemp("bench_" + kind)
    T, B, S = 5, 2, 239
    imgs = [det_input("benchval:%s:%d" % (kind, t), (B, 3, S, S), seed=11) for t in range(T)]
    labs = [det_labels("benchval:%s:%d" % (kind, t), (B, 1, S, S), K, seed=11) for t in range(T)]
    imgs_b = [perturbed(a, 1 * 100 + t, 1e-7) for t, a in enumerate(imgs)]  # = the worker's perturb_seed 1
    hip = {}
    for tag, ims in (("A", imgs), ("B", imgs_b)):
        mod = build(kind, "resnet101dilated", args={"clip_num": T})
        load_det(mod)
        zero_dropout(mod)
        mod.to(dev).train()
        ti = [_t(a, dev) for a in ims]
        tl = [_t(a, dev) for a in labs]
        taps = []
        ops.record_decisions(taps)
        try:
            loss, acc = mod({"img_data": ti[-1], "seg_label": tl[-1], "clipimgs_data": ti[:-1],
                             "cliplabels_data": tl[:-1]})
        finally:
            ops.record_decisions(None)
        loss.backward()
        torch.cuda.synchronize()
        g = {k: p.grad.detach().double().cpu().numpy() for k, p in mod.named_parameters() if p.grad is not None}
        bn_name = {id(p): n[:-len(".weight")] for n, p in mod.named_parameters() if n.endswith(".weight")}
        store = {}
        for what, key, t in taps:
            if what == "relu":
                store.setdefault(bn_name[id(key)], []).append((t > 0).cpu().numpy())
            else:  # max-pool taps [n, oh, ow, c] -> the oracle's [n, c, oh, ow]
                store.setdefault("encoder.maxpool", []).append(
                    t.permute(0, 3, 1, 2).contiguous().cpu().numpy().astype(np.int8))
        pack_decisions(store, str(tmp / ("dec_hip%s.npz" % tag)))
        hip[tag] = dict(loss=loss.item(), acc=acc.item(), g=g)
        del taps, store, mod, loss, acc, ti, tl
        torch.cuda.empty_cache()
    base = dict(kind=kind, arch="resnet101", T=T, B=B, S=S)
    m32, m64 = 24.0, 48.0  # GB per worker at this size (generous: measured peak RSS < 20 / 40), for the memory cap
    seq = dict(base, dtype="f32", gemm="sequential", mem_gb=m32)
    f64 = dict(base, dtype="f64", full_grads=True, mem_gb=m64)
    o = lambda name: str(tmp / name)  # noqa: E731
    jobs = [dict(f64, out=o("free64.npz")),                                                                     # 0
            dict(f64, decisions="inject", decisions_path=o("dec_hipA.npz"), out=o("inj_hipA.npz")),              # 1
            dict(f64, decisions="inject", decisions_path=o("dec_hipB.npz"), out=o("inj_hipB.npz"), perturb_seed=1),
            dict(seq, full_grads=True, decisions="record", decisions_path=o("dec_seqA.npz"), out=o("seqA.npz")),  # 3
            dict(seq, full_grads=True, decisions="record", decisions_path=o("dec_seqB.npz"), out=o("seqB.npz"),
                 perturb_seed=1),                                                                               # 4
            dict(f64, decisions="inject", decisions_path=o("dec_seqA.npz"), out=o("inj_seqA.npz"), after=3),      # 5
            dict(f64, decisions="inject", decisions_path=o("dec_seqB.npz"), out=o("inj_seqB.npz"), after=4,
                 perturb_seed=1)]                                                                               # 6
    jobs += [dict(seq, perturb_seed=i, out=o("ens%d.npz" % i)) for i in range(2, 9)]                           # 7..13
    t0 = time.time()
    res = run_oracle_jobs(jobs, str(tmp))
    print("oracle: %d worker processes, %.0f s wall (slowest %.0f s)"
          % (len(jobs), time.time() - t0, max(float(r["seconds"]) for r in res)))
    return dict(kind=kind, hip=hip, free64=res[0], inj_hip={"A": res[1], "B": res[2]}, seq={"A": res[3], "B": res[4]},
                inj_seq={"A": res[5], "B": res[6]}, ens=[res[3], res[4]] + res[7:])


def _errors(get, ref, names, scale):
    """per-parameter (error tensor, reference norm floored at 1e-3 of the largest parameter-gradient norm: conv biases
    in front of a training-mode BatchNorm have a zero gradient in exact arithmetic)"""
    out = {}
    for n in names:
        r = ref["g:" + n].astype(np.float64)
        out[n] = (get(n) - r, max(float(np.linalg.norm(r)), 1e-3 * scale))
    return out


# measured on the MI355X box (r05): per-parameter relative L2 of the HIP gradients against the float64 oracle on the SAME
# piecewise-linear branch - see the print of the test; the gates sit at ~2x those figures, a factor 10 below what the
# free comparison shows (3e-2 ... 7e-2), so a smooth error in ONE backward kernel cannot hide behind ReLU flips
# (r05, MI355X: median 3.91e-04, p99 4.68e-04, max 8.66e-04 at ppm_conv.ppm.0.0.weight; float64 oracle job 28 s)
PINNED_GATE = {"median": 8e-4, "p99": 1.0e-3, "max": 2e-3}


def test_pinned_decision_gradients_one_head_default_suite(dev, tmp_path):
    """Question (1) of the block above in the DEFAULT suite (the two-realisation, oracle-relative form stays opt-in): one
    TCB-PSP training step (ResNet-50, T=3, B=2, 239x239), the HIP forward's decisions (every ReLU mask, the max-pool taps)
    injected into ONE float64 oracle evaluation, every parameter gradient compared on that branch.  What remains is
    smooth float32 rounding, amplified by the random-weight network: absolute gates, calibrated on the measured
    figures."""
    import time

    from cvpr2021_vspw_implement_amd import ops
    from helpers import run_oracle_jobs
    from oracle_worker import pack_decisions

    kind, T, B, S = "clip_psp", 3, 2, 239
    imgs = [det_input("benchval:%s:%d" % (kind, t), (B, 3, S, S), seed=11) for t in range(T)]
    labs = [det_labels("benchval:%s:%d" % (kind, t), (B, 1, S, S), K, seed=11) for t in range(T)]
    mod = build(kind, "resnet50dilated", args={"clip_num": T})
    load_det(mod)
    zero_dropout(mod)
    mod.to(dev).train()
    ti, tl = [_t(a, dev) for a in imgs], [_t(a, dev) for a in labs]
    taps = []
    ops.record_decisions(taps)
    try:
        loss, acc = mod({"img_data": ti[-1], "seg_label": tl[-1], "clipimgs_data": ti[:-1], "cliplabels_data": tl[:-1]})
    finally:
        ops.record_decisions(None)
    loss.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().double().cpu().numpy() for k, p in mod.named_parameters() if p.grad is not None}
    bn_name = {id(p): n[:-len(".weight")] for n, p in mod.named_parameters() if n.endswith(".weight")}
    store = {}
    for what, key, t in taps:
        if what == "relu":
            store.setdefault(bn_name[id(key)], []).append((t > 0).cpu().numpy())
        else:
            store.setdefault("encoder.maxpool", []).append(t.permute(0, 3, 1, 2).contiguous().cpu().numpy().astype(np.int8))
    dec = str(tmp_path / "dec.npz")
    pack_decisions(store, dec)
    t0 = time.time()
    res = run_oracle_jobs([dict(kind=kind, arch="resnet50", T=T, B=B, S=S, dtype="f64", full_grads=True, mem_gb=24.0,
                                decisions="inject", decisions_path=dec, out=str(tmp_path / "inj.npz"))], str(tmp_path))[0]
    names = [str(n) for n in res["names"]]
    assert set(names) <= set(g), sorted(set(names) - set(g))[:5]
    scale = float(res["norms"].max())
    rel = np.array([np.linalg.norm(g[n] - res["g:" + n].astype(np.float64))
                    / max(float(np.linalg.norm(res["g:" + n].astype(np.float64))), 1e-3 * scale) for n in names])
    w = int(rel.argmax())
    stats = {"median": float(np.median(rel)), "p99": float(np.percentile(rel, 99)), "max": float(rel.max())}
    print("TCB-PSP R50 T=3 239^2, decisions pinned: loss hip %.7f float64 %.7f; per-parameter relative L2 of the gradients "
          "median %.2e p99 %.2e max %.2e (%s); oracle %.0f s"
          % (loss.item(), float(res["loss"]), stats["median"], stats["p99"], stats["max"], names[w], time.time() - t0))
    assert abs(loss.item() - float(res["loss"])) < 2e-5 * abs(float(res["loss"]))
    for what, lim in PINNED_GATE.items():
        assert stats[what] <= lim, (what, stats[what], lim)


@pytest.mark.live_oracle
def test_bench_workload_gradients_with_pinned_decisions(bench_case):
    """Question (1) above.  Loss of each HIP step within 2e-5 relative of its decision-injected float64 run."""
    c, kind = bench_case, bench_case["kind"]
    names = [str(n) for n in c["inj_hip"]["A"]["names"]]
    assert set(names) <= set(c["hip"]["A"]["g"]), sorted(set(names) - set(c["hip"]["A"]["g"]))[:5]
    scale = float(c["inj_hip"]["A"]["norms"].max())
    eh, eo = {}, {}
    for tag in ("A", "B"):
        inj = c["inj_hip"][tag]
        assert abs(c["hip"][tag]["loss"] - float(inj["loss"])) < 2e-5 * abs(float(inj["loss"]))
        eh[tag] = _errors(lambda n: c["hip"][tag]["g"][n], inj, names, scale)
        eo[tag] = _errors(lambda n: c["seq"][tag]["g:" + n].astype(np.float64), c["inj_seq"][tag], names, scale)
    rel = lambda e: np.array([np.linalg.norm(e[n][0]) / e[n][1] for n in names])  # noqa: E731
    rep = lambda e: np.array([max(float(np.vdot(e["A"][n][0], e["B"][n][0])), 0.0) ** 0.5  # noqa: E731
                              / (e["A"][n][1] * e["B"][n][1]) ** 0.5 for n in names])
    tot_h = np.maximum(rel(eh["A"]), rel(eh["B"]))
    tot_o = np.maximum(rel(eo["A"]), rel(eo["B"]))
    rep_h, rep_o = rep(eh), rep(eo)
    w = int(rep_h.argmax())
    st = lambda v: "median %.2e p99 %.2e max %.2e" % (np.median(v), np.percentile(v, 99), v.max())  # noqa: E731
    print("%s, decisions pinned, per-parameter relative L2 against float64:\n  total        HIP %s | float32 oracle %s\n"
          "  reproducible HIP %s (%s) | float32 oracle %s"
          % (kind, st(tot_h), st(tot_o), st(rep_h), names[w], st(rep_o)))
    diag = os.environ.get("VSPW_DIAG_DIR")
    if diag and os.path.isdir(diag):  # per-parameter table for offline inspection (tools/diag)
        cos_o = np.array([float(np.vdot(eo["A"][n][0], eo["B"][n][0]))
                          / max(np.linalg.norm(eo["A"][n][0]) * np.linalg.norm(eo["B"][n][0]), 1e-300) for n in names])
        cos_h = np.array([float(np.vdot(eh["A"][n][0], eh["B"][n][0]))
                          / max(np.linalg.norm(eh["A"][n][0]) * np.linalg.norm(eh["B"][n][0]), 1e-300) for n in names])
        np.savez(os.path.join(diag, "r03_pinned_%s.npz" % kind), names=np.array(names), hipA=rel(eh["A"]),
                 hipB=rel(eh["B"]), orA=rel(eo["A"]), orB=rel(eo["B"]), rep_h=rep_h, rep_o=rep_o, cos_h=cos_h, cos_o=cos_o)
    for what, f in (("median", np.median), ("p99", lambda v: np.percentile(v, 99)), ("max", np.max)):
        assert f(tot_h) <= 1.5 * f(tot_o), ("total", what, f(tot_h), f(tot_o))
        assert f(rep_h) <= max(1e-3, 1.5 * f(rep_o)), ("reproducible", what, f(rep_h), f(rep_o))


@pytest.mark.live_oracle
def test_bench_workload_against_fp32_ensemble(bench_case):
    """Question (2) above.  Free comparison (every implementation takes its own decisions) against the float64 oracle:
    loss 2e-4 relative, pixel accuracy 2e-3; per-parameter gradient-norm error - RMS, maximum, and the relative error of
    the whole vector of norms - no more than 1.5x the WORST of nine float32-oracle runs (the unperturbed one and eight
    on inputs perturbed by 1e-7 relative), each measured against the same float64 run."""
    c, kind = bench_case, bench_case["kind"]
    ref, hip = c["free64"], c["hip"]["A"]
    names = [str(n) for n in ref["names"]]
    r = ref["norms"].astype(np.float64)
    scale = float(r.max())
    assert abs(hip["loss"] - float(ref["loss"])) < 2e-4 * abs(float(ref["loss"])), (hip["loss"], float(ref["loss"]))
    assert abs(hip["acc"] - float(ref["acc"])) < 2e-3

    def stats(norms):
        e = np.abs(norms - r) / np.maximum(r, 1e-3 * scale)
        return float(np.sqrt((e ** 2).mean())), float(e.max()), float(np.sqrt(((norms - r) ** 2).sum() / (r ** 2).sum()))

    got = stats(np.array([np.linalg.norm(hip["g"][n]) for n in names]))
    members = []
    for m in c["ens"]:
        assert [str(n) for n in m["names"]] == names
        members.append(stats(m["norms"].astype(np.float64)))
    members = np.array(members)
    print("%s free comparison, gradient-norm error (rms, max, aggregate): HIP %.3e %.3e %.3e | float32 ensemble of %d: "
          "median %s worst %s" % ((kind,) + got + (len(members), np.median(members, 0).round(5), members.max(0).round(5))))
    for i, what in enumerate(("rms", "max", "aggregate")):
        assert got[i] <= 1.5 * members[:, i].max(), (what, got[i], members[:, i])


@pytest.mark.skipif(os.environ.get("VSPW_FULLSIZE_479") != "1",
                    reason="opt-in (VSPW_FULLSIZE_479=1): three oracle evaluations at 479x479, ~15 minutes")
@pytest.mark.parametrize("kind", ["clip_psp", "clip_ocr"])
def test_pinned_decisions_at_the_metrics_own_crop_size(dev, kind, tmp_path):
    """The pinned-decision comparison of test_bench_workload_gradients_with_pinned_decisions at EXACTLY the bench
    workload (R101, T = 5, B = 2, 479x479: 60x60 feature maps, BatchNorm populations of 36 000) - one rounding
    realisation, total error only.  Not in the default suite (the oracle costs 4x the 239^2 evaluations); the log of the
    last run is committed as profiles/r03_pinned_479.log."""
    import time

    from cvpr2021_vspw_implement_amd import ops
    from helpers import hip_decision_store, run_oracle_jobs
    from oracle_worker import pack_decisions

    T, B, S = 5, 2, 479
    mod = build(kind, "resnet101dilated", args={"clip_num": T})
    load_det(mod)
    zero_dropout(mod)
    mod.to(dev).train()
    imgs = [det_input("benchval:%s:%d" % (kind, t), (B, 3, S, S), seed=11) for t in range(T)]
    labs = [det_labels("benchval:%s:%d" % (kind, t), (B, 1, S, S), K, seed=11) for t in range(T)]
    ti, tl = [_t(a, dev) for a in imgs], [_t(a, dev) for a in labs]
    taps = []
    ops.record_decisions(taps)
    try:
        loss, acc = mod({"img_data": ti[-1], "seg_label": tl[-1], "clipimgs_data": ti[:-1], "cliplabels_data": tl[:-1]})
    finally:
        ops.record_decisions(None)
    loss.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().double().cpu().numpy() for k, p in mod.named_parameters() if p.grad is not None}
    pack_decisions(hip_decision_store(mod, taps), str(tmp_path / "dec_hip.npz"))
    hip_loss = loss.item()
    del taps, mod, loss, ti, tl
    torch.cuda.empty_cache()
    base = dict(kind=kind, arch="resnet101", T=T, B=B, S=S, full_grads=True)
    o = lambda n: str(tmp_path / n)  # noqa: E731
    t0 = time.time()
    inj_h, seq, inj_s = run_oracle_jobs(
        [dict(base, dtype="f64", decisions="inject", decisions_path=o("dec_hip.npz"), out=o("inj_hip.npz"), mem_gb=200.0),
         dict(base, dtype="f32", gemm="sequential", decisions="record", decisions_path=o("dec_seq.npz"), out=o("seq.npz"),
              mem_gb=100.0),
         dict(base, dtype="f64", decisions="inject", decisions_path=o("dec_seq.npz"), out=o("inj_seq.npz"), after=1,
              mem_gb=200.0)], str(tmp_path), parallel=2, threads=64)
    names = [str(n) for n in inj_h["names"]]
    scale = float(inj_h["norms"].max())
    eh = _errors(lambda n: g[n], inj_h, names, scale)
    eo = _errors(lambda n: seq["g:" + n].astype(np.float64), inj_s, names, scale)
    rel = lambda e: np.array([np.linalg.norm(e[n][0]) / e[n][1] for n in names])  # noqa: E731
    th, to = rel(eh), rel(eo)
    print("%s 479x479, decisions pinned (oracle %.0f s): loss hip %.7f float64 %.7f; per-parameter relative L2: HIP median "
          "%.2e p99 %.2e max %.2e | float32 oracle median %.2e p99 %.2e max %.2e"
          % (kind, time.time() - t0, hip_loss, float(inj_h["loss"]), np.median(th), np.percentile(th, 99), th.max(),
             np.median(to), np.percentile(to, 99), to.max()))
    assert abs(hip_loss - float(inj_h["loss"])) < 2e-5 * abs(float(inj_h["loss"]))
    for what, f in (("median", np.median), ("p99", lambda v: np.percentile(v, 99)), ("max", np.max)):
        assert f(th) <= max(1e-3, 1.5 * f(to)), (what, f(th), f(to))
