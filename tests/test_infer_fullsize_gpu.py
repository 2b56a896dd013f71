"""Inference at frame size, value by value, through the path test_clip2.py actually runs (reference
test_clip2.py:28-89, models/clip_psp.py:189-196, models/clip_ocr.py:170-177): ResNet-101 dilated, one 480x853 target
frame (+ 3 context frames for the TCB heads), softmax probabilities at segSize = (480, 853).

Without autograd the HIP path is NOT the training path: every conv + eval-mode BatchNorm (+ residual) + ReLU is one
launch on weights with the BatchNorm scale folded in (ops._conv_bn_folded, vspw_bn_fold_weights, vspw_conv2d_fwd_ex).
Checked here, for cfg 2 (per-frame PSPNet), TCB-PSP and TCB-OCR:
  * logits at 60x107 against the numpy oracle evaluated live on the same weights and calibrated running statistics, in
    float32 and float64: |hip - ref| <= max(1e-3, 2 x |ref32 - ref64|) (helpers.logit_tol's rule: north_star's 1e-3,
    unless the reference's own arithmetic type is further than that from exact);
  * probabilities at 480x853 within 1e-3 of softmax(bilinear(oracle logits)), normalised, arg-max identical wherever the
    oracle's top-2 log-probability gap exceeds 2 x tol;
  * the unfolded path (ops.set_inference_folding(False): conv, then BatchNorm apply) meets the same logit tolerance,
    and folded vs unfolded agree to 1e-5 relative L2 at the output of layer1 (10 convolutions deep) - further down the
    two float32 evaluation orders drift apart like any two float32 evaluations of this network do (measured at the
    logits: 2e-3, the size of |ref32 - ref64|)."""
import numpy as np
import pytest
import torch

from helpers import K, build, calibrate_bn_hip, load_det, run_oracle_jobs
from oracle.det_init import det_input

pytestmark = pytest.mark.gpu
H, W = 480, 853


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.live_oracle
@pytest.mark.parametrize("kind", ["seg_ppm", "clip_psp", "clip_ocr"])
def test_480p_inference_values_through_the_folded_path(dev, kind, tmp_path):
    from cvpr2021_vspw_implement_amd import ops
    from oracle import np_ops as O

    T = 1 if kind == "seg_ppm" else 4
    if kind == "seg_ppm":
        mod = build("seg", "resnet101dilated", "ppm_deepsup", 2048)
        tap = lambda m: m.decoder.conv_last_  # noqa: E731
    else:
        mod = build(kind, "resnet101dilated", args={"clip_num": T})
        tap = (lambda m: m.ppm_conv) if kind == "clip_psp" else (lambda m: m.head)  # noqa: E731
    load_det(mod)
    mod.to(dev)
    tag = "infer480:" + kind
    frames = [_t(det_input("%s:%d" % (tag, t), (1, 3, H, W)), dev) for t in range(T)]
    zeros = torch.zeros(1, 1, H, W, device=dev)

    def feed(fr, lab):
        d = {"img_data": fr[-1], "seg_label": lab}
        if T > 1:
            d.update(clipimgs_data=list(fr[:-1]), cliplabels_data=[lab] * (T - 1))
        return d

    # running statistics := batch statistics of two clips (the frames and their mirror images): eval mode is then
    # meaningful with random weights (the PPM scale-1 branch needs a population of 2 in training mode)
    two = [torch.cat([f, f.flip(-1)], 0) for f in frames]
    lab2 = torch.zeros(2, 1, H, W, device=dev)
    calibrate_bn_hip(mod, lambda: mod(feed(two, lab2)))
    mod.eval()
    state = str(tmp_path / "state.npz")
    np.savez(state, **{k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()})
    jobs = [dict(mode="eval", kind=kind, arch="resnet101", T=T, shape=[H, W], tag=tag, state=state, dtype=dt,
                 out=str(tmp_path / ("ref_%s.npz" % dt)), mem_gb=30.0) for dt in ("f32", "f64")]

    got = {}
    for folded in (True, False):
        store = {}
        hk = tap(mod).register_forward_hook(lambda m, i, o: store.__setitem__("l", o.detach().float().cpu().numpy()))
        hk1 = mod.encoder.layer1.register_forward_hook(lambda m, i, o: store.__setitem__("l1", o.detach().double().cpu()))
        ops.set_inference_folding(folded)
        try:
            with torch.no_grad():
                probs = mod(feed(frames, zeros), segSize=(H, W))
        finally:
            ops.set_inference_folding(True)
            hk.remove()
            hk1.remove()
        got[folded] = (store["l"], probs.float().cpu().numpy(), store["l1"])
    ref32, ref64 = run_oracle_jobs(jobs, str(tmp_path), parallel=2)
    l32, l64 = ref32["logits"], ref64["logits"]
    logits, probs, _ = got[True]
    assert logits.shape == l64.shape == (1, K, 60, 107) and probs.shape == (1, K, H, W)
    own = float(np.abs(l32 - l64).max())
    tol = max(1e-3, 2.0 * own)
    ref = l64 if own > 0.5e-3 else l32
    err = float(np.abs(logits - ref).max())
    err_u = float(np.abs(got[False][0] - ref).max())
    unf = float(np.abs(got[False][0] - logits).max())
    shallow = float((got[True][2] - got[False][2]).norm() / got[False][2].norm())
    print("%s 480x853: oracle %.0f / %.0f s; |logit| max %.2f; |ref32 - ref64| %.2e; |hip - ref| folded %.2e unfolded "
          "%.2e (tol %.2e); |folded - unfolded| logits %.2e, layer1 rel L2 %.2e"
          % (kind, float(ref32["seconds"]), float(ref64["seconds"]), float(np.abs(l64).max()), own, err, err_u, tol,
             unf, shallow))
    assert err <= tol and err_u <= tol, (err, err_u, tol)
    assert unf <= tol, unf
    assert shallow <= 1e-5, shallow
    O.set_dtype(np.float32)
    rp = O.softmax(O.interpolate_bilinear(O.Var(l32.astype(np.float32)), (H, W)), 1).v
    assert np.abs(probs - rp).max() <= 1e-3
    assert np.abs(probs.sum(1) - 1).max() < 1e-5
    s = np.sort(rp, axis=1)
    decisive = (np.log(s[:, -1]) - np.log(s[:, -2])) > 2 * tol
    flips = probs.argmax(1) != rp.argmax(1)
    print("   arg-max: %d of %d pixels differ, all among the %d near-ties" % (flips.sum(), flips.size, (~decisive).sum()))
    assert (flips & decisive).sum() == 0
