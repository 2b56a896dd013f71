"""Numerics of Winograd F(m x m, 3 x 3) tile sizes, CPU only (oracle arithmetic: float32 transforms, sequential fmaf-chain
GEMMs = the matrix cores' k loop).  Answers review item 1 of round 6: does F(3x3, 3x3) pass the parity gate that
F(4x4, 3x3) failed?

 part 1 (conv level): one layer-3 shaped convolution (256 -> 256, dilation 2), all three passes, relative L2 error
                      against the float64 direct convolution.
 part 2 (network level, the "pinned-decision" probe of DESIGN section 3/4): TCB-PSP R50, T=3, S^2; the float32 oracle
                      with every eligible 3x3 routed through the candidate records its ReLU / max-pool decisions; the
                      float64 oracle re-runs with those decisions injected; per-parameter relative L2 of the gradients.
 usage: wino_f33_probe.py [conv|net|all] [S=129] [kind=clip_psp]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import np_ops as O, np_wino as W

what = sys.argv[1] if len(sys.argv) > 1 else "all"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 129
kind = sys.argv[3] if len(sys.argv) > 3 else "clip_psp"
F = lambda a, b: __import__("fractions").Fraction(a, b)
CANDIDATES = [
    ("direct", None),
    ("F(2x2) {0,1,-1}", (2, [0, 1, -1])),
    ("F(3x3) {0,1,-1,2}", (3, [0, 1, -1, 2])),
    ("F(3x3) {0,1,-1,1/2}", (3, [0, 1, -1, F(1, 2)])),
    ("F(3x3) {0,1,-1,-1/2}", (3, [0, 1, -1, F(-1, 2)])),
    ("F(3x3) {0,1/2,-1/2,1}", (3, [0, F(1, 2), F(-1, 2), 1])),
    ("F(3x3) {0,1/2,-1/2,2}", (3, [0, F(1, 2), F(-1, 2), 2])),
    ("F(4x4) {0,1,-1,2,-2}", (4, [0, 1, -1, 2, -2])),
    ("F(4x4) {0,1,-1,1/2,-2}", (4, [0, 1, -1, F(1, 2), -2])),
    ("F(5x5) {0,1,-1,1/2,-1/2,2}", (5, [0, 1, -1, F(1, 2), F(-1, 2), 2])),
]
if os.environ.get("CANDS"):
    sel = [int(i) for i in os.environ["CANDS"].split(",")]
    CANDIDATES = [CANDIDATES[i] for i in sel]
rl = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
O.set_gemm(os.environ.get("ORACLE_GEMM", "sequential"))


def conv_level():
    rng = np.random.default_rng(5)
    nb, c, k, h, dil = 2, 256, 256, 60, int(os.environ.get("DIL", "2"))
    x = np.maximum(rng.standard_normal((nb, c, h, h)), 0).astype(np.float32)  # post-ReLU activations
    w = (rng.standard_normal((k, c, 3, 3)) * np.sqrt(2.0 / (9 * c))).astype(np.float32)
    g = rng.standard_normal((nb, k, h, h)).astype(np.float32)
    O.set_dtype(np.float64)
    xv, wv = O.Var(x.astype(np.float64), True), O.Var(w.astype(np.float64), True)
    y = O.conv2d(xv, wv, None, 1, dil, dil); y64 = y.v.copy()
    O.tape().backward(y, 1.0) if False else None
    y.g = g.astype(np.float64)
    for fn in reversed(O.tape().steps): fn()
    O.tape().steps = []
    dx64, dw64 = xv.g.copy(), wv.g.copy()
    O.set_dtype(np.float32)
    print("conv level: %d x %d -> %d, %dx%d, dilation %d; relative L2 against float64 (fwd | dgrad | wgrad)" % (nb, c, k, h, h, dil))
    for name, spec in CANDIDATES:
        t0 = time.time()
        for wt64, x64 in (((True, False),) if spec is None else ((True, True), (True, False), (False, False))):
            W.install(None if spec is None else W.Winograd(spec[0], spec[1], weight_f64=wt64, xform_f64=x64), min_c=128)
            xv, wv = O.Var(x, True), O.Var(w, True)
            y = O.conv2d(xv, wv, None, 1, dil, dil)
            y.g = g
            for fn in reversed(O.tape().steps): fn()
            O.tape().steps = []
            print("  %-26s %-14s %.3e | %.3e | %.3e   (%.0f s)" % (name, "" if spec is None else ("U f%d, x f%d" % (64 if wt64 else 32, 64 if x64 else 32)),
                                                                  rl(y.v, y64), rl(xv.g, dx64), rl(wv.g, dw64), time.time() - t0), flush=True)
    W.install(None)


def net_level():
    from helpers import K, build, det_numpy_state
    from oracle import np_models as NM
    from oracle.det_init import det_input, det_labels
    arch, T, B = "resnet50", 3, 2
    mod = build(kind, arch + "dilated", args={"clip_num": T}); sd = det_numpy_state(mod)
    imgs = [det_input("benchval:%s:%d" % (kind, t), (B, 3, S, S), seed=11) for t in range(T)]
    labs = [det_labels("benchval:%s:%d" % (kind, t), (B, 1, S, S), K, seed=11) for t in range(T)]
    fn = NM.clip_psp if kind == "clip_psp" else NM.clip_ocr

    def run(dt, mode=None, store=None):
        O.set_dtype(dt); O.set_decisions(mode, store)
        P = NM.Params({k: v.astype(dt) for k, v in sd.items()}, train_params=True)
        ol, oa = fn(P, arch, [a.astype(dt) for a in imgs], labs, True)
        O.tape().backward(ol); O.set_decisions(None); O.set_dtype(np.float32)
        return float(np.asarray(ol.v).reshape(())), {k: v.astype(np.float64) for k, v in P.grads().items()}

    def rel(a, b):
        sc = max(np.linalg.norm(v) for v in b.values())
        return np.array([np.linalg.norm(a[k] - b[k]) / max(np.linalg.norm(b[k]), 1e-3 * sc) for k in b])

    print("network level: %s %s T=%d B=%d %d^2; float32 candidate vs float64 with the candidate's decisions" % (kind, arch, T, B, S))
    for name, spec in CANDIDATES:
        t0 = time.time()
        W.install(None if spec is None else W.Winograd(spec[0], spec[1], weight_f64=True, xform_f64=os.environ.get("XFORM64", "0") == "1"), min_c=128)
        store = {}
        l32, g32 = run(np.float32, "record", store)
        n = W._active["count"]
        W.install(None)
        l64, g64 = run(np.float64, "inject", store)
        v = rel(g32, g64)
        print("  %-26s convs %3d  loss32 %.8f loss64 %.8f |d| %.2e   grad rel L2: median %.3e p90 %.3e p99 %.3e max %.3e   (%.0f s)" % (
            name, n, l32, l64, abs(l32 - l64), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max(), time.time() - t0), flush=True)


def fwd_level():
    """Raw-weight forward excess: relative L2 (and max abs) error of deep activations of the float32 candidate against
    the FREE float64 run, next to the direct float32 convolution's own - R101, training-mode BatchNorm, no injection."""
    from helpers import K, build, det_numpy_state
    from oracle import np_models as NM
    from oracle.det_init import det_input, det_labels
    arch, T, B = os.environ.get("ARCH", "resnet101"), 2, 2
    mod = build(kind, arch + "dilated", args={"clip_num": T}); sd = det_numpy_state(mod)
    imgs = [det_input("benchval:%s:%d" % (kind, t), (B, 3, S, S), seed=11) for t in range(T)]
    labs = [det_labels("benchval:%s:%d" % (kind, t), (B, 1, S, S), K, seed=11) for t in range(T)]
    fn = NM.clip_psp if kind == "clip_psp" else NM.clip_ocr
    nl3 = {"resnet101": 23, "resnet50": 6}[arch]
    keys = ["encoder.layer2.3.bn3", "encoder.layer3.%d.bn3" % (nl3 // 2), "encoder.layer3.%d.bn3" % (nl3 - 1), "encoder.layer4.2.bn3"] + \
           (["ppm_conv.conv_last_.1", "deepsup.1"] if kind == "clip_psp" else ["conv_3x3.1", "spatial_ocr_head.conv_bn_dropout.1"])
    relu0 = O.relu

    def run(dt):
        acts = {}
        def relu(x, key=None):
            o = relu0(x, key)
            if key in keys: acts.setdefault(key, o.v.astype(np.float64))
            return o
        O.relu = relu
        try:
            O.set_dtype(dt)
            P = NM.Params({k: v.astype(dt) for k, v in sd.items()}, train_params=False)
            ol, oa = fn(P, arch, [a.astype(dt) for a in imgs], labs, True)
            O.tape().steps = []
        finally:
            O.relu = relu0; O.set_dtype(np.float32)
        return float(np.asarray(ol.v).reshape(())), acts

    t0 = time.time(); l64, a64 = run(np.float64)
    print("forward level: %s %s T=%d B=%d %d^2 raw weights, free decisions (float64 run %.0f s); relative L2 | max abs per node" % (kind, arch, T, B, S, time.time() - t0))
    print("  %-26s %-12s " % ("", "|loss-l64|") + " ".join("%-21s" % k.replace("encoder.", "")[-21:] for k in keys))
    for name, spec in CANDIDATES:
        t0 = time.time()
        W.install(None if spec is None else W.Winograd(spec[0], spec[1], weight_f64=True), min_c=128)
        l32, a32 = run(np.float32)
        W.install(None)
        print("  %-26s %.3e    " % (name, abs(l32 - l64)) + " ".join("%.2e | %.2e " % (rl(a32[k], a64[k]), np.abs(a32[k] - a64[k]).max()) for k in keys) + "  (%.0f s)" % (time.time() - t0), flush=True)


if what in ("conv", "all"): conv_level()
if what in ("fwd", "all"): fwd_level()
if what in ("net", "all"): net_level()
