"""CPU check of the decision-injection idea (oracle only): float32 oracle records its ReLU / max-pool decisions; the
float64 oracle is run free and with those decisions injected.  Per-parameter relative L2 gradient error of the float32
run against each.  usage: decision_cpu.py clip_ocr|clip_psp [arch] [S] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import K, build, det_numpy_state
from oracle import np_models as NM, np_ops as O
from oracle.det_init import det_input, det_labels
kind = sys.argv[1] if len(sys.argv) > 1 else "clip_ocr"
arch = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 65
T = int(sys.argv[4]) if len(sys.argv) > 4 else 3
B = 2
mod = build(kind, arch + "dilated", args={"clip_num": T}); sd = det_numpy_state(mod)
imgs = [det_input("benchval:%s:%d" % (kind, t), (B, 3, S, S), seed=11) for t in range(T)]
labs = [det_labels("benchval:%s:%d" % (kind, t), (B, 1, S, S), K, seed=11) for t in range(T)]
fn = NM.clip_psp if kind == "clip_psp" else NM.clip_ocr
def run(dt, mode=None, store=None):
    t0 = time.time(); O.set_dtype(dt); O.set_decisions(mode, store)
    P = NM.Params({k: v.astype(dt) for k, v in sd.items()}, train_params=True)
    ol, oa = fn(P, arch, [a.astype(dt) for a in imgs], labs, True)
    O.tape().backward(ol); O.set_decisions(None); O.set_dtype(np.float32)
    print(dt.__name__, mode, "%.1f s" % (time.time() - t0), "loss %.9f" % float(np.asarray(ol.v).reshape(())))
    return {k: v.astype(np.float64) for k, v in P.grads().items()}
store = {}
g32 = run(np.float32, "record", store)
g64 = run(np.float64)
g64i = run(np.float64, "inject", store)
def rel(a, b):
    sc = max(np.linalg.norm(v) for v in b.values())
    return {k: np.linalg.norm(a[k] - b[k]) / max(np.linalg.norm(b[k]), 1e-6 * sc) for k in b}
for name, ref in (("free fp64", g64), ("decision-injected fp64", g64i)):
    r = rel(g32, ref); v = np.array(list(r.values()))
    worst = sorted(r.items(), key=lambda kv: -kv[1])[:5]
    print("fp32 vs %s: rel L2 max %.3e median %.3e p90 %.3e" % (name, v.max(), np.median(v), np.percentile(v, 90)))
    for k, e in worst: print("     %.3e %s" % (e, k))
# conditioning of the pinned-decision (smooth) function: float64, decisions injected, inputs perturbed by 1e-7 relative
rng = np.random.default_rng(0)
imgs0 = imgs
imgs = [a * (1 + 1e-7 * rng.standard_normal(a.shape)).astype(np.float32) for a in imgs0]
g64p = run(np.float64, "inject", store)
r = rel(g64p, g64i); v = np.array(list(r.values()))
print("fp64 injected, inputs perturbed 1e-7 rel (in fp32 representation!) vs unperturbed: max %.3e median %.3e" % (v.max(), np.median(v)))
order = ["head.weight", "ppm_conv.conv_last_.4.weight", "spatial_ocr_head.conv_bn_dropout.0.weight", "ppm_conv.conv_last_.0.weight", "conv_3x3.0.weight", "encoder.layer4.2.conv3.weight", "encoder.layer4.0.conv1.weight",
         "encoder.layer3.5.conv3.weight", "encoder.layer3.0.conv1.weight", "encoder.layer2.0.conv1.weight", "encoder.layer1.0.conv1.weight", "encoder.conv1.weight"]
r = rel(g32, g64i)
for k in order:
    if k in r: print("  %-50s fp32 vs inj64 %.3e   cond %.3e" % (k, r[k], rel(g64p, g64i)[k]))
