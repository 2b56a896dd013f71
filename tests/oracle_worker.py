"""Test infrastructure: one numpy-oracle evaluation of a clip-head training step in its own process.

The full-size parity tests (tests/test_fullsize_gpu.py) need several oracle evaluations of the same step - float64
free, float64 with injected decisions, a float32 ensemble on perturbed inputs.  Each takes 30-50 s of BLAS time; run as
separate processes they share the GPU box's host cores instead of queueing behind each other (and the test process,
which has HIP initialised, is never forked).

usage: oracle_worker.py JOB.json   (fields below; writes JOB["out"] as .npz)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def pack_decisions(store, path):
    """{key: [bool / int8 arrays]} -> npz with bit-packed masks."""
    out = {}
    for k, arrs in store.items():
        for i, a in enumerate(arrs):
            a = np.asarray(a)
            tag = "%s|%d" % (k, i)
            if a.dtype == np.bool_:
                out["b|" + tag] = np.packbits(a.reshape(-1))
                out["s|" + tag] = np.asarray(a.shape, dtype=np.int64)
            else:
                out["i|" + tag] = a.astype(np.int8)
    np.savez(path, **out)


def unpack_decisions(path):
    z = np.load(path)
    tmp = {}
    for f in z.files:
        kind, key, i = f.split("|")
        if kind == "s":
            continue
        if kind == "b":
            shape = tuple(int(d) for d in z["s|%s|%s" % (key, i)])
            a = np.unpackbits(z[f], count=int(np.prod(shape))).astype(np.bool_).reshape(shape)
        else:
            a = z[f]
        tmp.setdefault(key, {})[int(i)] = a
    return {k: [v[i] for i in sorted(v)] for k, v in tmp.items()}


def perturbed(a, seed, rel):
    """inputs * (1 + rel * N(0,1)), rounded back to float32: for rel ~ 1e-7 a random subset moves by one ulp."""
    rng = np.random.default_rng(seed)
    return (a.astype(np.float64) * (1.0 + rel * rng.standard_normal(a.shape))).astype(np.float32)


def eval_logits(job):
    """mode "eval": inference logits (before the final up-sampling) of one 480p prediction as test_clip2.py runs it
    (reference test_clip2.py:28-89): state dict from job["state"] (npz, calibrated running statistics included),
    frames = det_input(job["tag"] + ":<t>"), the current frame LAST."""
    from oracle import np_models as NM
    from oracle import np_ops as O
    from oracle.det_init import det_input

    kind, arch, T = job["kind"], job["arch"], job["T"]
    h, w = job["shape"]
    dt = {"f32": np.float32, "f64": np.float64}[job["dtype"]]
    z = np.load(job["state"])
    t0 = time.time()
    O.set_dtype(dt)
    O.set_gemm(job.get("gemm", "blas"))
    P = NM.Params({k: z[k].astype(dt) if z[k].dtype.kind == "f" else z[k] for k in z.files}, train_params=False)
    frames = [det_input("%s:%d" % (job["tag"], t), (1, 3, h, w)).astype(dt) for t in range(T)]
    if kind == "seg_ppm":  # per-frame PSPNet (reference models/models.py:938-995), logits of conv_last_
        feats = NM.resnet_dilated(P, O.Var(frames[-1]), arch, "encoder.", False)
        pooled = [O.adaptive_avg_pool2d(feats[-1], s) for s in (1, 2, 3, 6)]
        x = NM._head(P, NM._ppm_concat(P, feats[-1], pooled, "decoder.ppm.", 1, 2, False), "decoder.conv_last_", False)
    elif kind == "clip_psp":
        _, x = NM.clip_psp(P, arch, frames, None, False, seg_size=(8, 8))
    else:
        _, x = NM.clip_ocr(P, arch, frames, None, False, seg_size=(8, 8))
    O.set_gemm("blas")
    O.set_dtype(np.float32)
    np.savez(job["out"], logits=x.v.astype(np.float64), seconds=np.float64(time.time() - t0))


def main(job):
    if job.get("mode") == "eval":
        return eval_logits(job)
    from helpers import K, build, det_numpy_state
    from oracle import np_models as NM
    from oracle import np_ops as O
    from oracle.det_init import det_input, det_labels

    kind, arch, T, B, S = job["kind"], job["arch"], job["T"], job["B"], job["S"]
    dt = {"f32": np.float32, "f64": np.float64}[job["dtype"]]
    mod = build(kind, arch + "dilated", args={"clip_num": T})
    sd = det_numpy_state(mod)
    del mod
    tag = job.get("tag", "benchval")
    imgs = [det_input("%s:%s:%d" % (tag, kind, t), (B, 3, S, S), seed=11) for t in range(T)]
    labs = [det_labels("%s:%s:%d" % (tag, kind, t), (B, 1, S, S), K, seed=11) for t in range(T)]
    if job.get("perturb_seed") is not None:
        imgs = [perturbed(a, job["perturb_seed"] * 100 + t, job.get("perturb_rel", 1e-7)) for t, a in enumerate(imgs)]
    fn = NM.clip_psp if kind == "clip_psp" else NM.clip_ocr
    store = None
    if job.get("decisions") == "inject":
        store = unpack_decisions(job["decisions_path"])
    elif job.get("decisions") == "record":
        store = {}
    t0 = time.time()
    O.set_dtype(dt)
    O.set_gemm(job.get("gemm", "blas"))  # "sequential": float32 GEMMs as one fmaf chain per element (np_ops.set_gemm)
    O.set_decisions(job.get("decisions"), store)
    acts = {}
    if job.get("dump_acts"):  # diagnostics: keep the ReLU outputs of the named nodes (tools/diag/pinned.py)
        want, relu0 = set(job["dump_acts"]), O.relu

        def relu(x, key=None):
            o = relu0(x, key)
            if key in want:
                acts["a:" + key] = o.v.astype(np.float32)
            return o

        O.relu = relu
    P = NM.Params({k: v.astype(dt) for k, v in sd.items()}, train_params=True)
    loss, acc = fn(P, arch, [a.astype(dt) for a in imgs], labs, True)
    O.tape().backward(loss)
    O.set_decisions(None)
    O.set_gemm("blas")
    O.set_dtype(np.float32)
    grads = P.grads()
    names = sorted(grads)
    out = {"loss": np.float64(np.asarray(loss.v).reshape(())), "acc": np.float64(acc), "names": np.array(names),
           "norms": np.array([np.linalg.norm(grads[k].astype(np.float64)) for k in names]),
           "seconds": np.float64(time.time() - t0)}
    out.update(acts)
    if job.get("full_grads", False):
        for k in names:
            out["g:" + k] = grads[k].astype(np.float32)
    np.savez(job["out"], **out)
    if job.get("decisions") == "record":
        pack_decisions(store, job["decisions_path"])


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        main(json.load(f))
