"""Shared test plumbing: build the HIP-backed modules, fill them (and the numpy oracle) with the deterministic
name-keyed weights the golden fixtures were generated with, regenerate the seeded inputs."""
import os
import types

import numpy as np
import torch

from oracle.det_init import det_input, det_labels, det_tensor

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = 124


def golden(tag):
    return np.load(os.path.join(GOLDEN, tag + ".npz"))


def args_ns(**kw):
    base = dict(num_class=K, psp_weight=False, use_memory=False, memory_num=0, clipocr_all=False, clip_num=3)
    base.update(kw)
    return types.SimpleNamespace(**base)


def build(kind, arch, decoder=None, fc_dim=2048, **kw):
    import cvpr2021_vspw_implement_amd.models as M

    crit = torch.nn.NLLLoss(ignore_index=255)
    enc = M.ModelBuilder.build_encoder(arch=arch, fc_dim=fc_dim)
    if kind == "seg":
        dec = M.ModelBuilder.build_decoder(arch=decoder, fc_dim=fc_dim, num_class=K)
        return M.SegmentationModule(enc, dec, crit, kw.get("deep_sup_scale", 0.4))
    if kind == "clip_psp":
        return M.Clip_PSP(enc, crit, args_ns(**kw.get("args", {})), deep_sup_scale=0.4)
    if kind == "clip_ocr":
        return M.ClipOCRNet(enc, crit, args_ns(**kw.get("args", {})), deep_sup_scale=0.4)
    if kind == "nonlocal3d":
        return M.Non_local3d(args_ns(), enc, crit)
    if kind == "netwarp":
        dec = M.ModelBuilder.build_decoder(arch="ppm_deepsup_clip", fc_dim=fc_dim, num_class=K)
        return M.NetWarp(enc, dec, crit, args_ns(clip_num=2, flow_net=kw["flow_net"]), deep_sup_scale=0.4)
    if kind == "netwarp_ocr":
        return M.NetWarp_ocr(enc, crit, args_ns(clip_num=2, flow_net=kw["flow_net"]), deep_sup_scale=0.4)
    raise ValueError(kind)


def det_numpy_state(module, skip_prefix=(), fx=None):
    """Name-keyed deterministic weights; BatchNorm running statistics come from the fixture when it stores the
    calibrated ones ("bnstat:<key>", see tests/golden/make_golden.py:calibrate_bn)."""
    sd = {k: det_tensor(k, tuple(v.shape)) for k, v in module.state_dict().items()
          if not any(k.startswith(p) for p in skip_prefix)}
    if fx is not None:
        for f in fx.files:
            if f.startswith("bnstat:"):
                assert f[7:] in sd, f
                sd[f[7:]] = fx[f].astype(np.float32)
    return sd


def load_det(module, skip_prefix=(), fx=None):
    sd = det_numpy_state(module, skip_prefix, fx)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=not skip_prefix)
    return sd


def zero_dropout(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0


def seg_inputs(tag, train_shape=(2, 3, 65, 65), eval_shape=(1, 3, 64, 96)):
    return dict(
        eval_img=det_input(tag + ":eval", eval_shape),
        train_img=det_input(tag + ":train", train_shape),
        train_lab=det_labels(tag + ":train", (train_shape[0], 1) + train_shape[2:], K),
    )


def clip_inputs(tag, T=3, train_shape=(2, 3, 65, 65), eval_shape=(1, 3, 64, 96)):
    return dict(
        eval_imgs=[det_input("%s:eval:%d" % (tag, t), eval_shape) for t in range(T)],
        train_imgs=[det_input("%s:train:%d" % (tag, t), train_shape) for t in range(T)],
        train_labs=[det_labels("%s:train:%d" % (tag, t), (train_shape[0], 1) + train_shape[2:], K) for t in range(T)],
    )


def check_grad_norms(named_grads, fx, rtol, what, agg_rtol=None):
    """named_grads: {name: ndarray}; fixture holds grad_names / grad_norms from the reference.
    Per-parameter relative error <= rtol and (optionally) the error of the whole vector of norms <= agg_rtol."""
    names = [str(n) for n in fx["grad_names"]]
    ref = dict(zip(names, fx["grad_norms"]))
    scale = max(ref.values())
    worst = (0.0, None)
    num = den = 0.0
    for n, r in ref.items():
        assert n in named_grads, "%s: no gradient for %s" % (what, n)
        g = float(np.linalg.norm(named_grads[n].astype(np.float64)))
        err = abs(g - r) / max(r, 1e-3 * scale)
        num += (g - r) ** 2
        den += r ** 2
        if err > worst[0]:
            worst = (err, n)
    assert worst[0] <= rtol, "%s: grad-norm mismatch %.3e at %s" % (what, worst[0], worst[1])
    if agg_rtol is not None:
        agg = (num / den) ** 0.5
        assert agg <= agg_rtol, "%s: aggregate grad-norm error %.3e" % (what, agg)
    return worst


def check_argmax(pred_argmax, fx, probs_tol):
    """arg-max must be identical wherever the reference's top-2 margin exceeds 2*tol (near-ties may flip)."""
    ref = fx["eval_argmax"]
    margin = fx["eval_margin"]
    decisive = margin > 2 * probs_tol
    mism = (pred_argmax != ref) & decisive
    assert mism.sum() == 0, "%d decisive arg-max mismatches" % mism.sum()
    return int(((pred_argmax != ref) & ~decisive).sum()), int((~decisive).sum())


def logit_tol(fx, floor=1e-3, factor=2.0):
    """north_star tolerance: logits within 1e-3 of the reference's fp32 CPU path.  Where the reference's own fp32
    result is further than that from its float64 re-run (the OCR heads with random weights: 1.6e-3), the HIP result is
    held to 1.5x that measured rounding error AGAINST THE FLOAT64 LOGITS (helpers.logit_error): it may not be
    noticeably further from exact arithmetic than the reference itself is."""
    if "eval_logits64" in fx.files:
        return max(floor, factor * float(np.abs(fx["eval_logits"] - fx["eval_logits64"]).max()))
    return floor


def logit_error(fx, logits):
    """max |hip - reference|: against the float64 re-run when the fixture has one and the fp32 reference itself is
    more than the 1e-3 floor away from it, else against the reference's fp32 logits."""
    if "eval_logits64" in fx.files and float(np.abs(fx["eval_logits"] - fx["eval_logits64"]).max()) > 1e-3 / 2.0:
        return float(np.abs(logits.astype(np.float64) - fx["eval_logits64"]).max())
    return float(np.abs(logits - fx["eval_logits"]).max())


def calibrate_bn_hip(module, run_train_forward):
    """Same recipe as tests/golden/make_golden.py:calibrate_bn, on the HIP modules: one training-mode forward with momentum 1
    sets every running statistic to the batch statistic, so that eval mode is meaningful with random weights."""
    bns = [m for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    for m in bns:
        m.momentum = 1.0
    module.train()
    with torch.no_grad():
        run_train_forward()
    for m in bns:
        m.momentum = 0.1


# ------------------------------------------------------------------------------------------------------ RAFT
def raft_images(tag, shape):
    """Same two frames as tests/golden/make_golden.py:raft_images."""
    a = np.clip(det_input(tag + ":img1", shape) * 60.0 + 120.0, 0.0, 255.0).astype(np.float32)
    b = np.roll(a, (2, -3), axis=(2, 3)) + det_input(tag + ":noise", shape) * 4.0
    return a, np.clip(b, 0.0, 255.0).astype(np.float32)


def raft_state(fx):
    """Deterministic RAFT weights keyed by the reference's state_dict names (stored in the fixture)."""
    return {str(k): det_tensor(str(k), tuple(int(d) for d in str(s).split(",")) if str(s) else ())
            for k, s in zip(fx["sd_keys"], fx["sd_shapes"])}


# ------------------------------------------------------------------------------------------------------ oracle jobs
def run_oracle_jobs(jobs, workdir, parallel=None, threads=None, mem_gb=None):
    """Run tests/oracle_worker.py once per job (a dict, see the worker) in separate processes that share the host:
    at most `parallel` at a time (default: one per 16 CPUs), each with its share of the BLAS / OpenMP threads, and -
    when the jobs carry a "mem_gb" estimate - never more than 60 % of the host's available memory (`mem_gb` overrides
    the probe) committed at once.  job["after"] = index of a job that must have finished first (its decisions file).
    Returns the loaded result npz per job (same order)."""
    import json
    import subprocess
    import sys
    import time

    ncpu = os.cpu_count() or 8
    if parallel is None:
        parallel = max(1, min(len(jobs), ncpu // 16))
    if threads is None:
        threads = max(1, min(64, ncpu // parallel))
    if mem_gb is None:
        try:
            import psutil

            mem_gb = psutil.virtual_memory().available / 2 ** 30
        except Exception:
            mem_gb = 64.0
        for lim, use in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                         ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
            try:  # a container limit below what the host reports (the box dies, it does not swap)
                lv = open(lim).read().strip()
                if lv != "max" and int(lv) < 2 ** 60:
                    mem_gb = min(mem_gb, (int(lv) - int(open(use).read().strip())) / 2 ** 30)
            except Exception:
                pass
    budget = 0.6 * mem_gb
    env = dict(os.environ, OPENBLAS_NUM_THREADS=str(threads), OMP_NUM_THREADS=str(threads), PYTHONDONTWRITEBYTECODE="1",
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_worker.py")
    pending = list(range(len(jobs)))
    running, done = {}, set()
    while pending or running:
        used = sum(jobs[i].get("mem_gb", 0.0) for i in running)
        for i in list(pending):
            job = jobs[i]
            if len(running) >= parallel:
                break
            if job.get("after") is not None and job["after"] not in done:
                continue
            if running and used + job.get("mem_gb", 0.0) > budget:
                continue
            pending.remove(i)
            out = job.setdefault("out", os.path.join(workdir, "out%d.npz" % i))
            jp = os.path.join(workdir, "job_%s.json" % os.path.basename(out))
            with open(jp, "w") as f:
                json.dump(job, f)
            jenv = env
            if job.get("threads"):  # e.g. the jobs at the end of a dependency chain, which run when the pool is empty
                jenv = dict(env, OPENBLAS_NUM_THREADS=str(job["threads"]), OMP_NUM_THREADS=str(job["threads"]))
            running[i] = subprocess.Popen([sys.executable, worker, jp], env=jenv)
            used += job.get("mem_gb", 0.0)
        for i, p in list(running.items()):
            rc = p.poll()
            if rc is None:
                continue
            del running[i]
            if rc != 0:
                for q in running.values():
                    q.kill()
                raise RuntimeError("oracle worker %d failed (rc %d): %r" % (i, rc, jobs[i]))
            done.add(i)
        if running:
            time.sleep(0.2)
        elif pending and all(jobs[i].get("after") is not None and jobs[i]["after"] not in done for i in pending):
            raise RuntimeError("oracle jobs wait for jobs that never ran: %r" % pending)
    return [np.load(job["out"]) for job in jobs]


# ------------------------------------------------------------------------------------------------------ pinned decisions
def hip_decision_store(mod, taps):
    """ops.record_decisions taps of one HIP forward -> {oracle key: [arrays]} (see oracle.np_ops.set_decisions)."""
    bn_name = {id(p): n[:-len(".weight")] for n, p in mod.named_parameters() if n.endswith(".weight")}
    store = {}
    for what, key, t in taps:
        if what == "relu":
            store.setdefault(bn_name[id(key)], []).append((t > 0).cpu().numpy())
        else:  # max-pool taps [n, oh, ow, c] -> the oracle's [n, c, oh, ow]
            store.setdefault("encoder.maxpool", []).append(t.permute(0, 3, 1, 2).contiguous().cpu().numpy().astype(np.int8))
    return store


def oracle_train(fn, sd, dt, gemm="blas", decisions=None, store=None):
    """One oracle training step: fn(P) -> (loss Var, acc).  Returns (loss, {name: float64 gradient})."""
    from oracle import np_models as NM
    from oracle import np_ops as O

    O.set_dtype(dt)
    O.set_gemm(gemm)
    O.set_decisions(decisions, store)
    try:
        P = NM.Params({k: (v.astype(dt) if v.dtype.kind == "f" else v.copy()) for k, v in sd.items()}, train_params=True)
        loss, _ = fn(P, dt)
        O.tape().backward(loss)
        return float(np.asarray(loss.v).reshape(())), {k: v.astype(np.float64) for k, v in P.grads().items()}
    finally:
        O.set_decisions(None)
        O.set_gemm("blas")
        O.set_dtype(np.float32)


def pinned_gradient_errors(fn, sd, hip_store, hip_grads):
    """Per-parameter relative L2 of the HIP gradients against the float64 oracle with the HIP forward's decisions
    injected, next to the same for the float32 oracle (GEMMs in matrix-core accumulation order) with ITS decisions:
    (names, e_hip, e_oracle).  See tests/test_fullsize_gpu.py for the rationale."""
    _, g64h = oracle_train(fn, sd, np.float64, decisions="inject", store=hip_store)
    rec = {}
    _, g32 = oracle_train(fn, sd, np.float32, gemm="sequential", decisions="record", store=rec)
    _, g64o = oracle_train(fn, sd, np.float64, decisions="inject", store=rec)
    names = sorted(g64h)
    scale = max(float(np.linalg.norm(v)) for v in g64h.values())
    rel = lambda a, b: np.array([float(np.linalg.norm(a[n].astype(np.float64) - b[n]))  # noqa: E731
                                 / max(float(np.linalg.norm(b[n])), 1e-3 * scale) for n in names])
    return names, rel(hip_grads, g64h), rel(g32, g64o)
