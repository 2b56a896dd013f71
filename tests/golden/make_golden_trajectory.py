#!/usr/bin/env python
"""Multi-step TCB training trajectories pinned on the reference's OWN training loop (build container only).

For `Clip_PSP` and `ClipOCRNet` (resnet50dilated, T = 3 frames, B = 2 clips, 97 x 97 crops, Dropout2d off) the
reference's `train()` (train_clip2.py:26-124) is called as it stands - same batch assembly (`img_data` = first frame),
`adjust_learning_rate` (:237-252), `create_optimizers` (:215-236: four SGD groups from the model's
get_1x / get_10x generators, head at 10x the encoder rate, bias groups without decay, every parameter listed once per
enclosing module) - on a list of seeded batches, for STEPS = 20 optimisation steps of a 40-iteration poly schedule:
  * "f32" / "f64": the run in float32 and in float64;
  * "p0".."p5": float32 runs whose first image carries a relative perturbation of 1e-7 (one float32 ulp) - how far
    rounding-sized differences carry a float32 trajectory of THIS loop: the yardstick the GPU test gates against;
  * "n32": float32 with this container's torch 2.10 `SGD.step` left alone (see below) - recorded for information.
Stored per run: loss / accuracy of every step; at the end every parameter's norm, the momentum-buffer norms, the running
statistics of five BatchNorm layers, the parameter norm of each SGD group.  Arrays only; inputs / weights come from
seeds (oracle/det_init.py), so nothing of the reference travels.

Two adaptations, both outside the arithmetic of the model:
  * `train()` moves every batch with `.cuda(args.start_gpu)` (:45-46); there is no GPU here, so `Tensor.cuda` is the
    identity while it runs.
  * README.md:13 pins PyTorch 1.3.1, whose `SGD.step` adds the weight decay INTO the gradient tensor in place
    (`d_p = p.grad.data; d_p.add_(weight_decay, p.data)`, torch/optim/sgd.py of v1.3.1) - with the reference's duplicate
    listings the decay therefore accumulates across the k applications of one parameter.  torch 2.10 adds it out of
    place.  The runs above execute `sgd_step_1_3_1` below - a restatement of that published 1.3.1 loop - as the
    optimizer's step; "n32" keeps 2.10's own.

    python tests/golden/make_golden_trajectory.py [clip_psp] [clip_ocr]
"""
import os
import sys
import time
import types
import warnings

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import make_golden as G  # noqa: E402
from oracle.det_init import damp_residual_gammas, det_input, det_labels  # noqa: E402

K = G.K
STEPS, NUM_EPOCH, T, B, S = 20, 2, 3, 2, 97
LR = float(os.environ.get("TRAJ_LR", "0.004"))
# weights: the deterministic He-normal set of every fixture with the residual-closing BatchNorm gammas x0.25
# (det_init.damp_residual_gammas, the "damped" variant of the full-size fixtures).  With the raw set this loop is chaotic:
# a one-ulp perturbation of one image moves the loss by 5e-4 after ONE update and 2e-2 after three - no yardstick.
DAMPED = os.environ.get("TRAJ_RAW", "0") != "1"
BN_LAYERS = ("encoder.bn1", "encoder.layer1.2.bn3", "encoder.layer3.2.bn2", "encoder.layer4.2.bn3")
HEAD_BN = {"clip_psp": "ppm_conv.conv_last_.1", "clip_ocr": "conv_3x3.1"}


def sgd_step_1_3_1(self, closure=None):
    """torch.optim.SGD.step of PyTorch 1.3.1 (the version README.md:13 pins), restated: plain Python loop over
    group['params'] - duplicates included - with the weight decay added to the gradient tensor IN PLACE."""
    for group in self.param_groups:
        weight_decay, momentum = group["weight_decay"], group["momentum"]
        dampening, nesterov = group["dampening"], group["nesterov"]
        for p in group["params"]:
            if p.grad is None:
                continue
            d_p = p.grad.data
            if weight_decay != 0:
                d_p.add_(p.data, alpha=weight_decay)
            if momentum != 0:
                state = self.state[p]
                if "momentum_buffer" not in state:
                    buf = state["momentum_buffer"] = torch.clone(d_p).detach()
                else:
                    buf = state["momentum_buffer"]
                    buf.mul_(momentum).add_(d_p, alpha=1 - dampening)
                d_p = d_p.add(buf, alpha=momentum) if nesterov else buf
            p.data.add_(d_p, alpha=-group["lr"])


def batches(tag, dt, pert):
    out = []
    for it in range(STEPS):
        imgs, labs = [], []
        for t in range(T):
            x = det_input("%s:img:%d:%d" % (tag, it, t), (B, 3, S, S))
            if pert is not None and it == 0 and t == 0:
                x = x * (1.0 + 1e-7 * np.random.RandomState(900 + pert).randn(*x.shape)).astype(np.float32)
            imgs.append(torch.from_numpy(x).to(dt))
            labs.append(torch.from_numpy(det_labels("%s:lab:%d:%d" % (tag, it, t), (B, 1, S, S), K)))
        out.append((imgs, labs))
    return out


def case(M, ref_train, kind):
    tag = "tcb_train_trajectory_" + kind
    res = {}
    runs = [(torch.float32, "f32", None, True), (torch.float64, "f64", None, True)] + \
           [(torch.float32, "p%d" % i, i, True) for i in range(6)] + [(torch.float32, "n32", None, False)]
    cuda0 = torch.Tensor.cuda
    for dt, name, pert, old_sgd in runs:
        t0 = time.time()
        torch.manual_seed(0)
        args = G.args_ns(clip_num=T, method=kind, start_gpu=0, lr=LR, fix=False, dilation_num=0)
        enc = M.ModelBuilder.build_encoder(arch="resnet50dilated", fc_dim=2048)
        crit = torch.nn.NLLLoss(ignore_index=255)
        mod = (M.Clip_PSP if kind == "clip_psp" else M.ClipOCRNet)(enc, crit, args, deep_sup_scale=0.4)
        G.load_det(mod)
        if DAMPED:
            sd = mod.state_dict()
            assert damp_residual_gammas(sd)
            mod.load_state_dict(sd)
        G.zero_dropout(mod)
        mod.to(dt)
        cfg = ref_train.cfg
        cfg.TRAIN.num_epoch, cfg.TRAIN.fix_bn, cfg.TRAIN.weight_decay = NUM_EPOCH, False, 1e-4
        cfg.TRAIN.running_lr_encoder = cfg.TRAIN.lr_encoder  # as the driver's __main__ does (train_clip2.py:524-525)
        cfg.TRAIN.running_lr_decoder = cfg.TRAIN.lr_decoder
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # "duplicate parameters" - the reference's listing, kept
            opt = ref_train.create_optimizers(mod, cfg, args)
        if old_sgd:
            opt.step = types.MethodType(sgd_step_1_3_1, opt)
        history = {"train": {"epoch": [], "loss": [], "acc": []}}
        torch.Tensor.cuda = lambda self, *a, **k: self
        devnull = open(os.devnull, "w")
        stdout, sys.stdout = sys.stdout, devnull
        try:
            ref_train.train(mod, batches(tag, dt, pert), opt, history, 1, cfg, args)
        finally:
            sys.stdout = stdout
            devnull.close()
            torch.Tensor.cuda = cuda0
        res[name + ":loss"] = np.array(history["train"]["loss"], dtype=np.float64)
        res[name + ":acc"] = np.array(history["train"]["acc"], dtype=np.float64)
        named = list(mod.named_parameters())
        res["param_names"] = np.array([k for k, _ in named])
        res[name + ":param_norms"] = np.array([float(p.detach().double().norm()) for _, p in named])
        seen, mom = set(), []
        for g in opt.param_groups:
            for p in g["params"]:
                if id(p) not in seen:
                    seen.add(id(p))
                    mom.append(float(opt.state[p]["momentum_buffer"].double().norm()))
        res[name + ":momentum_norms"] = np.array(mom)
        gn = []
        for g in opt.param_groups:
            uniq = {id(p): p for p in g["params"]}
            gn.append(float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in uniq.values()))))
        res[name + ":group_norms"] = np.array(gn)
        res["group_sizes"] = np.array([len(g["params"]) for g in opt.param_groups])
        res["group_unique"] = np.array([len({id(p) for p in g["params"]}) for g in opt.param_groups])
        sd = mod.state_dict()
        for bn in BN_LAYERS + (HEAD_BN[kind],):
            res["%s:rm:%s" % (name, bn)] = sd[bn + ".running_mean"].double().numpy()
            res["%s:rv:%s" % (name, bn)] = sd[bn + ".running_var"].double().numpy()
        print("  %s %s: %.0f s, loss %.6f -> %.6f" % (tag, name, time.time() - t0, res[name + ":loss"][0],
                                                     res[name + ":loss"][-1]), flush=True)
    res["bn_layers"] = np.array(BN_LAYERS + (HEAD_BN[kind],))
    res["meta"] = np.array([STEPS, NUM_EPOCH, T, B, S])
    res["lr"] = np.float64(LR)
    np.savez_compressed(os.path.join(G.OUT, tag + ".npz"), **res)
    ens = np.stack([np.abs(res["p%d:loss" % i] - res["f64:loss"]) for i in range(6)] +
                   [np.abs(res["f32:loss"] - res["f64:loss"])]).max(0)
    print(tag, "loss f64", res["f64:loss"][[0, 4, 9, 14, 19]], "\n  float32 ensemble max |. - f64| per step", ens,
          "\n  |n32 (torch 2.10 SGD) - f64|", np.abs(res["n32:loss"] - res["f64:loss"]))


def main():
    M = G.import_reference()
    ref_train, _, _ = G.import_reference_drivers()
    want = sys.argv[1:] or ["clip_psp", "clip_ocr"]
    for kind in want:
        case(M, ref_train, kind)


if __name__ == "__main__":
    main()
