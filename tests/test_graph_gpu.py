"""hipGraph-captured training step == the eagerly launched one; inference weight folding follows parameter updates."""

import numpy as np
import pytest
import torch

from helpers import build, clip_inputs, golden, load_det, zero_dropout

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _make(dev, tag="r50_clip_psp"):
    fx = golden(tag)
    mod = build("clip_psp", "resnet50dilated")
    load_det(mod, fx=fx)
    zero_dropout(mod)
    return mod.to(dev).train(), clip_inputs(tag)


def test_graphed_step_equals_eager_steps(dev):
    """warm-up (2 eager steps on a side stream) + capture + 3 replays with the poly schedule advancing == 5 eager
    steps: same losses, same parameters, same BatchNorm running statistics (the kernels are deterministic)."""
    from cvpr2021_vspw_implement_amd import optim
    from cvpr2021_vspw_implement_amd.graph import GraphedStep

    results = []
    for use_graph in (False, True):
        mod, inp = _make(dev)
        imgs = [_t(a, dev) for a in inp["train_imgs"]]
        labs = [_t(a, dev) for a in inp["train_labs"]]
        opt = optim.create_optimizers(mod, lr=0.01, weight_decay=1e-4, momentum=0.9)
        losses = []

        def body():
            mod.zero_grad()
            loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": list(imgs[:-1]),
                             "cliplabels_data": list(labs[:-1])})
            loss.backward()
            opt.step()
            return loss

        if use_graph:
            optim.adjust_learning_rate(opt, 0, 10, 0.01)
            g = GraphedStep(body, warmup=2)  # two eager steps at the schedule's first value
            for it in (2, 3, 4):
                optim.adjust_learning_rate(opt, it, 10, 0.01)
                opt.set_lrs()
                losses.append(float(g.replay().item()))
        else:
            for it in (0, 0, 2, 3, 4):
                optim.adjust_learning_rate(opt, it, 10, 0.01)
                loss = body()
                if it >= 2:
                    losses.append(float(loss.item()))
        torch.cuda.synchronize()
        results.append((losses, {k: v.detach().float().cpu().numpy().copy() for k, v in mod.state_dict().items()}))
    (l0, s0), (l1, s1) = results
    assert np.allclose(l0, l1, rtol=1e-6, atol=0), (l0, l1)
    assert l0[0] != l0[-1], "the steps must actually train"
    for k in s0:
        assert np.allclose(s0[k], s1[k], rtol=1e-5, atol=1e-7), k


def test_eval_after_training_step_uses_current_weights(dev):
    """eval -> SGD step -> eval: the folded conv+BN weights cached for inference must be rebuilt after the fused SGD
    kernel / the training-mode BatchNorm finalize rewrote parameters and running statistics through raw pointers."""
    from cvpr2021_vspw_implement_amd import ops, optim

    mod, inp = _make(dev)
    ev = [_t(a, dev) for a in inp["eval_imgs"]]
    imgs = [_t(a, dev) for a in inp["train_imgs"]]
    labs = [_t(a, dev) for a in inp["train_labs"]]
    opt = optim.create_optimizers(mod, lr=0.05, weight_decay=1e-4, momentum=0.9)

    def predict():
        mod.eval()
        with torch.no_grad():
            return mod({"img_data": ev[-1], "clipimgs_data": list(ev[:-1]),
                        "seg_label": torch.zeros(1, 1, 64, 96, device=dev)}, segSize=(64, 96)).float().cpu().numpy()

    p0 = predict()
    mod.train()
    for _ in range(2):
        mod.zero_grad()
        loss, _ = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": list(imgs[:-1]),
                       "cliplabels_data": list(labs[:-1])})
        loss.backward()
        opt.step()
    p1 = predict()
    ops.set_inference_folding(False)
    try:
        p1_unfolded = predict()
    finally:
        ops.set_inference_folding(True)
    assert np.abs(p1 - p0).max() > 1e-4, "two SGD steps at lr 0.05 must change the prediction"
    assert np.abs(p1 - p1_unfolded).max() < 2e-4, "folded inference path is stale after the parameter update"


def test_eval_between_graph_replays_uses_current_weights(dev):
    """eval -> replays -> eval (what `train_clip2 --hip_graph` does at every checkpoint epoch): a hipGraph replay rewrites
    parameters and running statistics through raw pointers without running any Python, so the replay itself must
    invalidate the folded conv+BN weights cached by the first validation (the second validation of a
    --hip_graph run reported the metrics of stale weights)."""
    from cvpr2021_vspw_implement_amd import ops, optim
    from cvpr2021_vspw_implement_amd.graph import GraphedStep

    mod, inp = _make(dev)
    ev = [_t(a, dev) for a in inp["eval_imgs"]]
    imgs = [_t(a, dev) for a in inp["train_imgs"]]
    labs = [_t(a, dev) for a in inp["train_labs"]]
    opt = optim.create_optimizers(mod, lr=0.01, weight_decay=1e-4, momentum=0.9)

    def predict():
        mod.eval()
        with torch.no_grad():
            p = mod({"img_data": ev[-1], "clipimgs_data": list(ev[:-1]),
                     "seg_label": torch.zeros(1, 1, 64, 96, device=dev)}, segSize=(64, 96)).float().cpu().numpy()
        mod.train()
        return p

    def body():
        mod.zero_grad()
        loss, acc = mod({"img_data": imgs[-1], "seg_label": labs[-1], "clipimgs_data": list(imgs[:-1]),
                         "cliplabels_data": list(labs[:-1])})
        loss.backward()
        opt.step()
        return loss

    g = GraphedStep(body, warmup=2)
    p0 = predict()          # fills the folded-weight cache
    for _ in range(3):
        g.replay()          # pure replays: no Python-side generation bump except the one under test
    p1 = predict()
    ops.set_inference_folding(False)
    try:
        p1_unfolded = predict()
    finally:
        ops.set_inference_folding(True)
    change, gap = float(np.abs(p1 - p0).max()), float(np.abs(p1 - p1_unfolded).max())
    print("prediction moved by %.3e over the replays; folded vs unfolded after them %.3e" % (change, gap))
    assert change > 1e-4, "the replayed SGD steps must change the prediction"
    # a stale cache reproduces the OLD weights in every conv+BN layer: the gap would be of the order of `change`
    assert gap < 0.05 * change and gap < 1e-3, "folded inference path is stale after the graph replays"
