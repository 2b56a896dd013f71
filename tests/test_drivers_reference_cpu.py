"""Driver row pinned on captures of the REFERENCE's own code (tests/golden/drivers_reference.npz,
written by tests/golden/make_golden.py:case_drivers): Evaluator, get_common, parse_devices, the four SGD groups and the
poly schedule, every argparse flag / default of train_clip2.py and test_clip2.py, config/defaults.py, the yaml files of
the in-scope configurations and the prediction palette."""
import ast
import os

import numpy as np
import pytest
import torch

from helpers import args_ns, golden

PKG_CONFIG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cvpr2021_vspw_implement_amd",
                          "config")


@pytest.fixture(scope="module")
def fx():
    return golden("drivers_reference")


def _metrics(ev):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.array([ev.Pixel_Accuracy(), ev.Pixel_Accuracy_Class(), ev.Mean_Intersection_over_Union(),
                         ev.Frequency_Weighted_Intersection_over_Union()], dtype=np.float64)


@pytest.mark.parametrize("name,ncls", [("ev124", 124), ("ev7", 7)])
def test_evaluator_equals_reference(fx, name, ncls):
    from cvpr2021_vspw_implement_amd.utils import Evaluator

    ev = Evaluator(ncls)
    for gt, pr in zip(fx[name + ":gt"], fx[name + ":pred"]):
        ev.add_batch(gt, pr)
    assert np.array_equal(ev.confusion_matrix, fx[name + ":cm"])
    np.testing.assert_allclose(_metrics(ev), fx[name + ":metrics"], rtol=1e-13, atol=0, equal_nan=True)
    ev.beforeval()
    assert np.array_equal(ev.confusion_matrix, fx[name + ":cm_beforeval"])
    np.testing.assert_allclose(_metrics(ev), fx[name + ":metrics_beforeval"], rtol=1e-13, atol=0, equal_nan=True)
    ev.reset()
    assert ev.confusion_matrix.sum() == float(fx[name + ":cm_reset_sum"]) == 0.0


def test_video_consistency_equals_reference(fx):
    from cvpr2021_vspw_implement_amd.utils import get_common

    gl, pl = list(fx["vc:gt"]), list(fx["vc:pred"])
    h, w = gl[0].shape
    for cn in (2, 4, 8):
        got = np.array(get_common(gl, pl, cn, h, w), dtype=np.float64)
        np.testing.assert_allclose(got, fx["vc:accs%d" % cn], rtol=1e-14, atol=0)


def test_parse_devices_equals_reference(fx):
    from cvpr2021_vspw_implement_amd.utils import parse_devices

    for inp, out in zip(fx["parse_devices:in"], fx["parse_devices:out"]):
        assert ",".join(parse_devices(str(inp))) == str(out), inp


@pytest.mark.parametrize("fix", [False, True])
def test_optimizer_groups_and_poly_schedule_equal_reference(fx, fix):
    """create_optimizers / adjust_learning_rate of train_clip2.py:215-252 on Clip_PSP(resnet50dilated): group sizes
    (duplicates counted), weight decays, start rates, momentum, and the learning-rate trace at seven iterations."""
    import cvpr2021_vspw_implement_amd.models as M
    from cvpr2021_vspw_implement_amd import optim

    enc = M.ModelBuilder.build_encoder(arch="resnet50dilated", fc_dim=2048)
    mod = M.Clip_PSP(enc, torch.nn.NLLLoss(ignore_index=255), args_ns(), deep_sup_scale=0.4)
    opt = optim.create_optimizers(mod, 0.002, weight_decay=1e-4, momentum=0.9, fix=fix)
    k = "opt:fix%d:" % fix
    assert [sum(g["mult"]) for g in opt.param_groups] == [int(v) for v in fx[k + "group_sizes"]]
    assert [g["weight_decay"] for g in opt.param_groups] == [float(v) for v in fx[k + "group_wd"]]
    np.testing.assert_allclose([g["lr"] for g in opt.param_groups], fx[k + "group_lr0"], rtol=1e-15)
    assert opt.param_groups[0]["momentum"] == float(fx[k + "momentum"])
    k = "lr:fix%d:" % fix
    max_iters = int(fx[k + "max_iters"])
    for it, want, running in zip(fx[k + "iters"], fx[k + "trace"], fx[k + "running_lr_encoder"]):
        r = optim.adjust_learning_rate(opt, int(it), max_iters, 0.002, lr_pow=0.9, fix=fix)
        np.testing.assert_allclose([g["lr"] for g in opt.param_groups], want, rtol=1e-15, atol=0)
        assert r == pytest.approx(float(running), rel=1e-15)


@pytest.mark.parametrize("driver", ["train_clip2.py", "test_clip2.py"])
def test_argparse_surface_equals_reference(fx, driver):
    """Every flag of the reference driver exists here with the same default, type and choices; the only additions are
    the documented ones (no reference counterpart)."""
    if driver == "train_clip2.py":
        from cvpr2021_vspw_implement_amd.train_clip2 import build_parser

        extra_ok = {"ckpt_every", "raft_weights", "hip_graph", "syncbn_formula"}
    else:
        from cvpr2021_vspw_implement_amd.test_clip2 import build_parser

        extra_ok = {"raft_weights"}
    ours = {}
    for act in build_parser()._actions:
        if act.dest == "help":
            continue
        tname = getattr(act.type, "__name__", str(act.type)) if act.type is not None else "None"
        ours[act.dest] = (repr(act.default), tname, repr(list(act.choices)) if act.choices else "None",
                          "opt" if act.option_strings else "pos")
    k = "argparse:%s:" % driver
    ref = {str(d): (str(a), str(b), str(c), str(e)) for d, a, b, c, e in
           zip(fx[k + "dest"], fx[k + "default"], fx[k + "type"], fx[k + "choices"], fx[k + "kind"])}
    assert len(ref) > 30
    for dest, spec in ref.items():
        assert dest in ours, "missing flag --%s" % dest
        assert ours[dest] == spec, (dest, ours[dest], spec)
    assert set(ours) - set(ref) <= extra_ok, set(ours) - set(ref)


def _flatten(node, prefix=""):
    out = {}
    for k, v in node.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


def test_config_defaults_equal_reference(fx):
    from cvpr2021_vspw_implement_amd.config.defaults import _DEFAULTS, CfgNode

    ours = _flatten(CfgNode(_DEFAULTS))
    ref = {str(k): ast.literal_eval(str(v)) for k, v in zip(fx["cfg:keys"], fx["cfg:values"])}
    assert ours == ref


IN_SCOPE_YAML = ["vsp-resnet101dilated-ppm_deepsup_clip.yaml", "vsp-resnet101dilated-ppm_deepsup.yaml",
                 "vsp-resnet101dilated-ocr_deepsup.yaml", "vsp-resnet101dilated-nonlocal2d.yaml",
                 "vsp-resnet101dilated-ppm_clip.yaml", "vsp-resnet50dilated-ppm_deepsup_clip.yaml",
                 "vsp-resnet50dilated-ppm_deepsup.yaml", "vsp-resnet18dilated-ppm_deepsup_clip.yaml",
                 "vsp-resnet18dilated-ppm_deepsup.yaml"]


@pytest.mark.parametrize("name", IN_SCOPE_YAML)
def test_yaml_configs_merge_to_the_reference_values(fx, name):
    """`cfg.merge_from_file(<our yaml>)` yields what the reference's defaults + its yaml of the same name yield (the
    reference's yaml files restate every key; ours list only the differences)."""
    from cvpr2021_vspw_implement_amd.config.defaults import _DEFAULTS, CfgNode

    names = [str(n) for n in fx["yaml:names"]]
    assert name in names
    ref = {str(k): ast.literal_eval(str(v)) for k, v in zip(fx["cfg:keys"], fx["cfg:values"])}
    for key, val in ast.literal_eval(str(fx["yaml:flat"][names.index(name)])):
        v = ast.literal_eval(val)
        if isinstance(v, str) and v.startswith("(") and v.endswith(")"):
            v = ast.literal_eval(v)  # yaml has no tuples: "(300, 375, ...)" is coerced by the config loader
        if isinstance(ref[key], float):
            v = float(v)
        ref[key] = v
    cfg = CfgNode(_DEFAULTS)
    cfg.merge_from_file(os.path.join(PKG_CONFIG, name))
    assert _flatten(cfg) == ref


def test_palette_equals_reference(fx):
    from cvpr2021_vspw_implement_amd.utils import vspw_palette

    assert vspw_palette() == [int(v) for v in fx["palette"]]
