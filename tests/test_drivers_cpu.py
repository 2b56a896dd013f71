"""Host logic of the driver mirrors (train_clip2.py / test_clip2.py / config): flags, schedule fields, feed_dict
assembly, checkpoint key handling.  CPU only, no compute."""
import os

import pytest
import torch


def _cfg():
    from cvpr2021_vspw_implement_amd.config import cfg

    return cfg.clone()


def test_train_flags_and_schedule_fields():
    import cvpr2021_vspw_implement_amd.train_clip2 as T

    here = os.path.dirname(os.path.abspath(T.__file__))
    args = T.build_parser().parse_args(["--cfg", os.path.join(here, "config", "vsp-resnet101dilated-ppm_deepsup_clip.yaml"),
                                        "--method", "clip_psp", "--lr", "0.002", "--totalepoch", "120", "--clip_num", "4",
                                        "--dilation2", "3,6,9", "--gpus", "0-7", "TRAIN.epoch_iters", "10"])
    # defaults of train_clip2.py:399-478
    assert (args.num_class, args.batchsize, args.cropsize, args.dilation_num, args.weight_decay) == (124, 16, 531, 3, 1e-4)
    assert args.validation is True and args.multi_scale is False and args.psp_weight is False
    cfg = _cfg()
    gpus = T.prepare(args, cfg)
    assert gpus == list(range(8))
    assert cfg.MODEL.arch_encoder == "resnet101dilated" and cfg.MODEL.arch_decoder == "ppm_deepsup_clip"
    assert cfg.TRAIN.num_epoch == 120 and cfg.TRAIN.epoch_iters == 10 and cfg.TRAIN.max_iters == 1200
    assert cfg.TRAIN.running_lr_encoder == 0.002 and cfg.TRAIN.weight_decay == 1e-4
    assert args.max_distances == [10]
    with pytest.raises(KeyError):
        cfg.merge_from_list(["TRAIN.no_such_key", "1"])
    with pytest.raises(SystemExit):
        T.build_parser().parse_args(["--method", "bogus"])


def test_feed_dict_assembly():
    import cvpr2021_vspw_implement_amd.train_clip2 as T

    frames = [torch.full((2, 3, 4, 4), float(t)) for t in range(4)]
    labels = [torch.full((2, 1, 4, 4), float(t)) for t in range(4)]
    a = T.build_parser().parse_args(["--method", "clip_psp", "--clip_num", "4"])
    b = T.make_batch(a, frames, labels, 7)
    assert b["img_data"] is frames[0] and len(b["clipimgs_data"]) == 3 and b["step"] == 7
    a = T.build_parser().parse_args(["--method", "netwarp", "--clip_num", "2", "--dilation_num", "0"])
    b = T.make_batch(a, frames[:2], labels[:2], 1)
    assert float(b["img_data"][0, 0, 0, 0]) == 1.0 and float(b["clipimgs_data"][0][0, 0, 0, 0]) == 0.0
    a = T.build_parser().parse_args(["--method", "nonlocal3d", "--clip_num", "4"])
    b = T.make_batch(a, frames, labels, 1)
    assert "img_data" not in b and len(b["clipimgs_data"]) == 4
    a = T.build_parser().parse_args(["--method", "tdnet"])
    with pytest.raises(NotImplementedError):
        T.build_module(_cfg(), a, 124)


def test_checkpoint_keys_and_poly_schedule(tmp_path):
    import cvpr2021_vspw_implement_amd.train_clip2 as T

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = torch.nn.Conv2d(3, 4, 1)
            self.head = torch.nn.Conv2d(4, 2, 1)

    net = Net()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    args = T.build_parser().parse_args(["--method", "clip_psp", "--saveroot", str(tmp_path / "ck")])
    T.checkpoint(opt, net, {}, args, 20)
    sd = torch.load(str(tmp_path / "ck" / "model_epoch_20.pth"))
    assert all(k.startswith("module.") for k in sd) and os.path.exists(str(tmp_path / "ck" / "opt_epoch_20.pth"))
    net.load_state_dict(T.strip_module_prefix(sd))          # what both loaders do
    net.load_state_dict(T.strip_module_prefix(net.state_dict()))  # prefix-free checkpoints load too
    # poly schedule of train_clip2.py:239-252 on the four groups
    groups = [{"params": [p]} for p in net.parameters()]
    opt = torch.optim.SGD(groups, lr=0.02)
    cfg = _cfg()
    T.adjust_learning_rate(opt, 25, cfg, 100, args)
    run = 0.02 * (1 - 25 / 100) ** 0.9
    assert [g["lr"] for g in opt.param_groups] == pytest.approx([run * 0.1, run, run * 0.1, run])
    assert cfg.TRAIN.running_lr_encoder == pytest.approx(run)


def test_eval_flags_and_palette():
    import cvpr2021_vspw_implement_amd.test_clip2 as E

    a = E.build_parser().parse_args(["--method", "clip_ocr", "--use_memory", "true"])
    assert (a.batchsize, a.split, a.vc_clip_num, a.memory_num, a.clip_num) == (4, "val", 8, 8, 5) and a.use_memory
    assert E._palette[:9] == [0, 0, 0, 128, 0, 0, 0, 128, 0] and E._palette[27:30] == [191, 0, 0]
    assert E._palette[22 * 3:22 * 3 + 3] == [22, 22, 22] and len(E._palette) == 768


def test_optimizer_checkpoint_layout_is_torchs_with_duplicates_expanded():
    """opt_epoch_N.pth interchange (reference train_clip2.py:179-189,347-357): optim.SGD.state_dict() equals what
    torch.optim.SGD - built, like the reference's, on parameter lists that repeat a parameter - writes: same index
    lists, same state keys; and a torch-written state dict loads back into the de-duplicated optimizer."""
    import warnings

    import torch

    from cvpr2021_vspw_implement_amd import optim

    ps = [torch.nn.Parameter(torch.randn(3, 2, 1, 1)), torch.nn.Parameter(torch.randn(4)),
          torch.nn.Parameter(torch.randn(2, 2))]
    lists = [[ps[0], ps[1], ps[0]], [ps[2], ps[2], ps[2]]]
    mk = lambda: [{"params": list(lists[0]), "lr": 0.1, "weight_decay": 1e-4},  # noqa: E731
                  {"params": list(lists[1]), "lr": 0.01, "weight_decay": 0.0}]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = torch.optim.SGD(mk(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    for p in ps:
        ref.state[p]["momentum_buffer"] = torch.randn_like(p)
    want = ref.state_dict()
    ours = optim.SGD(mk(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    assert [g["mult"] for g in ours.param_groups] == [[2, 1], [3]]
    ours.load_state_dict(want)
    got = ours.state_dict()
    assert [g["params"] for g in got["param_groups"]] == [g["params"] for g in want["param_groups"]] == [[2, 1, 2], [5, 5, 5]]
    assert set(got["state"]) == set(want["state"]) == {1, 2, 5}
    for k in want["state"]:
        assert torch.equal(got["state"][k]["momentum_buffer"], want["state"][k]["momentum_buffer"])
    for g, w in zip(got["param_groups"], want["param_groups"]):
        assert g["lr"] == w["lr"] and g["weight_decay"] == w["weight_decay"] and g["momentum"] == w["momentum"]
        assert "mult" not in g and "order" not in g
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        back = torch.optim.SGD(mk(), lr=0.5, momentum=0.9)
    back.load_state_dict(got)  # and torch reads what we write
    assert back.param_groups[1]["lr"] == 0.01
    assert torch.equal(back.state[ps[2]]["momentum_buffer"], ref.state[ps[2]]["momentum_buffer"])


def test_optimizer_resumes_from_its_round2_layout_and_can_checkpoint_again():
    """An `opt_epoch_N.pth` written by this class in round 2 (de-duplicated lists, 'mult', no 'order') loads, and the next
    state_dict() - what checkpoint() calls at the end of the first resumed epoch - works and speaks torch's layout."""
    import torch

    from cvpr2021_vspw_implement_amd import optim

    ps = [torch.nn.Parameter(torch.randn(3, 2, 1, 1)), torch.nn.Parameter(torch.randn(4)),
          torch.nn.Parameter(torch.randn(2, 2))]
    mk = lambda: [{"params": [ps[0], ps[1], ps[0]], "lr": 0.1, "weight_decay": 1e-4},  # noqa: E731
                  {"params": [ps[2], ps[2], ps[2]], "lr": 0.01, "weight_decay": 0.0}]
    bufs = [torch.randn_like(p) for p in ps]
    round2 = {"state": {0: {"momentum_buffer": bufs[0]}, 1: {"momentum_buffer": bufs[1]}, 2: {"momentum_buffer": bufs[2]}},
              "param_groups": [{"lr": 0.05, "momentum": 0.9, "weight_decay": 1e-4, "mult": [2, 1], "params": [0, 1]},
                               {"lr": 0.005, "momentum": 0.9, "weight_decay": 0.0, "mult": [3], "params": [2]}]}
    ours = optim.SGD(mk(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    ours.load_state_dict(round2)
    assert [g["mult"] for g in ours.param_groups] == [[2, 1], [3]]
    assert [g["order"] for g in ours.param_groups] == [[0, 1, 0], [0, 0, 0]]
    assert ours.param_groups[0]["lr"] == 0.05 and ours.param_groups[1]["lr"] == 0.005
    got = ours.state_dict()
    assert [g["params"] for g in got["param_groups"]] == [[2, 1, 2], [5, 5, 5]]
    assert set(got["state"]) == {1, 2, 5}
    assert torch.equal(got["state"][2]["momentum_buffer"], bufs[0]) and torch.equal(got["state"][5]["momentum_buffer"], bufs[2])
    bad = {"state": {}, "param_groups": [dict(round2["param_groups"][0], mult=[1, 1]), round2["param_groups"][1]]}
    import pytest

    with pytest.raises(ValueError):
        optim.SGD(mk(), lr=0.1, momentum=0.9).load_state_dict(bad)
