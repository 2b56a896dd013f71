"""Per-frame drivers end to end on the tiny VSPW tree (the entry points of BASELINE configs 1-2, reference train.py /
test.py): decode -> device input pipeline -> SegmentationModule step -> one fused SGD per net -> the four checkpoint
files with the reference's key names -> test.main on them (per-video metrics, palette PNGs); plus the per-frame test
dataset and the frames-as-one-batch feed through the device pipeline against the reference's tensors."""
import os
import random

import numpy as np
import pytest
import torch

from oracle.det_data import make_tiny_vspw

from helpers import args_ns, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("vspw_tiny"))
    make_tiny_vspw(root)
    return root


def test_per_frame_test_dataset_bit_exact(dev, tree):
    import cvpr2021_vspw_implement_amd.dataset2 as D

    fx = golden("vspw_dataset_frame")
    tf = D.DeviceTransform(dev)
    ts = D.TestDataset(tree, "v_b", args_ns(lesslabel=False, use_720p=False))
    for index in (0, len(ts) - 1):
        imgs, labs = tf([ts[index]])
        assert np.array_equal(imgs[0][0].cpu().numpy(), fx["test:%d:img" % index])
        assert np.array_equal(labs[0][0].cpu().numpy(), fx["test:%d:seg" % index])
    imgs, labs = tf([D.TestDataset(tree, "v_b", args_ns(lesslabel=False, use_720p=True))[3]])
    img, seg = imgs[0][0].cpu().numpy(), labs[0][0].cpu().numpy()
    assert list(img.shape) == [int(v) for v in fx["test720:3:shape"]]  # 3 x 720 x 1080: PIL's resize, bit for bit
    assert np.array_equal(img[:, ::8, ::8], fx["test720:3:img_sub"])
    assert np.array_equal(seg[:, ::8, ::8], fx["test720:3:seg_sub"])
    assert abs(float(img.astype(np.float64).sum()) - float(fx["test720:3:img_sum"])) < 1e-6 * abs(float(fx["test720:3:img_sum"]))
    assert float(seg.astype(np.float64).sum()) == float(fx["test720:3:seg_sum"])


def test_frames_as_one_batch_equals_concatenation(dev, tree):
    """train.py:41-44 feeds a per-frame model torch.cat(clip_imgs, dim=0): DeviceTransform(frames_as_batch=True) writes
    that tensor directly."""
    import cvpr2021_vspw_implement_amd.dataset2 as D
    from cvpr2021_vspw_implement_amd import ops

    a = args_ns(cropsize=40, dataroot=tree, trainfps=1, clip_num=4, dilation2="3,6,9", multi_scale=True,
                lesslabel=False, dilation_num=0)
    ds = D.BaseDataset_longclip(a, "train")
    samples = []
    for seed in (0, 1, 2):
        np.random.seed(100 + seed)
        random.seed(200 + seed)
        samples.append(ds[seed])
    tf = D.DeviceTransform(dev)
    imgs, labs = tf(samples)
    one_i, one_l = tf(samples, frames_as_batch=True)
    assert len(one_i) == 1 and one_i[0].shape == (12, 3, 40, 40) and ops.is_nhwc(one_i[0])
    assert torch.equal(one_i[0], torch.cat(imgs, dim=0)) and torch.equal(one_l[0], torch.cat(labs, dim=0))


@pytest.mark.parametrize("use_clip", [False, True])
def test_train_checkpoint_eval(dev, tree, tmp_path, use_clip):
    import cvpr2021_vspw_implement_amd.test as E
    import cvpr2021_vspw_implement_amd.train as T
    from cvpr2021_vspw_implement_amd.config import cfg as base_cfg

    save = str(tmp_path / "ck")
    here = os.path.dirname(os.path.abspath(T.__file__))
    yaml = os.path.join(here, "config", "vsp-resnet18dilated-ppm_deepsup.yaml")
    args = T.build_parser().parse_args([
        "--cfg", yaml, "--predir", "", "--dataroot", tree, "--saveroot", save, "--batchsize", "3", "--cropsize", "40",
        "--trainfps", "5", "--totalepoch", "2", "--lr", "0.01", "--multi_scale", "true", "--workers", "0", "--gpus", "0",
        "--use_clipdataset", "true" if use_clip else "false", "--clip_num", "4", "--dilation2", "3,6,9"])
    cfg = base_cfg.clone()
    T.prepare(args, cfg)
    assert cfg.MODEL.arch_encoder == "resnet18dilated" and cfg.MODEL.arch_decoder == "ppm_deepsup"
    lines = []
    hist = T.main(cfg, [0], args)
    losses = hist["train"]["loss"]
    # per-frame: 18 frames / 3 = 6 iterations per epoch; clip dataset: 3 videos / 3 = 1 (12 frames per step)
    assert len(losses) == (2 if use_clip else 12) and all(np.isfinite(losses))
    for f in ("encoder_epoch_2.pth", "decoder_epoch_2.pth", "opt_encoder_epoch_2.pth", "opt_decoder_epoch_2.pth"):
        assert os.path.exists(os.path.join(save, f)), f
    enc_sd = torch.load(os.path.join(save, "encoder_epoch_2.pth"), map_location="cpu")
    dec_sd = torch.load(os.path.join(save, "decoder_epoch_2.pth"), map_location="cpu")
    assert "conv1.weight" in enc_sd and "layer4.1.bn2.running_var" in enc_sd          # reference key names, no prefix
    keys = golden("state_keys")["seg:resnet18dilated:ppm_deepsup:keys"]   # the reference's own module, key by key
    assert list(enc_sd) == [str(k)[len("encoder."):] for k in keys if str(k).startswith("encoder.")]
    assert list(dec_sd) == [str(k)[len("decoder."):] for k in keys if str(k).startswith("decoder.")]
    opt_sd = torch.load(os.path.join(save, "opt_encoder_epoch_2.pth"), map_location="cpu")
    assert len(opt_sd["param_groups"]) == 2 and opt_sd["param_groups"][1]["weight_decay"] == 0.0
    # evaluation driver on the files train.py wrote
    eargs = E.build_parser().parse_args([
        "--cfg", yaml, "--dataroot", tree, "--split", "test", "--load_en", os.path.join(save, "encoder_epoch_2.pth"),
        "--load_de", os.path.join(save, "decoder_epoch_2.pth"), "--batchsize", "2", "--is_save", "true",
        "--saveroot", str(tmp_path / "pred")])
    ecfg = base_cfg.clone()
    E.prepare(eargs, ecfg)
    eargs.workers = 0
    eargs.dump_video_miou = False
    out = E.main(ecfg, 0, eargs, log=lambda *a: lines.append(a))
    for k in ("Acc", "mIoU", "fwIoU", "video_mIoU", "video_fwIoU"):
        assert 0.0 <= out[k] <= 1.0, (k, out[k])
    assert len(os.listdir(str(tmp_path / "pred" / "v_b"))) == 9  # one palette PNG per frame of the video


def test_hip_graph_per_frame_loop_equals_the_eager_loop(dev, tmp_path):
    """train.train with hip_graph (the step - both nets' SGDs included - captured once and replayed over static batch
    buffers) against the plain loop: the same loss trace and bit-identical parameters, buffers and momentum of both
    optimizers (Dropout2d disabled: the capture's warm-up advances the Philox offset)."""
    import cvpr2021_vspw_implement_amd.train as T
    from cvpr2021_vspw_implement_amd.config import cfg as base_cfg
    from helpers import load_det, zero_dropout

    here = os.path.dirname(os.path.abspath(T.__file__))
    yaml = os.path.join(here, "config", "vsp-resnet18dilated-ppm_deepsup.yaml")

    def run(hip_graph):
        args = T.build_parser().parse_args(["--cfg", yaml, "--predir", "", "--totalepoch", "1", "--lr", "0.01",
                                            "--gpus", "0"])
        args.hip_graph = hip_graph
        cfg = base_cfg.clone()
        T.prepare(args, cfg)
        mod, nets = T.build_module(cfg, args)
        load_det(mod)
        zero_dropout(mod)
        mod.to(dev)
        opts = T.create_optimizers(nets, cfg)
        g = torch.Generator().manual_seed(12)
        batches = [(torch.randn(3, 3, 40, 40, generator=g), torch.randint(0, args.num_class, (3, 1, 40, 40), generator=g).float())
                   for _ in range(4)]

        class Feed(object):  # stands in for loader + device transform: hands out the prepared per-frame batches
            device = dev

            def __call__(self, data, frames_as_batch=False):
                return [data[0].to(dev)], [data[1].to(dev)]

        hist = {"train": {"epoch": [], "loss": [], "acc": []}}
        T.train(mod, batches, opts, hist, 1, cfg, args, Feed(), log=lambda *a: None)
        torch.cuda.synchronize()
        state = {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}
        mom = [o.state[p]["momentum_buffer"].cpu().numpy() for o in opts for grp in o.param_groups for p in grp["params"]
               if "momentum_buffer" in o.state[p]]
        return hist["train"]["loss"], state, mom, getattr(args, "_graphed_step", None)

    l0, s0, m0, g0 = run(False)
    l1, s1, m1, g1 = run(True)
    assert g0 is None and g1 is not None
    assert len(l0) == 4 and l0 == l1, (l0, l1)
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k
    assert len(m0) == len(m1) > 0 and all(np.array_equal(a, b) for a, b in zip(m0, m1))
