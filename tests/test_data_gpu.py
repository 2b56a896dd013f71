"""Input pipeline on the device (csrc/data.hip through cvpr2021_vspw_implement_amd.dataset2.DeviceTransform): the
tensors handed to the model equal, bit for bit, what the reference's dataset2.py classes produce on the CPU
(tests/golden/vspw_datasets.npz) - decode on the host, everything after it in HIP kernels."""
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


def _check(fx, key, imgs, labs, b=0):
    gi = np.stack([t[b].cpu().numpy() for t in imgs])
    gl = np.stack([t[b].cpu().numpy() for t in labs])
    assert gi.shape == fx[key + ":imgs"].shape, (key, gi.shape)
    assert np.array_equal(gi, fx[key + ":imgs"]), (key, np.abs(gi - fx[key + ":imgs"]).max())
    assert np.array_equal(gl, fx[key + ":labs"]), key


def test_train_batches_bit_exact(dev, tree):
    import cvpr2021_vspw_implement_amd.dataset2 as D
    from cvpr2021_vspw_implement_amd import ops

    fx = golden("vspw_datasets")
    tf = D.DeviceTransform(dev)
    for ms in (False, True):
        a = args_ns(cropsize=40, dataroot=tree, trainfps=1, clip_num=4, dilation2="3,6,9", multi_scale=ms,
                    lesslabel=False, dilation_num=0, method="clip_psp")
        ds = D.BaseDataset_longclip(a, "train")
        samples = []
        for seed in (0, 1, 2, 3, 4):
            np.random.seed(100 + seed)
            random.seed(200 + seed)
            samples.append(ds[seed % len(ds)])
        imgs, labs = tf(D.collate_raw(samples))  # one batch of 5 clips x 4 frames
        assert len(imgs) == 4 and imgs[0].shape == (5, 3, 40, 40) and labs[0].shape == (5, 1, 40, 40)
        assert ops.is_nhwc(imgs[0])  # already in the kernels' layout
        for b, seed in enumerate((0, 1, 2, 3, 4)):
            _check(fx, "longclip:ms%d:seed%d" % (ms, seed), imgs, labs, b)
    a = args_ns(cropsize=40, dataroot=tree, trainfps=1, clip_num=2, dilation_num=0, multi_scale=True, lesslabel=False,
                method="netwarp")
    ds = D.BaseDataset_clip(a, "train")
    for seed in (0, 1, 2):
        np.random.seed(300 + seed)
        random.seed(400 + seed)
        imgs, labs = tf([ds[seed % len(ds)]])
        _check(fx, "clip:seed%d" % seed, imgs, labs)


def test_test_frames_bit_exact(dev, tree):
    import cvpr2021_vspw_implement_amd.dataset2 as D

    fx = golden("vspw_datasets")
    tf = D.DeviceTransform(dev)
    a = args_ns(clip_num=4, dilation2="3,6,9", lesslabel=False, method="clip_psp")
    ts = D.TestDataset_longclip(tree, "v_b", a, is_train=False)
    for index in (0, 7):
        imgs, labs = tf([ts[index]])
        _check(fx, "test_longclip:%d" % index, imgs, labs)
    a = args_ns(clip_num=3, dilation_num=1, lesslabel=False, method="netwarp")
    tc = D.TestDataset_clip(tree, "v_c", a, is_train=False)
    for index in (0, 9, 19):
        imgs, labs = tf([tc[index]])
        _check(fx, "test_clip:%d" % index, imgs, labs)


def test_loader_with_workers_and_no_cpu_fallback(dev, tree):
    import cvpr2021_vspw_implement_amd.dataset2 as D

    a = args_ns(cropsize=40, dataroot=tree, trainfps=1, clip_num=4, dilation2="3,6,9", multi_scale=True,
                lesslabel=False, dilation_num=0, method="clip_psp")
    ds = D.BaseDataset_longclip(a, "train")
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=True, num_workers=2, drop_last=True,
                                         collate_fn=D.collate_raw)
    tf = D.DeviceTransform(dev)
    n = 0
    for batch in loader:
        imgs, labs = tf(batch)
        assert imgs[0].shape == (3, 3, 40, 40) and torch.isfinite(imgs[0]).all()
        vals = torch.unique(labs[0])
        assert ((vals <= 123) | (vals == 255)).all()
        n += 1
    assert n == 1
    with pytest.raises(RuntimeError):
        D.DeviceTransform("cpu")
