"""Input pipeline on the device (csrc/data.hip through cvpr2021_vspw_implement_amd.dataset2.DeviceTransform): the
tensors handed to the model equal, bit for bit, what the reference's dataset2.py classes produce on the CPU
(tests/golden/vspw_datasets.npz) - decode on the host, everything after it in HIP kernels."""
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


def test_per_frame_batches_bit_exact(dev, tree):
    """dataset2.BaseDataset (the per-frame feed of train.py, cfg 1-2) through the device pipeline: a batch of six
    differently flipped / scaled / cropped frames and the un-augmented val frames equal the reference's tensors."""
    import cvpr2021_vspw_implement_amd.dataset2 as D

    fx = golden("vspw_dataset_frame")
    tf = D.DeviceTransform(dev)
    for ms in (False, True):
        a = args_ns(cropsize=40, dataroot=tree, trainfps=5, multi_scale=ms, lesslabel=False, train_filter=False)
        ds = D.BaseDataset(a, "train")
        samples = []
        for seed in (0, 1, 2, 3, 4, 5):
            np.random.seed(500 + seed)
            random.seed(600 + seed)
            samples.append(ds[(7 * seed + 1) % len(ds)])
        imgs, labs = tf(D.collate_raw(samples))
        assert len(imgs) == 1 and imgs[0].shape == (6, 3, 40, 40)
        for b in range(6):
            assert np.array_equal(imgs[0][b].cpu().numpy(), fx["train:ms%d:seed%d:img" % (ms, b)]), (ms, b)
            assert np.array_equal(labs[0][b].cpu().numpy(), fx["train:ms%d:seed%d:seg" % (ms, b)]), (ms, b)
    a = args_ns(cropsize=40, dataroot=tree, trainfps=5, multi_scale=True, lesslabel=False, train_filter=False)
    dv = D.BaseDataset(a, "val")
    for index in (0, len(dv) - 1):
        imgs, labs = tf([dv[index]])
        assert np.array_equal(imgs[0][0].cpu().numpy(), fx["val:%d:img" % index])
        assert np.array_equal(labs[0][0].cpu().numpy(), fx["val:%d:seg" % index])


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


@pytest.mark.plumbing
def test_pipeline_bench_smoke(dev, tmp_path):
    """tools/pipeline_bench.py end to end on a small synthetic VSPW tree: JPEG decode in 2 DataLoader workers ->
    DeviceTransform (HIP) -> TCB-PSP R101 training steps fed by the loader (reference
    dataset2.py:852-1048)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pipeline_bench.py"), "--videos", "6", "--frames", "14",
                        "--workers", "2", "--steps", "2", "--root", str(tmp_path / "tree")], capture_output=True,
                       text=True, timeout=240, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["loader"][0]["workers"] == 2 and out["loader"][0]["clips_per_s"] > 0
    assert out["transform"]["ms_per_batch_stream"] > 0 and out["transform"]["bytes_out"] == 2 * 5 * 479 * 479 * 16
    e2e = out["end_to_end"][0]
    assert e2e["workers"] == 2 and e2e["ms_per_step"] >= 0.8 * out["resident_ms_per_step"]
