"""Evaluation-side scripts of the reference on the GPU: the nearest flow-warp of TC_cal.py against the reference's own
function (bit-exact label maps), TC_cal.main end to end (HIP RAFT + HIP warp) against the same pipeline with ATen's
grid_sample on the CPU, change2_480p against PIL."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

from helpers import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_nearest_flowwarp_equals_reference(dev, name):
    """vspw_flowwarp_nearest vs TC_cal.flowwarp (grid_sample mode='nearest'): sub-pixel flows, exact halves (ties go to
    the even pixel), sources outside the image (zeros)."""
    from cvpr2021_vspw_implement_amd import TC_cal

    fx = golden("metric_tools_reference")
    lab, flo = torch.from_numpy(fx["warp:%s:lab" % name]), torch.from_numpy(fx["warp:%s:flow" % name])
    out = TC_cal.flowwarp(lab.to(dev), flo.to(dev)).cpu().numpy()
    want = fx["warp:%s:out" % name]
    assert out.shape == want.shape and np.array_equal(out, want)
    assert (want == 0).any() and (want != 0).any()


def _cpu_warp(x, flo):
    B, C, H, W = x.shape
    xx = torch.arange(0, W).view(1, -1).repeat(H, 1).view(1, 1, H, W).repeat(B, 1, 1, 1)
    yy = torch.arange(0, H).view(-1, 1).repeat(1, W).view(1, 1, H, W).repeat(B, 1, 1, 1)
    vgrid = torch.cat((xx, yy), 1).float() + flo
    gx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0
    gy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack([gx, gy], dim=-1), mode="nearest", align_corners=False)


def test_tc_cal_end_to_end(dev, tmp_path):
    """TC_cal.main on the tiny tree with synthetic prediction PNGs: same score as recomputing it from the HIP flows with
    ATen's nearest grid_sample on the CPU and the reference-pinned Evaluator."""
    from cvpr2021_vspw_implement_amd import TC_cal
    from cvpr2021_vspw_implement_amd.utils import Evaluator

    root, pred = str(tmp_path / "tree"), str(tmp_path / "pred")
    rs = np.random.RandomState(3)
    videos = ["va", "vb"]
    os.makedirs(root)
    with open(os.path.join(root, "val.txt"), "w") as f:
        f.write("".join(v + "\n" for v in videos))
    for video, (h, w) in zip(videos, ((131, 150), (136, 171))):  # the flow network needs >= 128 rows after padding to 8
        os.makedirs(os.path.join(root, "data", video, "origin"))
        os.makedirs(os.path.join(pred, video))
        base = rs.randint(0, 256, (h // 4 + 3, w // 4 + 3, 3)).astype(np.float32)
        lab = rs.randint(0, 124, (h // 8 + 2, w // 8 + 2))
        for t in range(4):
            img = np.kron(base, np.ones((4, 4, 1), np.float32))[t:t + h, 2 * t:2 * t + w]
            img = np.clip(img + rs.randn(h, w, 3) * 4.0, 0, 255).astype(np.uint8)
            Image.fromarray(img, "RGB").save(os.path.join(root, "data", video, "origin", "%08d.jpg" % (3 * t + 1)), quality=95)
            p = np.kron(lab, np.ones((8, 8), np.int64))[t:t + h, t:t + w]
            p = np.where(rs.rand(h, w) < 0.9, p, rs.randint(0, 124, (h, w))).astype(np.uint8)
            Image.fromarray(p, "L").save(os.path.join(pred, video, "%08d.png" % (3 * t + 1)))
    torch.manual_seed(0)
    model = TC_cal.load_raft("", dev)  # random-initialised flow network (no checkpoint on the box): any flow field does
    args = TC_cal.build_parser().parse_args(["--dataroot", root, "--pred", pred, "--split", "val.txt"])
    lines = []
    tc = TC_cal.main(args, model=model, log=lambda *a: lines.append(a))
    ev = Evaluator(124)
    for video in videos:
        frames = sorted(os.listdir(os.path.join(root, "data", video, "origin")))
        for i, name in enumerate(frames[:-1]):
            flow = TC_cal.pair_flow(model, TC_cal._frame(os.path.join(root, "data", video, "origin", name), dev),
                                    TC_cal._frame(os.path.join(root, "data", video, "origin", frames[i + 1]), dev))
            p0 = np.array(Image.open(os.path.join(pred, video, name.split(".")[0] + ".png")))
            p1 = np.array(Image.open(os.path.join(pred, video, frames[i + 1].split(".")[0] + ".png")))
            w = _cpu_warp(torch.from_numpy(p1)[None, None].float(), flow.cpu().contiguous())
            ev.add_batch(p0[None], w.int().squeeze(1).numpy())
    assert 0.0 < tc < 1.0 and tc == ev.Mean_Intersection_over_Union()


def test_change2_480p_equals_pil(dev, tmp_path):
    """A small "full resolution" tree -> 480p: decoded output frames and masks equal PIL's resize of the decoded
    inputs (bilinear / nearest), the tree layout is kept, a frame without a mask is still converted."""
    from cvpr2021_vspw_implement_amd import change2_480p as C

    src, dst = str(tmp_path / "src"), str(tmp_path / "dst")
    rs = np.random.RandomState(5)
    cases = [("v1", "00000001", 270, 480, True), ("v1", "00000004", 270, 480, False), ("v2", "00000001", 96, 131, True)]
    for video, stem, h, w, mask in cases:
        os.makedirs(os.path.join(src, "data", video, "origin"), exist_ok=True)
        os.makedirs(os.path.join(src, "data", video, "mask"), exist_ok=True)
        Image.fromarray(rs.randint(0, 256, (h, w, 3)).astype(np.uint8), "RGB").save(
            os.path.join(src, "data", video, "origin", stem + ".jpg"), quality=95)
        if mask:
            Image.fromarray(rs.randint(0, 125, (h, w)).astype(np.uint8), "L").save(
                os.path.join(src, "data", video, "mask", stem + ".png"))
    C.main(C.build_parser().parse_args(["--src", src, "--dst", dst]), log=lambda *a: None)
    for video, stem, h, w, mask in cases:
        nw = int(480 * w / h)
        ref = Image.open(os.path.join(src, "data", video, "origin", stem + ".jpg")).resize((nw, 480), Image.BILINEAR)
        ref.save(str(tmp_path / "ref.jpg"))
        got = np.array(Image.open(os.path.join(dst, "data", video, "origin", stem + ".jpg")))
        assert got.shape == (480, nw, 3) and np.array_equal(got, np.array(Image.open(str(tmp_path / "ref.jpg"))))
        mp = os.path.join(dst, "data", video, "mask", stem + ".png")
        assert os.path.exists(mp) == mask
        if mask:
            refm = Image.open(os.path.join(src, "data", video, "mask", stem + ".png")).resize((nw, 480), Image.NEAREST)
            assert np.array_equal(np.array(Image.open(mp)), np.array(refm))
