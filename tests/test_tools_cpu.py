"""Evaluation-side scripts of the reference (VC_perclip.py, change2_480p.py - host parts) against captures of the
reference's own functions (tests/golden/metric_tools_reference.npz, make_golden.py:case_metric_tools)."""
import os

import numpy as np
from PIL import Image

from helpers import golden


def test_vc_perclip_equals_reference(tmp_path):
    """get_common on the captured sequences, then VC_perclip.main end to end on PNG files of the same sequences: the
    score is the nan-mean of the reference's per-window accuracies; a video with <= clip_num frames is skipped."""
    from cvpr2021_vspw_implement_amd import VC_perclip as V
    from cvpr2021_vspw_implement_amd.utils import get_common

    fx = golden("metric_tools_reference")
    gts, preds = list(fx["vc:gt"]), list(fx["vc:pred"])
    h, w = gts[0].shape
    for cn in (2, 5):
        np.testing.assert_allclose(np.array(get_common(gts, preds, cn, h, w)), fx["vc:accs%d" % cn], rtol=1e-14, atol=0)
    root, pred = str(tmp_path / "tree"), str(tmp_path / "pred")
    with open_split(root, ["long", "short"]):
        pass
    for video, n in (("long", 12), ("short", 5)):
        os.makedirs(os.path.join(root, "data", video, "mask"))
        os.makedirs(os.path.join(pred, video))
        for t in range(n):
            Image.fromarray(gts[t].astype(np.uint8), "L").save(os.path.join(root, "data", video, "mask", "%04d.png" % t))
            Image.fromarray(preds[t].astype(np.uint8), "L").save(os.path.join(pred, video, "%04d.png" % t))
    lines = []
    args = V.build_parser().parse_args(["--dataroot", root, "--pred", pred, "--split", "val.txt", "--clip_num", "5"])
    score = V.main(args, log=lambda *a: lines.append(a))
    assert abs(score - np.nanmean(fx["vc:accs5"])) < 1e-15  # `short` (5 frames) contributes nothing


class open_split(object):
    def __init__(self, root, videos):
        os.makedirs(root, exist_ok=True)
        with open(os.path.join(root, "val.txt"), "w") as f:
            for v in videos:
                f.write(v + "\n")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def test_480p_target_size_and_flag_defaults():
    from cvpr2021_vspw_implement_amd import TC_cal, VC_perclip, change2_480p

    fx = golden("metric_tools_reference")
    for (w, h), want in zip(fx["size480:in"], fx["size480:out"]):
        assert change2_480p.target_size(int(w), int(h)) == (int(want), 480)
    a = TC_cal.build_parser().parse_args([])
    assert (a.num_class, a.max_videos, a.split, a.pred) == (124, 100, "val.txt", "./prediction")  # TC_cal.py:42-51,72
    v = VC_perclip.build_parser().parse_args([])
    assert (v.clip_num, v.split, v.pred) == (16, "val.txt", "./predicts")                          # VC_perclip.py:30-38
