"""ORACLE / test infrastructure: a tiny deterministic VSPW-format tree (dataset2.py:866-884 layout) and a numpy
restatement of what the reference's datasets do to a decoded frame (dataset2.py:921-977,1015-1035).  tests/golden/make_golden.py
writes the tree, runs the REFERENCE dataset classes on it and stores their outputs; the tests rebuild the same tree
(same PIL build => same bytes) and compare the HIP input pipeline with those outputs."""
import os
import zlib

import numpy as np
from PIL import Image

VIDEOS = (("v_a", 24, 36, 52), ("v_b", 9, 44, 60), ("v_c", 20, 50, 64))  # name, frames, h, w
MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def _rs(tag):
    return np.random.RandomState(zlib.crc32(tag.encode()) & 0x7FFFFFFF)


def make_tiny_vspw(root):
    os.makedirs(root, exist_ok=True)
    for split, vids in (("train", VIDEOS), ("val", VIDEOS[:2]), ("test", VIDEOS[1:])):
        with open(os.path.join(root, split + ".txt"), "w") as f:
            for v in vids:
                f.write(v[0] + "\n")
    for name, n, h, w in VIDEOS:
        od = os.path.join(root, "data", name, "origin")
        md = os.path.join(root, "data", name, "mask")
        os.makedirs(od, exist_ok=True)
        os.makedirs(md, exist_ok=True)
        rs = _rs("vspw:" + name)
        base = rs.randint(0, 256, size=(h // 4 + 2, w // 4 + 2, 3)).astype(np.float32)
        lab = rs.randint(0, 125, size=(h // 6 + 2, w // 6 + 2))
        for t in range(n):
            up = np.kron(base, np.ones((4, 4, 1), np.float32))[t % 4:t % 4 + h, (2 * t) % 4:(2 * t) % 4 + w]
            img = np.clip(up + rs.randn(h, w, 3) * 6.0 + t, 0, 255).astype(np.uint8)
            seg = np.kron(lab, np.ones((6, 6), np.int64))[t % 6:t % 6 + h, (t // 2) % 6:(t // 2) % 6 + w].astype(np.uint8)
            seg[rs.rand(h, w) < 0.02] = 255
            stem = "%08d" % (3 * t + 1)
            Image.fromarray(img, "RGB").save(os.path.join(od, stem + ".jpg"), quality=92)
            Image.fromarray(seg, "L").save(os.path.join(md, stem + ".png"))


def np_frame_transform(img_u8, seg_u8, flip, new_hw, crop):
    """The reference's per-frame chain on numpy/PIL: flip -> PIL resize -> /255 -> pad -> crop -> normalise;
    label: flip -> nearest resize -> pad 255 -> crop -> 0->255, v->v-1 (uint8) -> float."""
    im, sg = Image.fromarray(img_u8, "RGB"), Image.fromarray(seg_u8, "L")
    if flip:
        im, sg = im.transpose(Image.FLIP_LEFT_RIGHT), sg.transpose(Image.FLIP_LEFT_RIGHT)
    if new_hw is not None and tuple(new_hw) != img_u8.shape[:2]:
        im = im.resize((new_hw[1], new_hw[0]), Image.BILINEAR)
        sg = sg.resize((new_hw[1], new_hw[0]), Image.NEAREST)
    a = np.float32(np.array(im)) / 255.
    s = np.array(sg)
    if crop is not None:
        ph, pw, y, x, oh, ow = crop
        a = np.pad(a, ((ph, ph), (pw, pw), (0, 0)), "constant")[y:y + oh, x:x + ow]
        s = np.pad(s, ((ph, ph), (pw, pw)), "constant", constant_values=(255, 255))[y:y + oh, x:x + ow]
    a = (a.transpose(2, 0, 1) - MEAN[:, None, None]) / STD[:, None, None]
    s = s.copy()
    s[s == 0] = 255
    s = s - 1
    s[s == 254] = 255
    return a.astype(np.float32), s.astype(np.float32)[None]
