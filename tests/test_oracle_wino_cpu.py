"""oracle/np_wino.py (the Winograd emulation behind tools/diag/wino_f33_probe.py) and the transform tables compiled into
csrc/winograd_f3.hip: the Cook-Toom construction is exact (float64 Winograd == float64 direct convolution in all three
passes, ragged tiles and dilation sub-grids included), and the kernel's B^T / A^T / G tables are that construction's output
for the shipped point sets."""
import os
import re
from fractions import Fraction

import numpy as np
import pytest

from oracle import np_ops as O
from oracle import np_wino as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POINTS = {2: [0, 1, -1], 3: [0, 1, -1, 2], 4: [0, 1, -1, Fraction(1, 2), -2],
          5: [0, 1, -1, Fraction(1, 2), Fraction(-1, 2), 2]}


@pytest.mark.parametrize("m", [2, 3, 4, 5])
@pytest.mark.parametrize("dil,h,w", [(1, 7, 9), (2, 11, 8), (4, 15, 13)])
def test_winograd_emulation_is_exact_in_float64(m, dil, h, w):
    rng = np.random.default_rng(10 * m + dil)
    x = rng.standard_normal((2, 5, h, w))
    wt = rng.standard_normal((4, 5, 3, 3))
    g = rng.standard_normal((2, 4, h, w))
    O.set_dtype(np.float64)
    try:
        xv, wv = O.Var(x, True), O.Var(wt, True)
        y = O.conv2d(xv, wv, None, 1, dil, dil)
        y.g = g
        for fn in reversed(O.tape().steps):
            fn()
        O.tape().steps = []
        wn = W.Winograd(m, POINTS[m], weight_f64=True)
        kept = {}
        yw = wn.forward(x, wt, dil, kept)
        assert np.abs(yw - y.v).max() < 1e-11
        assert np.abs(wn.backward_data(g, wt, dil) - xv.g).max() < 1e-11
        assert np.abs(wn.backward_weight(g, x, dil, kept) - wv.g).max() < 1e-10
    finally:
        O.set_dtype(np.float32)


@pytest.mark.parametrize("m", [3, 4, 5])
def test_kernel_tables_are_the_cook_toom_construction(m):
    src = open(os.path.join(ROOT, "cvpr2021_vspw_implement_amd", "csrc", "winograd_f3.hip")).read()

    def table(name):
        body = re.search(r"%s\[\d+\]\[\d+\] = \{(.*?)\};" % name, src, re.S).group(1)
        rows = re.findall(r"\{([^{}]*)\}", body)

        def val(tok):
            tok = tok.strip().replace("f", "")
            if "/" in tok:
                a, b = tok.split("/")
                return float(a) / float(b)
            return float(tok)

        return np.array([[val(t) for t in r.split(",")] for r in rows])

    AT, G, BT = W.cook_toom(m, POINTS[m])
    assert np.array_equal(table("kBT%d" % m), BT)
    assert np.array_equal(table("kAT%d" % m), AT)
    assert np.abs(table("kG%d" % m) - G).max() < 1e-16
