"""ORACLE-SIDE DIAGNOSTIC (test infrastructure, NOT the product): Winograd F(m x m, 3 x 3) convolution of a stride-1
3x3 convolution emulated in numpy with the product's arithmetic shape - float32 input / output transforms, one float32
GEMM per transform position whose k loop is a sequential fmaf chain (oracle/csrc/seq_gemm.c), weight transforms in
float32 or float64-then-rounded - for ALL THREE passes (forward, data gradient, weight gradient), so that the
pinned-decision gradient probe (tools/diag/wino_f33_probe.py) can price a tile size before any kernel is written.

Transforms come from the Cook-Toom construction over a list of finite interpolation points plus infinity (exact
rational arithmetic, `fractions`).  The convolution replaced is F.conv2d with kernel 3, stride 1, padding = dilation
(reference models/resnet.py:72-92 after the rewrite of models/models.py:737-750).  Dilation d = d x d independent
undilated convolutions on the sub-grids {y = sy, x = sx (mod d)} - as csrc/winograd.hip lays its tiles out.
"""
from fractions import Fraction

import numpy as np

from . import np_ops as O


def _poly_mul(p, q):
    r = [Fraction(0)] * (len(p) + len(q) - 1)
    for i, a in enumerate(p):
        for j, b in enumerate(q):
            r[i + j] += a * b
    return r


def cook_toom(m, points, r=3):
    """A^T [m, n], G [n, r], B^T [n, n] (n = m + r - 1) as float64 arrays; points = the n - 1 finite points."""
    n = m + r - 1
    pts = [Fraction(p) for p in points]
    if len(pts) != n - 1 or len(set(pts)) != n - 1:
        raise ValueError("F(%d,%d) needs %d distinct finite points" % (m, r, n - 1))
    at = [[Fraction(0)] * n for _ in range(m)]
    g = [[Fraction(0)] * r for _ in range(n)]
    bt = [[Fraction(0)] * n for _ in range(n)]
    full = [Fraction(1)]
    for a in pts:
        full = _poly_mul(full, [-a, Fraction(1)])
    for j, a in enumerate(pts):
        nj = Fraction(1)
        mj = [Fraction(1)]
        for k, b in enumerate(pts):
            if k != j:
                nj *= a - b
                mj = _poly_mul(mj, [-b, Fraction(1)])
        for i in range(m):
            at[i][j] = a ** i
        for k in range(r):
            g[j][k] = a ** k / nj
        for l, c in enumerate(mj):
            bt[j][l] = c
    at[m - 1][n - 1] = Fraction(1)
    g[n - 1][r - 1] = Fraction(1)
    for l, c in enumerate(full):
        bt[n - 1][l] = c
    f = lambda rows: np.array([[float(v) for v in row] for row in rows], dtype=np.float64)
    AT, G, BT = f(at), f(g), f(bt)
    # exactness check of the bilinear identity: sum_j AT[i,j] G[j,k] BT[j,l] = [l == i + k]
    t = np.einsum("ij,jk,jl->ikl", AT, G, BT)
    want = np.zeros_like(t)
    for i in range(m):
        for k in range(r):
            want[i, k, i + k] = 1
    if np.abs(t - want).max() > 1e-9:
        raise AssertionError("Cook-Toom identity fails for points %r" % (points,))
    return AT, G, BT


class Winograd:
    """conv(x, w) for kernel 3, stride 1, pad = dil through F(m x m, 3 x 3)."""

    def __init__(self, m, points, weight_f64=True, xform_f64=False):
        """weight_f64: G g G^T / G^T dU G evaluated in float64 and rounded once (weights are tiny: free on the device);
        xform_f64: the activation-side transforms (B^T d B, A^T M A, A dY A^T) likewise evaluated in float64 registers
        and rounded once when stored - the transforms are HBM-bound passes, the fp64 vector rate equals the fp32 one."""
        self.m, self.n = m, m + 2
        self.AT, self.G, self.BT = cook_toom(m, points)
        self.weight_f64 = weight_f64
        self.xform_f64 = xform_f64

    def _x(self, mat, t):
        """mat t mat^T on the last two axes in the transform precision, rounded to t's dtype."""
        dt = np.float64 if self.xform_f64 else t.dtype
        mm = mat.astype(dt)
        return np.matmul(np.matmul(mm, t.astype(dt)), mm.T).astype(t.dtype)

    # ---- pieces (all arithmetic in the dtype of the operand) ------------------------------------------------------
    def _tiles(self, h):
        return -(-h // self.m)

    def _patches(self, xs):
        """xs [n,c,h,w] (one sub-grid) -> d [n, th, tw, c, n, n] of overlapping (m+2)^2 patches, zero padded."""
        m, n = self.m, self.n
        nb, c, h, w = xs.shape
        th, tw = self._tiles(h), self._tiles(w)
        xp = np.zeros((nb, c, th * m + 2, tw * m + 2), dtype=xs.dtype)
        xp[:, :, 1:1 + h, 1:1 + w] = xs
        s = xp.strides
        d = np.lib.stride_tricks.as_strided(xp, (nb, th, tw, c, n, n), (s[0], s[2] * m, s[3] * m, s[1], s[2], s[3]))
        return d, th, tw

    def input_transform(self, xs):
        d, th, tw = self._patches(xs)
        v = self._x(self.BT, d)  # [n, th, tw, c, a, b]
        nb, c = xs.shape[:2]
        return np.ascontiguousarray(v.transpose(4, 5, 0, 1, 2, 3).reshape(self.n * self.n, nb * th * tw, c)), th, tw

    def weight_transform(self, w, dt):
        """w [k, c, 3, 3] -> U [n*n, k, c]"""
        g = self.G if self.weight_f64 else self.G.astype(dt)
        ww = w.astype(np.float64) if self.weight_f64 else w
        u = np.matmul(np.matmul(g, ww), g.T).astype(dt)
        return np.ascontiguousarray(u.transpose(2, 3, 0, 1).reshape(self.n * self.n, w.shape[0], w.shape[1]))

    def output_transform(self, mm, nb, th, tw, h, w):
        """mm [n*n, tiles, k] -> [nb, k, h, w]"""
        k = mm.shape[-1]
        t = mm.reshape(self.n, self.n, nb, th, tw, k).transpose(2, 3, 4, 5, 0, 1)
        y = self._x(self.AT, t)  # [nb, th, tw, k, m, m]
        y = y.transpose(0, 3, 1, 4, 2, 5).reshape(nb, k, th * self.m, tw * self.m)
        return y[:, :, :h, :w]

    def dy_transform(self, gs):
        """gs [nb, k, h, w] -> dM [n*n, tiles, k] = A dY A^T per tile (zero padded to whole tiles)."""
        m = self.m
        nb, k, h, w = gs.shape
        th, tw = self._tiles(h), self._tiles(w)
        gp = np.zeros((nb, k, th * m, tw * m), dtype=gs.dtype)
        gp[:, :, :h, :w] = gs
        t = gp.reshape(nb, k, th, m, tw, m).transpose(0, 2, 4, 1, 3, 5)
        dm = self._x(self.AT.T, t)  # [nb, th, tw, k, n, n]
        return np.ascontiguousarray(dm.transpose(4, 5, 0, 1, 2, 3).reshape(self.n * self.n, nb * th * tw, k))

    def dw_transform(self, du, dt):
        """du [n*n, k, c] -> dW [k, c, 3, 3] = G^T dU G"""
        g = self.G if self.weight_f64 else self.G.astype(dt)
        t = du.reshape(self.n, self.n, du.shape[1], du.shape[2]).transpose(2, 3, 0, 1)
        t = t.astype(np.float64) if self.weight_f64 else t
        return np.matmul(np.matmul(g.T, t), g).astype(dt)

    # ---- whole passes ---------------------------------------------------------------------------------------------
    def forward(self, x, w, dil, keep=None):
        nb, c, h, wd = x.shape
        k = w.shape[0]
        u = self.weight_transform(w, x.dtype)
        y = np.empty((nb, k, h, wd), dtype=x.dtype)
        for sy in range(dil):
            for sx in range(dil):
                xs = x[:, :, sy::dil, sx::dil]
                v, th, tw = self.input_transform(xs)
                if keep is not None:
                    keep[(sy, sx)] = v
                mm = O._mm_nt(v, u)
                y[:, :, sy::dil, sx::dil] = self.output_transform(mm, nb, th, tw, xs.shape[2], xs.shape[3])
        return y

    def backward_data(self, g, w, dil):
        wr = np.ascontiguousarray(w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))
        return self.forward(g, wr, dil)

    def backward_weight(self, g, x, dil, kept=None, chunk=1024):
        k, c = g.shape[1], x.shape[1]
        du = np.zeros((self.n * self.n, k, c), dtype=np.float64)
        for sy in range(dil):
            for sx in range(dil):
                v = kept[(sy, sx)] if kept else self.input_transform(x[:, :, sy::dil, sx::dil])[0]
                dm = self.dy_transform(g[:, :, sy::dil, sx::dil])
                du += O._mm_nt(np.ascontiguousarray(dm.transpose(0, 2, 1)), np.ascontiguousarray(v.transpose(0, 2, 1)), chunk)
        return self.dw_transform(du.astype(g.dtype), g.dtype)


_active = {"wino": None, "min_c": 128, "count": 0}


def install(wino, min_c=128):
    """Route every eligible O.conv2d (3x3, stride 1, pad == dil, min(c, k) >= min_c - the product's `_wino_ok`) of the
    float32 oracle through `wino` (None: back to the direct convolution).  float64 runs are never rerouted."""
    _active["wino"], _active["min_c"], _active["count"] = wino, min_c, 0
    if not hasattr(O, "_conv2d_direct"):
        O._conv2d_direct = O.conv2d

        def conv2d(x, w, b=None, stride=1, pad=0, dil=1):
            wn = _active["wino"]
            k, c, kh, kw = w.v.shape
            if (wn is None or O.F32 is not np.float32 or kh != 3 or kw != 3 or stride != 1 or pad != dil
                    or min(c, k) < _active["min_c"]):
                return O._conv2d_direct(x, w, b, stride, pad, dil)
            _active["count"] += 1
            kept = {}
            y = wn.forward(x.v, w.v, dil, kept if w.needs else None)
            if b is not None:
                y = y + b.v.reshape(1, -1, 1, 1)
            o = O._out(y, *((x, w) + ((b,) if b is not None else ())))

            def bw(g):
                if w.needs:
                    w.acc(wn.backward_weight(g, x.v, dil, kept))
                if b is not None and b.needs:
                    b.acc(g.sum((0, 2, 3), dtype=np.float64))
                if x.needs:
                    x.acc(wn.backward_data(g, w.v, dil))

            return O._rec(o, bw)

        O.conv2d = conv2d
