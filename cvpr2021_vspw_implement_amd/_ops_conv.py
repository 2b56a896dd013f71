"""Convolution operators of the hot path (see ops.py): forward / data gradient / weight gradient through the implicit
GEMM and Winograd kernels, the derived-weight caches, the weight-gradient side stream and the plain conv2d node."""
import ctypes
import os

import torch

from . import _C
from ._C import ConvDesc
from ._opbase import (_Timed, _conv_desc, _conv_flops, _conv_tag, _ktimer, _p, _require_gpu, _stream, _ws, empty_nhwc,
                      is_nhwc, to_nhwc)

# --------------------------------------------------------------------------------------------------- conv
# Forward apply deferred into the consumer (residual blocks): a conv+BN+residual+ReLU node whose output z has exactly
# one next reader - a pointwise conv - leaves z unwritten and hands (y, scale/shift, residual) to that conv, whose GEMM
# evaluates z while staging its A operand and writes it for everyone else (vspw_conv2d_fwd_apply).  Saves the separate
# read-read-write pass of vspw_bn_apply plus the GEMM's own read of z.
_fwd_apply = {"enabled": os.environ.get("VSPW_NO_FWD_APPLY", "0") != "1", "nodes": 0,
              # ... and into the input transform of a Winograd 3x3 reader (conv1 -> conv2 of a bottleneck)
              "wino": os.environ.get("VSPW_NO_FWD_APPLY_WINO", "0") != "1", "wino_nodes": 0}


# Winograd F(2x2,3x3) for stride-1 3x3 convolutions, forward and data gradient (csrc/winograd.hip): 4/9 of the direct
# multiplications, run by the pointwise MFMA kernel as 16 batched GEMMs.  VSPW_WINOGRAD=0 switches back to the direct
# implicit GEMM; VSPW_WINO_MINC = smallest channel count (both sides) that takes this path.
_wino = {"enabled": os.environ.get("VSPW_WINOGRAD", "1") == "1", "min_c": int(os.environ.get("VSPW_WINO_MINC", "128")),
         "wgrad": os.environ.get("VSPW_WINO_WGRAD", "1") == "1", "launches": 0,
         "keep_v": os.environ.get("VSPW_WINO_KEEP_V", "1") == "1",
         # the GEMM evaluates the input transform itself (vspw_wino_gemm_fused); forward: off, its V is reused by the
         # weight gradient
         "fuse_fwd": os.environ.get("VSPW_WINO_FUSE_FWD", "0") == "1",
         "fuse_dgrad": os.environ.get("VSPW_WINO_FUSE_DGRAD", "1") == "1",
         "fuse_max_rows": int(os.environ.get("VSPW_WINO_FUSE_MAXROWS", "512")),
         # the four GEMMs of a transform row in one workgroup (csrc/wino_rows.hip) where the library expects it to win
         "rows": os.environ.get("VSPW_WINO_ROWS", "1") == "1",
         # F(3x3,3x3) / F(4x4,3x3) (csrc/winograd_f3.hip): 25 / 36 GEMMs over 3x3 / 4x4 output tiles - 25/81 resp. 36/144 of the
         # direct multiplications instead of F(2x2)'s 36/81.  "tile": 0 = per geometry the size that executes fewer
         # multiplications (3 divides the 30 / 15 pixel sub-grids of the dilated stages, 4 divides 60), 2 / 3 / 4 = forced
         # (VSPW_WINO_TILE; VSPW_WINO_F3=0 is the old spelling of 2).  VSPW_WINO_F3_MINC: smallest channel count (both
         # sides) that leaves F(2x2)
         "tile": 2 if os.environ.get("VSPW_WINO_F3", "1") != "1" else int(os.environ.get("VSPW_WINO_TILE", "0")),
         "f3_min_c": int(os.environ.get("VSPW_WINO_F3_MINC", "128")), "f3_launches": 0, "f4_launches": 0, "f5_launches": 0,
         # F(5x5,3x3) (49/225 of the direct multiplications, conv-level rounding error 2.8x F(3x3)'s): taken in automatic mode
         # where min(Cin, Cout) >= f5_min_c and max(Cin, Cout) >= f5_min_cmax (see _wino_f3)
         "f5_dgrad": os.environ.get("VSPW_WINO_F5_DGRAD", "1") == "1",
         "f5_min_c": int(os.environ.get("VSPW_WINO_F5_MINC", "512")), "f5_min_cmax": int(os.environ.get("VSPW_WINO_F5_MINCMAX", "0"))}
_wino_tile_cache = {}


_strided_pw = {"enabled": os.environ.get("VSPW_STRIDED_PW_DGRAD", "1") == "1"}  # compact GEMM + scatter (conv2d_backward_data)


def set_accum_chunk(k):
    """Two-level accumulation of the pointwise GEMMs with K >= 2k (csrc/conv_igemm.hip, vspw_set_accum_chunk): k terms
    per fp32 chain (multiple of 32), 0 = one k-sequential chain (the library default).  Returns the previous setting."""
    prev = int(_C.query("vspw_get_accum_chunk"))
    _C.call("vspw_set_accum_chunk", int(k))
    return prev


def accum_chunk_supported():
    """The two-level accumulation variants exist in diagnostic builds only (-DVSPW_WITH_ACCUM_CHUNK)."""
    return int(_C.query("vspw_accum_chunk_compiled")) == 1


def set_winograd(enabled):
    _wino["enabled"] = bool(enabled)


def _wino_ok(d):
    return (_wino["enabled"] and d.kh == 3 and d.kw == 3 and d.stride == 1 and min(d.c, d.k) >= _wino["min_c"]
            and _C.query("vspw_wino_supported", ctypes.byref(d)) == 1)


def set_winograd_f3(enabled):
    """False: F(2x2,3x3) everywhere (round-5 behaviour); True: automatic tile size.  Returns the previous setting as a
    value this function accepts back (True / False / a forced tile)."""
    return set_winograd_tile(0 if enabled is True else (2 if enabled is False else int(enabled)))


def set_winograd_tile(m):
    """0: automatic, 2 / 3 / 4: force F(m x m, 3x3) where the geometry allows it.  Returns the previous value."""
    if m not in (0, 2, 3, 4, 5):
        raise ValueError("Winograd tile %r" % (m,))
    prev = _wino["tile"]
    _wino["tile"] = m
    return False if prev == 2 else (True if prev == 0 else prev)


class winograd_tile_hint(object):
    """`with winograd_tile_hint(5): ...` - the FORWARD convolutions issued inside take F(m x m, 3x3) where the geometry
    allows it and the mode is automatic (their weight gradients follow through the kept V).  The model uses it for where
    a layer sits in the network: rounding error injected late is amplified by few later blocks (models/resnet.py)."""

    def __init__(self, m):
        self.m = m

    def __enter__(self):
        self.prev, _wino["hint"] = _wino.get("hint", 0), (self.m or 0)

    def __exit__(self, *a):
        _wino["hint"] = self.prev
        return False


def _wino_f3(d, data_gradient=False):
    """Output tile edge m (3, 4 or 5) when this (Winograd-eligible, see _wino_ok) convolution leaves F(2x2), else 0.
    Forward and weight gradient decide alike (they share V).  The DATA gradient shares nothing with them and its result
    never reaches a forward activation, so in automatic mode it takes F(5x5) wherever that is supported (layers 2 / 3:
    184 -> 153 us per 256-channel launch) - `VSPW_WINO_F5_DGRAD=0`: same tile as the other two passes."""
    t = _wino["tile"]
    if t == 2 or min(d.c, d.k) < _wino["f3_min_c"]:
        return 0
    h = _wino.get("hint", 0)
    if not data_gradient and t == 0 and h in (3, 4, 5):
        key = ("hint", h, d.n, d.h, d.w, d.c, d.k, d.dil, d.pad, d.pad_w, d.stride, d.kh, d.kw)
        m = _wino_tile_cache.get(key)
        if m is None:
            m = _wino_tile_cache[key] = h if _C.query("vspw_wino%d_supported" % h, ctypes.byref(d)) == 1 else 0
        if m:
            return m
    if data_gradient and t == 0 and _wino["f5_dgrad"]:
        key = ("dgrad", d.n, d.h, d.w, d.c, d.k, d.dil, d.pad, d.pad_w, d.stride, d.kh, d.kw)
        m = _wino_tile_cache.get(key)
        if m is None:
            m = _wino_tile_cache[key] = 5 if _C.query("vspw_wino5_supported", ctypes.byref(d)) == 1 else 0
        if m:
            return m
    key = (t, _wino["f5_min_c"], _wino["f5_min_cmax"], d.n, d.h, d.w, d.c, d.k, d.dil, d.pad, d.pad_w, d.stride, d.kh, d.kw)
    m = _wino_tile_cache.get(key)
    if m is None:
        ok3 = _C.query("vspw_wino3_supported", ctypes.byref(d)) == 1
        ok4 = _C.query("vspw_wino4_supported", ctypes.byref(d)) == 1
        if t == 3:
            m = 3 if ok3 else 0
        elif t == 5:
            m = 5 if _C.query("vspw_wino5_supported", ctypes.byref(d)) == 1 else (3 if ok3 else 0)
        elif t == 4:
            m = 4 if ok4 else (3 if ok3 else 0)
        elif (min(d.c, d.k) >= _wino["f5_min_c"] and max(d.c, d.k) >= _wino["f5_min_cmax"]
              and _C.query("vspw_wino5_supported", ctypes.byref(d)) == 1):
            m = 5
        elif ok3 and ok4:
            # executed multiplications per (cin, cout) pair = tiles x positions.  Measured (tools/diag/wino3_probe.py,
            # profiles/r06_wino34_probe.log, us for the three passes, F(3x3) -> F(4x4)): F(4x4) wins where the GEMMs dominate
            # its 36-plane transforms, i.e. from 512 channels on - the heads' 1024 / 4096 -> 512 on undilated 60x60 maps
            # (19 % fewer multiplications) 2852 -> 2389 and 2384 -> 2026, the dilated 512 -> 512 of layer 4 (30 -> 32 / 15 -> 16
            # padding: 8 % fewer) 1529 -> 1442; it loses on 256 channels with dilation (507 -> 518) and ties on 128 (195 -> 196)
            t3, t4 = int(_C.query("vspw_wino3_tiles", ctypes.byref(d))), int(_C.query("vspw_wino4_tiles", ctypes.byref(d)))
            # (... and from 256 channels on where 4 divides the sub-grid exactly - 19 % fewer: layer3.0's 490 -> 458)
            m = 4 if ((36 * t4 <= 0.95 * 25 * t3 and min(d.c, d.k) >= 512) or
                      (36 * t4 <= 0.85 * 25 * t3 and min(d.c, d.k) >= 256)) else 3
        else:
            m = 3 if ok3 else (4 if ok4 else 0)
        _wino_tile_cache[key] = m
    return m


def _wino3_conv(d, src, w, rows, reduce_c, data_gradient, bias, dst, front=None, part=None, what="fwd", u=None,
                addend=None, act=0, m=None, pending=None):
    """_wino_conv through F(m x m, 3x3), m = 3 or 4: input transform, (m+2)^2 batched GEMMs, output transform
    (winograd_f3.hip)."""
    dev, st = src.device, _stream()
    m = m or _wino_f3(d, data_gradient)
    P, api = (m + 2) * (m + 2), "vspw_wino%d_" % m
    T = int(_C.query(api + "tiles", ctypes.byref(d)))
    if u is None:
        u = _wino3_weights(w, data_gradient, m)
    v = torch.empty((P, T, reduce_c), device=dev, dtype=torch.float32)
    if pending is not None:  # src = relu(scale*y + shift) is evaluated (and written) by the transform, see _fwd_apply
        _C.call(api + "input_apply", ctypes.byref(d), _p(pending[0]), _p(pending[1]), _p(src), reduce_c, _p(v), st)
    else:
        _C.call(api + "input", ctypes.byref(d), _p(src), reduce_c, _p(v), st)
    mm = torch.empty((P, T, rows), device=dev, dtype=torch.float32)
    with _Timed("igemm_nt_kernel", 2.0 * P * T * rows * reduce_c, _conv_tag(d, what + "-wino%d" % m), _conv_flops(d)):
        _C.call("vspw_bmm_nt", _p(v), _p(u), _p(mm), P, T, rows, reduce_c, st)
    z = y_ = mean = invstd = None
    if front is not None:
        z, y_, mean, invstd = front
    _C.call(api + "output", ctypes.byref(d), _p(mm), rows, _p(bias), _p(dst), _p(z), _p(y_), _p(mean), _p(invstd),
            _p(part), _p(addend), act, st)
    _wino["launches"] += 1
    _wino["f%d_launches" % m] += 1
    return v


def _wino_conv(d, src, w, rows, reduce_c, data_gradient, bias, dst, front=None, part=None, what="fwd", u=None,
               addend=None, act=0, fuse=None, pending=None):
    """dst = conv(src) through U, V, M (see winograd.hip); rows = output channels, reduce_c = channels of src.
    u: transformed weights supplied by the caller (inference: of the BatchNorm-folded weights).
    pending = (y_prev, scale_shift): src has not been written - the input transform evaluates it (see _fwd_apply)."""
    if u is None and _wino_f3(d, data_gradient):
        return _wino3_conv(d, src, w, rows, reduce_c, data_gradient, bias, dst, front, part, what, None, addend, act,
                           pending=pending)
    dev = src.device
    st = _stream()
    T = int(_C.query("vspw_wino_tiles", ctypes.byref(d)))
    if u is None:
        u = _wino_weights(w, data_gradient)
    v = None
    # measured (bench shapes): staging the transform costs the GEMM ~10 % (4 loads + 16 VALU per staged float4 on the
    # lanes fp32 MFMA shares), the separate transform pass costs time proportional to the INPUT only: fusing wins up
    # to 512 output rows (256->256: -31 us per launch) and loses beyond (512->1024, 512->4096)
    if fuse is None:
        fuse = _wino["fuse_dgrad"] if data_gradient else _wino["fuse_fwd"]
    fused = fuse and rows <= _wino["fuse_max_rows"]
    if pending is not None and fused:
        raise RuntimeError("deferred input + fused Winograd operand (see _wino_takes_pending)")
    # row-fused form (csrc/wino_rows.hip): the four GEMMs of a transform row in one workgroup, half of the output
    # transform in its registers - the GEMM writes (and the output transform reads) 8 planes instead of 16
    tpad = 0
    if _wino["rows"] and _C.query("vspw_wino_rows_prefer", ctypes.byref(d), reduce_c, rows, 1 if fused else 0) == 1:
        tpad = int(_C.query("vspw_wino_rows_tpad", ctypes.byref(d), reduce_c, rows, 1 if fused else 0))
    m = torch.empty((8, tpad, rows) if tpad else (16, T, rows), device=dev, dtype=torch.float32)
    if fused:  # the input transform is evaluated by the GEMM while it stages its A operand: V is never written
        with _Timed("igemm_nt_kernel", 2.0 * 16 * T * rows * reduce_c, _conv_tag(d, what + "-winof"), _conv_flops(d)):
            if tpad:
                _C.call("vspw_wino_gemm_fused_rows", ctypes.byref(d), _p(src), reduce_c, _p(u), rows, _p(m), st)
            else:
                _C.call("vspw_wino_gemm_fused", ctypes.byref(d), _p(src), reduce_c, _p(u), rows, _p(m), st)
    else:
        v = torch.empty((16, T, reduce_c), device=dev, dtype=torch.float32)
        if pending is not None:
            _C.call("vspw_wino_input_apply", ctypes.byref(d), _p(pending[0]), _p(pending[1]), _p(src), reduce_c, _p(v), st)
        else:
            _C.call("vspw_wino_input", ctypes.byref(d), _p(src), reduce_c, _p(v), st)
        with _Timed("igemm_nt_kernel", 2.0 * 16 * T * rows * reduce_c, _conv_tag(d, what + "-wino"), _conv_flops(d)):
            if tpad:
                _C.call("vspw_wino_gemm_rows", ctypes.byref(d), _p(v), reduce_c, _p(u), rows, _p(m), st)
            else:
                _C.call("vspw_bmm_nt", _p(v), _p(u), _p(m), 16, T, rows, reduce_c, st)
    z = y_ = mean = invstd = None
    if front is not None:
        z, y_, mean, invstd = front
    if tpad:
        _C.call("vspw_wino_output_rows", ctypes.byref(d), _p(m), tpad, rows, _p(bias), _p(dst), _p(z), _p(y_), _p(mean),
                _p(invstd), _p(part), _p(addend), act, st)
    else:
        _C.call("vspw_wino_output", ctypes.byref(d), _p(m), rows, _p(bias), _p(dst), _p(z), _p(y_), _p(mean), _p(invstd),
                _p(part), _p(addend), act, st)
    _wino["launches"] += 1
    return v


def _wino_takes_pending(d, pending, wgrad):
    """A deferred input (see _fwd_apply) can be evaluated by the Winograd input transform when that transform is a pass
    of its own (V kept for the weight gradient; the fused-operand GEMM reads every pixel four times per position) and
    the deferred node has no residual branch."""
    return (_fwd_apply["wino"] and pending[2] is None and _wino["keep_v"] and bool(wgrad) and _wino["wgrad"]
            and not _wino["fuse_fwd"] and d.c % 4 == 0)


def conv2d_forward(x, w, bias, stride, pad, dil, want_stats=False, pending=None, wgrad=True):
    """x NHWC-memory [N,C,H,W]; w [K,C,KH,KW] in channels_last memory ([K][KH][KW][C]).
    pending = (y_prev, scale_shift, residual): x has not been written yet - it is relu(scale*y_prev + shift +
    residual) of the node that produced it; this (pointwise) GEMM evaluates it while staging and fills x.
    wgrad: a weight gradient will be asked for (the autograd node's needs_input_grad of w)."""
    _require_gpu(x, "conv2d")
    x = to_nhwc(x)
    if not is_nhwc(w):
        w = w.contiguous(memory_format=torch.channels_last)
    k, c, kh, kw = w.shape
    if c != x.shape[1]:
        raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (x.shape[1], c))
    d = _conv_desc(x, k, kh, kw, stride, pad, dil)
    y = empty_nhwc(d.n, k, d.oh, d.ow, x.device)
    part = None
    if _wino_ok(d) and (pending is None or _wino_takes_pending(d, pending, wgrad)):
        if want_stats:
            fm = _wino_f3(d)
            nparts = _C.query(("vspw_wino%d_stat_partials" % fm) if fm else "vspw_wino_stat_partials", ctypes.byref(d))
            part = torch.empty((nparts, 2, k), device=x.device, dtype=torch.float32)
        # the input transform is kept for this convolution's weight gradient (same V: saves its recomputation there) -
        # only when there will be one: frozen weights / no_grad evaluation take the GEMM that transforms its A operand
        # itself (V, four times the size of x, is then never written)
        needs_v = _wino["keep_v"] and bool(wgrad) and _wino["wgrad"]
        v = _wino_conv(d, x, w, k, c, False, bias, y, part=part, fuse=None if needs_v else True,
                       pending=None if pending is None else (pending[0], pending[1]))
        if needs_v and v is not None:
            y._vspw_wino_v = v  # picked up (and removed) by the autograd node that called us
        return y, part, d
    if want_stats:
        tiles = _C.query("vspw_conv2d_stats_partials", ctypes.byref(d))
        part = torch.empty((tiles, 2, k), device=x.device, dtype=torch.float32)
    with _Timed("igemm_nt_kernel", _conv_flops(d), _conv_tag(d, "fwd")):
        if pending is not None:
            py, pss, pres = pending
            _C.call("vspw_conv2d_fwd_apply", ctypes.byref(d), _p(py), _p(pres), _p(pss), _p(x), _p(w), _p(bias), _p(y),
                    _p(part), _stream())
        else:
            _C.call("vspw_conv2d_fwd", ctypes.byref(d), _p(x), _p(w), _p(bias), _p(y), _p(part), _stream())
    return y, part, d


_wt_cache = {"gen": 0}  # generation counter shared by every derived-weight cache (see invalidate_inference_cache)
_WT_ENTRY = None


def _wt_key(w):
    return (w.data_ptr(), w._version, _wt_cache["gen"], tuple(w.shape))


class _DerivedWeights(object):
    """Per-step cache of tensors derived from convolution weights (the [Cin][taps][Cout] copies of the data-gradient
    GEMMs; the Winograd transforms).  Weights change once per step (the optimizer), so the derived tensors are
    refreshed once per step - ALL of them by one multi-tensor launch over a device table (struct vspw_wt_entry),
    triggered by the first use that finds its entry stale - instead of one small launch per layer inside the critical
    path.  alloc(w) -> buffer; single(w, buf, stream); multi = C entry point taking (table, n, tiles, stream);
    tiles(k, c, kh, kw) -> workgroups of one tensor in the multi launch."""

    def __init__(self, alloc, single, multi, tiles):
        self.alloc, self.single, self.multi, self.tiles = alloc, single, multi, tiles
        self.clear()

    def clear(self):
        self.entries, self.order, self.table, self.table_n, self.total = {}, [], None, 0, 0

    def _upload(self, device):
        import numpy as np

        global _WT_ENTRY
        if _WT_ENTRY is None:  # struct vspw_wt_entry (include/vspw_hip.h)
            _WT_ENTRY = np.dtype([("w", "<u8"), ("wT", "<u8"), ("tile0", "<i8"), ("k", "<i4"), ("taps", "<i4"),
                                  ("c", "<i4"), ("reserved", "<i4")])
        ents = [self.entries[i] for i in self.order]
        rec = np.zeros(len(ents), dtype=_WT_ENTRY)
        t0 = 0
        for i, e in enumerate(ents):
            k, c, kh, kw = e["shape"]
            rec[i] = (e["ptr"], e["buf"].data_ptr(), t0, k, kh * kw, c, 0)
            t0 += int(self.tiles(k, c, kh, kw))
        self.table = torch.from_numpy(rec.view(np.uint8).copy()).to(device)
        self.table_n = len(ents)
        self.total = t0

    def get(self, w):
        import weakref

        ents = self.entries
        ident = (w.data_ptr(), tuple(w.shape))
        e = ents.get(ident)
        key = _wt_key(w)
        if e is not None and e["ref"]() is None:
            # the tensor this entry was made for is gone: its storage may have been freed and handed to ANOTHER weight
            # with the same address / shape / version, so nothing cached under this identity can be trusted
            del ents[ident]
            self.order = [i for i in self.order if i != ident]
            self.table = None
            e = None
        if e is not None and e["key"] == key:
            return e["buf"]
        capturing = torch.cuda.is_current_stream_capturing()
        if e is None:
            # first sight of this weight: own launch now, member of the batched refresh from the next step on
            buf = self.alloc(w)
            if capturing:  # a buffer from the graph's private pool must not leak into the eager cache
                self.single(w, buf, _stream())
                return buf
            ents[ident] = e = {"buf": buf, "ptr": w.data_ptr(), "shape": tuple(w.shape), "key": None,
                               "ref": weakref.ref(w)}
            self.order.append(ident)
            self.table = None
        if not capturing:
            dead = [i for i, x in ents.items() if x["ref"]() is None]
            if dead:  # weights of a model that no longer exists
                for i in dead:
                    del ents[i]
                self.order = [i for i in self.order if i in ents]
                self.table = None
            if self.table is None and len(ents) > 1 and all(
                    x["key"] is None or x["key"][2] != _wt_cache["gen"] for x in ents.values()):
                self._upload(w.device)
        if self.table is not None and self.table_n == len(ents):
            # refresh every registered tensor in one launch (they all went stale together: same optimizer step)
            _C.call(self.multi, _p(self.table), self.table_n, self.total, _stream())
            for x in ents.values():
                t = x["ref"]()
                x["key"] = _wt_key(t) if t is not None else None
            e["key"] = key
            return e["buf"]
        self.single(w, e["buf"], _stream())
        e["key"] = key
        return e["buf"]


def _wt_alloc(w):
    k, c, kh, kw = w.shape
    return torch.empty((c, kh, kw, k), device=w.device, dtype=torch.float32)


def _wt_single(w, buf, st):
    k, c, kh, kw = w.shape
    _C.call("vspw_weight_transpose", _p(w), _p(buf), k, kh * kw, c, st)


_wt_copies = _DerivedWeights(_wt_alloc, _wt_single, "vspw_weight_transpose_multi",
                             lambda k, c, kh, kw: _C.query("vspw_weight_transpose_tiles", k, kh * kw, c))


def _transposed_weight(w):
    """wT for the data gradient of a conv with weight w ([K][KH][KW][C] memory), from the per-step cache."""
    return _wt_copies.get(w)


def _wu_alloc(w):
    k, c, kh, kw = w.shape
    return torch.empty((2, 16, k * c), device=w.device, dtype=torch.float32)


def _wu_single(w, buf, st):
    k, c, kh, kw = w.shape
    _C.call("vspw_wino_weights", _p(w), _p(buf[0]), k, c, 0, st)
    _C.call("vspw_wino_weights", _p(w), _p(buf[1]), k, c, 1, st)


_wu_copies = _DerivedWeights(_wu_alloc, _wu_single, "vspw_wino_weights_multi",
                             lambda k, c, kh, kw: _C.query("vspw_wino_weight_tiles", k, c))


def _wum_cache(m):
    P = (m + 2) * (m + 2)

    def alloc(w):
        k, c, kh, kw = w.shape
        return torch.empty((2, P, k * c), device=w.device, dtype=torch.float32)

    def single(w, buf, st):
        k, c, kh, kw = w.shape
        _C.call("vspw_wino%d_weights" % m, _p(w), _p(buf[0]), k, c, 0, st)
        _C.call("vspw_wino%d_weights" % m, _p(w), _p(buf[1]), k, c, 1, st)

    return _DerivedWeights(alloc, single, "vspw_wino%d_weights_multi" % m,
                           lambda k, c, kh, kw: _C.query("vspw_wino_weight_tiles", k, c))


_wu3_copies = {3: _wum_cache(3), 4: _wum_cache(4), 5: _wum_cache(5)}


def _wino3_weights(w, data_gradient, m=3):
    """U [(m+2)^2][Cout][Cin] (forward) or U' [..][Cin][Cout] (data gradient) of a 3x3 weight, from the per-step cache."""
    return _wu3_copies[m].get(w)[1 if data_gradient else 0]


def _wino_weights(w, data_gradient):
    """U [16][Cout][Cin] (forward) or U' [16][Cin][Cout] (data gradient) of a 3x3 weight, from the per-step cache."""
    return _wu_copies.get(w)[1 if data_gradient else 0]


def drop_weight_transpose_cache():
    _wt_copies.clear()
    _wu_copies.clear()
    for cache in _wu3_copies.values():
        cache.clear()


def conv2d_backward_data(dy, w, d, addend=None, bn_front=None, aff=None):
    """dx = conv_backward_input(dy, w) [+ addend, folded into the GEMM epilogue].
    bn_front = (z, link): additionally apply the ReLU mask of the node that produced this conv's input z and leave the
    two batch-norm-backward reductions of that node in link.partials (see BNLink); returns the masked gradient.
    aff = (y, coef): `dy` is really g, the gradient w.r.t. the BatchNorm OUTPUT; the GEMM stages
    coef[0]*g + coef[1]*y + coef[2] (BatchNorm's backward apply) as its operand (pointwise convs only)."""
    k, c, kh, kw = w.shape
    dx = empty_nhwc(d.n, d.c, d.h, d.w, dy.device)
    if aff is None and addend is None and _wino_ok(d):
        front = part = None
        if bn_front is not None:
            z, link = bn_front
            front = (z, link.y, link.mean, link.invstd)
            fm = _wino_f3(d, True)
            part = torch.empty((_C.query(("vspw_wino%d_stat_partials" % fm) if fm else "vspw_wino_stat_partials",
                                         ctypes.byref(d)), 2, d.c), device=dy.device, dtype=torch.float32)
        _wino_conv(d, dy, w, d.c, d.k, True, None, dx, front=front, part=part, what="dgrad")
        if bn_front is not None:
            link.partials, link.g = part, dx
        return dx
    wT = _transposed_weight(w)
    if aff is not None:
        y_, coef = aff
        if addend is not None:
            addend = to_nhwc(addend)
        zz = link = part = None
        if bn_front is not None:
            zz, link = bn_front
            tiles = _C.query("vspw_conv2d_bwd_data_bn_partials", ctypes.byref(d))
            part = torch.empty((tiles, 2, d.c), device=dy.device, dtype=torch.float32)
        with _Timed("igemm_nt_kernel", _conv_flops(d), _conv_tag(d, "dgrad")):
            _C.call("vspw_conv2d_bwd_data_aff", ctypes.byref(d), _p(dy), _p(y_), _p(coef), _p(wT), _p(addend), _p(zz),
                    _p(link.y) if link else None, _p(link.mean) if link else None,
                    _p(link.invstd) if link else None, _p(dx), _p(part), _stream())
        if link is not None:
            link.partials, link.g = part, dx
        return dx
    if addend is not None:
        addend = to_nhwc(addend)
        if tuple(addend.shape) != tuple(dx.shape):
            raise RuntimeError("conv2d_backward_data: addend %s vs dx %s" % (tuple(addend.shape), tuple(dx.shape)))
    if bn_front is not None:
        z, link = bn_front
        tiles = _C.query("vspw_conv2d_bwd_data_bn_partials", ctypes.byref(d))
        part = torch.empty((tiles, 2, d.c), device=dy.device, dtype=torch.float32)
        with _Timed("igemm_nt_kernel", _conv_flops(d), _conv_tag(d, "dgrad")):
            _C.call("vspw_conv2d_bwd_data_bn", ctypes.byref(d), _p(dy), _p(wT), _p(addend), _p(z), _p(link.y),
                    _p(link.mean), _p(link.invstd), _p(dx), _p(part), _stream())
        link.partials, link.g = part, dx
        return dx
    if (addend is None and _strided_pw["enabled"] and kh == 1 and kw == 1 and d.stride > 1 and d.pad == 0 and d.pad_w == 0
            and d.c % 4 == 0 and d.k % 4 == 0):
        # strided pointwise conv (the stride-2 downsample): only every stride-th input pixel receives a gradient - a plain
        # GEMM on the OUTPUT pixels, then a scatter into the zero-filled input grid, instead of the data-gradient gather
        # that multiplies zeros for three quarters of its MFMAs
        dc = ConvDesc(d.n, d.oh, d.ow, d.c, d.oh, d.ow, d.k, 1, 1, 1, 0, 1, 0)
        tmp = empty_nhwc(d.n, d.c, d.oh, d.ow, dy.device)
        with _Timed("igemm_nt_kernel", _conv_flops(d), _conv_tag(d, "dgrad")):
            _C.call("vspw_conv2d_bwd_data", ctypes.byref(dc), _p(dy), _p(wT), _p(tmp), _stream())
        _C.call("vspw_strided_scatter_nhwc", _p(tmp), _p(dx), d.n, d.oh, d.ow, d.h, d.w, d.c, d.stride, _stream())
        return dx
    with _Timed("igemm_nt_kernel", _conv_flops(d), _conv_tag(d, "dgrad")):
        if addend is None:
            _C.call("vspw_conv2d_bwd_data", ctypes.byref(d), _p(dy), _p(wT), _p(dx), _stream())
        else:
            _C.call("vspw_conv2d_bwd_data_acc", ctypes.byref(d), _p(dy), _p(wT), _p(addend), _p(dx), _stream())
    return dx


# Weight gradients are leaves of the backward pass: nothing downstream of a convolution's dW is needed before the
# optimizer step (or the bucket all-reduce), while dX is on the critical path.  They are issued on a second HIP stream
# (fork after dY is ready; joined by an autograd end-of-backward callback, and before any bucket all-reduce) so that
# the split-K weight-gradient GEMM of layer i overlaps the BatchNorm-backward passes and the data-gradient GEMM of
# layer i-1 and fills their launch tails; under a captured hipGraph the fork/join become graph edges (no host events).
# Measured on the bench step: 116.1 -> 114.4 ms, bit-identical results.  VSPW_WGRAD_STREAM=0 disables it.
_wgrad_side = {"enabled": os.environ.get("VSPW_WGRAD_STREAM", "1") == "1", "stream": None, "keep": [], "dirty": False}


def set_wgrad_side_stream(enabled):
    join_side_streams()
    _wgrad_side["enabled"] = bool(enabled)


def join_side_streams():
    """Make the current stream wait for every weight-gradient GEMM issued on the side stream (call before anything
    reads parameter gradients: optimizer step, gradient all-reduce, gradient inspection)."""
    if _wgrad_side["dirty"]:
        torch.cuda.current_stream().wait_stream(_wgrad_side["stream"])
        _wgrad_side["keep"].clear()
        _wgrad_side["dirty"] = False


def _wino_wgrad(dy, x, d, dw, v=None):
    """dW of a stride-1 3x3 convolution in the Winograd domain (see winograd.hip): 4/9 of the direct multiplications.
    v: the input transform kept by the forward pass (recomputed from x when absent)."""
    dev, st = dy.device, _stream()
    fm = _wino_f3(d)
    if v is not None and v.dim() == 3 and int(v.shape[0]) in (25, 36, 49) and fm:
        fm = {25: 3, 36: 4, 49: 5}[int(v.shape[0])]  # the tile the forward pass took (it may have carried a hint)
    if fm:
        P, api = (fm + 2) * (fm + 2), "vspw_wino%d_" % fm
        T = int(_C.query(api + "tiles", ctypes.byref(d)))
        if v is None or tuple(v.shape) != (P, T, d.c):
            v = torch.empty((P, T, d.c), device=dev, dtype=torch.float32)
            _C.call(api + "input", ctypes.byref(d), _p(x), d.c, _p(v), st)
        dm = torch.empty((P, T, d.k), device=dev, dtype=torch.float32)
        _C.call(api + "dy", ctypes.byref(d), _p(dy), d.k, _p(dm), st)
        du = torch.empty((P, d.k, d.c), device=dev, dtype=torch.float32)
        nbytes = _C.query("vspw_bmm_tn_workspace", P, T, d.k, d.c)
        ws = _ws(nbytes, dev) if nbytes else None
        with _Timed("igemm_tn_kernel", 2.0 * P * T * d.k * d.c, _conv_tag(d, "wgrad-wino%d" % fm), _conv_flops(d)):
            _C.call("vspw_bmm_tn", _p(dm), _p(v), _p(du), P, T, d.k, d.c, _p(ws), nbytes, st)
        _C.call(api + "dw", _p(du), _p(dw), d.k, d.c, st)
        _wino["launches"] += 1
        _wino["f%d_launches" % fm] += 1
        return
    T = int(_C.query("vspw_wino_tiles", ctypes.byref(d)))
    if v is None or tuple(v.shape) != (16, T, d.c):
        v = torch.empty((16, T, d.c), device=dev, dtype=torch.float32)
        _C.call("vspw_wino_input", ctypes.byref(d), _p(x), d.c, _p(v), st)
    dm = torch.empty((16, T, d.k), device=dev, dtype=torch.float32)
    _C.call("vspw_wino_dy", ctypes.byref(d), _p(dy), d.k, _p(dm), st)
    du = torch.empty((16, d.k, d.c), device=dev, dtype=torch.float32)
    nbytes = _C.query("vspw_bmm_tn_workspace", 16, T, d.k, d.c)
    ws = _ws(nbytes, dev) if nbytes else None
    with _Timed("igemm_tn_kernel", 2.0 * 16 * T * d.k * d.c, _conv_tag(d, "wgrad-wino"), _conv_flops(d)):
        _C.call("vspw_bmm_tn", _p(dm), _p(v), _p(du), 16, T, d.k, d.c, _p(ws), nbytes, st)
    _C.call("vspw_wino_dw", _p(du), _p(dw), d.k, d.c, st)
    _wino["launches"] += 1


def _wgrad_launch(dy, x, d, aff=None, wino_v=None):
    dw = torch.empty((d.k, d.kh, d.kw, d.c), device=dy.device, dtype=torch.float32).permute(0, 3, 1, 2)
    if aff is None and _wino["wgrad"] and _wino_ok(d):
        _wino_wgrad(dy, x, d, dw, wino_v)
        return dw, None
    nbytes = _C.query("vspw_conv2d_bwd_weight_workspace", ctypes.byref(d))
    ws = _ws(nbytes, dy.device) if nbytes else None
    with _Timed("igemm_tn_kernel", _conv_flops(d), _conv_tag(d, "wgrad")):
        if aff is None:
            _C.call("vspw_conv2d_bwd_weight", ctypes.byref(d), _p(dy), _p(x), _p(dw), _p(ws), nbytes, _stream())
        else:
            _C.call("vspw_conv2d_bwd_weight_aff", ctypes.byref(d), _p(dy), _p(aff[0]), _p(aff[1]), _p(x), _p(dw), _p(ws),
                    nbytes, _stream())
    return dw, ws


def conv2d_backward_weight(dy, x, d, aff=None, wino_v=None):
    """aff = (y, coef): see conv2d_backward_data.  wino_v: see _wino_wgrad."""
    if not _wgrad_side["enabled"] or _ktimer["on"]:
        return _wgrad_launch(dy, x, d, aff, wino_v)[0]
    main = torch.cuda.current_stream()
    side = _wgrad_side["stream"]
    if side is None:
        side = _wgrad_side["stream"] = torch.cuda.Stream(device=dy.device)
    side.wait_stream(main)  # fork: dY (and X) are complete on the main stream
    with torch.cuda.stream(side):
        dw, ws = _wgrad_launch(dy, x, d, aff, wino_v)
    # dY / X / the workspace were allocated on the main stream's pool: keep them alive until the join so that the
    # allocator cannot hand their memory to a later main-stream kernel while the side-stream GEMM still reads it
    # (dW itself must NOT be referenced here: with a second owner autograd's AccumulateGrad would clone it - a copy on
    # the main stream that races with the side-stream GEMM - instead of adopting the tensor as p.grad)
    if torch.cuda.is_current_stream_capturing():
        _wgrad_side["keep"].append((dy, x, ws, aff, wino_v))  # graph-private pool: nothing is recycled before the join anyway
    else:
        # eager: tell the caching allocator that the side stream uses these blocks - each is recycled as soon as ITS
        # GEMM has finished, so saved activations and dY tensors are released progressively during backward (a list
        # held until the join kept the sum of all dY tensors + split-K workspaces of a backward pass alive)
        for t in (dy, x, ws, wino_v) + (tuple(aff) if aff is not None else ()):
            if t is not None:
                t.record_stream(side)
    if not _wgrad_side["dirty"]:
        _wgrad_side["dirty"] = True
        try:  # join when this backward pass ends, so that p.grad is safe to read on the main stream afterwards
            torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
        except RuntimeError:
            pass  # not inside a backward pass (direct call from a test): the caller joins
    return dw


def colsum(a2d_rows, c, a, b=None):
    nbytes = _C.query("vspw_colsum_workspace", a2d_rows, c)
    ws = _ws(nbytes, a.device)
    out = torch.empty(c, device=a.device, dtype=torch.float32)
    _C.call("vspw_colsum_prod", _p(a), _p(b), _p(out), a2d_rows, c, _p(ws), nbytes, _stream())
    return out


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d forward/backward on the implicit-GEMM MFMA kernels (csrc/conv_igemm.hip)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil):
        x = to_nhwc(x)
        y, _, d = conv2d_forward(x, w, bias, stride, pad, dil, wgrad=ctx.needs_input_grad[1])
        ctx.d = d
        ctx.has_bias = bias is not None
        ctx.wino_v = getattr(y, "_vspw_wino_v", None)
        y._vspw_wino_v = None
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        d = ctx.d
        dy = to_nhwc(dy)
        if not is_nhwc(w):
            w = w.contiguous(memory_format=torch.channels_last)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_backward_data(dy, w, d)
        if ctx.needs_input_grad[1]:
            dw = conv2d_backward_weight(dy, x, d, wino_v=ctx.wino_v)
        ctx.wino_v = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(d.n * d.oh * d.ow, d.k, dy)
        return dx, dw, db, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, dil=1):
    return Conv2dFn.apply(x, w, bias, stride, pad, dil)
