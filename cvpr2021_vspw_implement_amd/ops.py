"""Host-side operators of the VSPW hot path: thin autograd.Functions that launch the hand-written HIP kernels of
libvspw_hip.so through its C ABI (include/vspw_hip.h).  PyTorch is used for device memory, streams and autograd
bookkeeping only; every FLOP of these operators runs in csrc/*.hip.

Tensors keep the reference's logical NCHW shapes but live in NHWC memory (torch channels_last), so `x.size()` reads
like the reference while the kernels see [pixel rows][channel columns].

There is no CPU fallback: a CPU tensor raises (SURVEY.md §8b "Errors").
"""
import ctypes
import os

import torch

from . import _C
from ._C import ConvDesc

_vp = ctypes.c_void_p
_NLL_FIXED = 1048576.0  # VSPW_NLL_FIXED


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else _vp(t.data_ptr())


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            "%s: the VSPW hot path runs only through the HIP kernels on a GPU tensor (got device %s); "
            "there is no CPU fallback" % (what, t.device)
        )
    if t.dtype != torch.float32:
        raise RuntimeError("%s: fp32 tensors only (got %s)" % (what, t.dtype))


def is_nhwc(x):
    return x.dim() == 4 and x.permute(0, 2, 3, 1).is_contiguous()


def to_nhwc(x):
    """Logical NCHW tensor -> same logical tensor in NHWC memory (no-op when already so)."""
    if is_nhwc(x):
        return x
    if x.dim() != 4:
        raise RuntimeError("expected a 4-D NCHW tensor, got %s" % (tuple(x.shape),))
    _require_gpu(x, "to_nhwc")
    n, c, h, w = x.shape
    out = empty_nhwc(n, c, h, w, x.device)
    if x.is_contiguous():
        _C.call("vspw_nchw_to_nhwc", _p(x), _p(out), n, c, h * w, _stream())
    else:  # arbitrary strides: let torch gather it (plumbing, not compute)
        out.copy_(x)
    return out


def empty_nhwc(n, c, h, w, device, dtype=torch.float32):
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def _conv_desc(x, k, kh, kw, stride, pad, dil):
    """pad: int or (pad_h, pad_w) as in nn.Conv2d(padding=...)."""
    n, c, h, w = x.shape
    ph, pw = (pad, pad) if isinstance(pad, int) else (int(pad[0]), int(pad[1]))
    oh = (h + 2 * ph - dil * (kh - 1) - 1) // stride + 1
    ow = (w + 2 * pw - dil * (kw - 1) - 1) // stride + 1
    return ConvDesc(n, h, w, c, oh, ow, k, kh, kw, stride, ph, dil, pw)


# Optional per-launch timing of the MFMA GEMM kernels (bench.py's roofline leg): HIP events are recorded on the
# stream the kernels are launched on (torch's current stream) around every igemm launch.
_ktimer = {"on": False, "records": [], "hbm": []}


def kernel_timer(enable, reset=True):
    _ktimer["on"] = bool(enable)
    _C.trace = _hbm_trace if enable else None
    if enable and reset:
        _ktimer["records"] = []
        _ktimer["hbm"] = []


def kernel_timer_reset():
    _ktimer["records"] = []
    _ktimer["hbm"] = []


# HBM-bound kernel families timed next to the GEMMs (bench.py's roofline_hbm): entry point -> ALGORITHMIC bytes of one
# launch from its arguments = every operand stream read once + every result written once (per-channel vectors ignored).
def _nn(*ptrs):
    return sum(1 for q in ptrs if q is not None)


def _wino_bytes(a, chan_idx, streams_full, m_idx=None, planes=16.0):
    d = a[0]._obj
    T = int(_C.query("vspw_wino_tiles", ctypes.byref(d)))
    c = int(a[chan_idx])
    return 4.0 * c * (planes * T + d.n * d.h * d.w * streams_full)


_HBM_BYTES = {
    # x [, residual] -> z
    "vspw_bn_apply": lambda a: 4.0 * a[6] * a[7] * (2 + _nn(a[3])),
    # dz, z?, x -> dx [, dres]
    "vspw_bn_bwd_apply": lambda a: 4.0 * a[9] * a[10] * _nn(a[0], a[1], a[2], a[14], a[15]),
    "vspw_bn_bwd_reduce_pg": lambda a: 4.0 * a[6] * a[7] * _nn(a[0], a[1], a[2]),
    "vspw_bn_stats": lambda a: 4.0 * a[1] * a[2],
    "vspw_wino_input": lambda a: _wino_bytes(a, 2, 1),
    # y -> V, z
    "vspw_wino_input_apply": lambda a: _wino_bytes(a, 4, 2),
    "vspw_wino_dy": lambda a: _wino_bytes(a, 2, 1),
    # M -> y (+ relu_src / bn_y / addend operand streams when present)
    "vspw_wino_output": lambda a: _wino_bytes(a, 2, 1 + _nn(a[5], a[6], a[10])),
    # P (8 planes) -> y
    "vspw_wino_output_rows": lambda a: _wino_bytes(a, 3, 1 + _nn(a[6], a[7], a[11]), planes=8.0),
}


class _HbmTimed:
    def __init__(self, name, nbytes):
        self.name, self.nbytes = name, nbytes

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *a):
        self.e1.record()
        _ktimer["hbm"].append((self.name, self.nbytes, self.e0, self.e1))
        return False


class _NoTrace:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_TRACE = _NoTrace()


def _hbm_trace(name, args):
    f = _HBM_BYTES.get(name)
    if f is None or torch.cuda.is_current_stream_capturing():
        return _NO_TRACE
    return _HbmTimed(name, float(f(args)))


def hbm_timer_records():
    """[(entry point, algorithmic bytes, ms)] of every timed launch of the HBM-bound families (synchronises)."""
    torch.cuda.synchronize()
    return [(n, b, e0.elapsed_time(e1)) for n, b, e0, e1 in _ktimer.get("hbm", [])]


def kernel_timer_records():
    """[(kernel name, flops the launch executes, ms, tag, direct-convolution-equivalent flops)] for every timed launch
    (synchronises).  The last two differ for the Winograd GEMMs only (4/9 of the direct multiplications)."""
    torch.cuda.synchronize()
    return [(n, f, e0.elapsed_time(e1), tag, eff) for n, f, e0, e1, tag, eff in _ktimer["records"]]


class _Timed:
    def __init__(self, name, flops, tag=None, eff=None):
        self.name, self.flops, self.tag, self.eff = name, flops, tag, (flops if eff is None else eff)

    def __enter__(self):
        if _ktimer["on"]:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if _ktimer["on"]:
            self.e1.record()
            _ktimer["records"].append((self.name, self.flops, self.e0, self.e1, self.tag, self.eff))
        return False


def _conv_flops(d):
    return 2.0 * d.n * d.oh * d.ow * d.k * d.kh * d.kw * d.c


def _conv_tag(d, what):
    return "%s n%d %dx%d c%d->k%d %dx%d s%d d%d" % (what, d.n, d.h, d.w, d.c, d.k, d.kh, d.kw, d.stride, d.dil)


def _ws(nbytes, device):
    return torch.empty((max(int(nbytes), 8) + 7) // 8, device=device, dtype=torch.float64)


# --------------------------------------------------------------------------------------------------- conv
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
         "rows": os.environ.get("VSPW_WINO_ROWS", "1") == "1"}


def set_winograd(enabled):
    _wino["enabled"] = bool(enabled)


def _wino_ok(d):
    return (_wino["enabled"] and d.kh == 3 and d.kw == 3 and d.stride == 1 and min(d.c, d.k) >= _wino["min_c"]
            and _C.query("vspw_wino_supported", ctypes.byref(d)) == 1)


def _wino_conv(d, src, w, rows, reduce_c, data_gradient, bias, dst, front=None, part=None, what="fwd", u=None,
               addend=None, act=0, fuse=None, pending=None):
    """dst = conv(src) through U, V, M (see winograd.hip); rows = output channels, reduce_c = channels of src.
    u: transformed weights supplied by the caller (inference: of the BatchNorm-folded weights).
    pending = (y_prev, scale_shift): src has not been written - the input transform evaluates it (see _fwd_apply)."""
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
            part = torch.empty((_C.query("vspw_wino_stat_partials", ctypes.byref(d)), 2, k), device=x.device,
                               dtype=torch.float32)
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


def _wino_weights(w, data_gradient):
    """U [16][Cout][Cin] (forward) or U' [16][Cin][Cout] (data gradient) of a 3x3 weight, from the per-step cache."""
    return _wu_copies.get(w)[1 if data_gradient else 0]


def drop_weight_transpose_cache():
    _wt_copies.clear()
    _wu_copies.clear()


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
            part = torch.empty((_C.query("vspw_wino_stat_partials", ctypes.byref(d)), 2, d.c), device=dy.device,
                               dtype=torch.float32)
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


# --------------------------------------------------------------------------------------------------- batch norm
_sync_group = {"enabled": False, "group": None, "force": False, "clamp_var": False, "timer": None, "exchange": None}
# populations up to this many rows take their statistics two-pass in fp64 from the activations (see bn.hip:
# bn_small_finalize_kernel) instead of from the convolution epilogue's fp32 tile partials
_BN_SMALL_ROWS = 1024


def set_sync_bn(enabled, group=None, force=False, clamp_var=False, exchange=None):
    """Enable the cross-rank exchange of BatchNorm statistics (SynchronizedBatchNorm semantics,
    models/sync_batchnorm/batchnorm.py:110-150) over torch.distributed (RCCL on ROCm).
    `force` issues the collectives even in a 1-rank group (exercises the RCCL path on a single-GPU box).
    clamp_var: invstd = clamp(var, eps)^-1/2 on the exchanged statistics - bit-for-bit the formula of the reference's
    multi-device path (batchnorm.py:150); default False = (var + eps)^-1/2 everywhere, i.e. a multi-rank run computes
    what ONE device would compute on the full batch (F.batch_norm; the numerics the oracle and the fixtures pin)."""
    _sync_group["enabled"] = bool(enabled)
    _sync_group["group"] = group
    _sync_group["force"] = bool(force)
    _sync_group["clamp_var"] = bool(clamp_var)
    # peer_exchange.PeerExchange (hipIpc arenas + one small kernel per exchange) or None = torch.distributed all-reduce
    _sync_group["exchange"] = exchange if enabled else None


def sync_bn_timer(store):
    """store = list: every statistics exchange appends (event before, event after) recorded on the launch stream
    (bench.py's multi-GPU diagnostics); None switches it off."""
    _sync_group["timer"] = store


def _sync_world():
    """Number of ranks whose statistics are combined; 0 means 'one rank, but run the collectives anyway'."""
    if not _sync_group["enabled"]:
        return 1
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 1
    w = dist.get_world_size(_sync_group["group"])
    return 0 if (w == 1 and _sync_group["force"]) else w


def _all_reduce_sums(sums):
    import torch.distributed as dist

    tm = _sync_group["timer"]
    if tm is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    xc = _sync_group["exchange"]
    if xc is not None and xc.usable(sums):
        xc.all_reduce(sums)
    else:
        from . import distributed as vdist  # (imports this module: resolved at call time)

        vdist.all_reduce(sums, op=dist.ReduceOp.SUM, group=_sync_group["group"])
    if tm is not None:
        e1.record()
        tm.append((e0, e1))


def _sync_finalize(sums, world, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                   c, st):
    """sums [2][c] (this rank's) -> cross-rank totals (world != 1) -> mean / invstd / scale / shift + running statistics.
    With the peer exchange the all-reduce and the finalisation are ONE launch (vspw_xchg_bn_finalize)."""
    xc = _sync_group["exchange"] if world != 1 else None
    if xc is not None and xc.usable(sums):
        tm = _sync_group["timer"]
        if tm is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        xc.bn_finalize(sums, c, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                       _sync_group["clamp_var"])
        if tm is not None:
            e1.record()
            tm.append((e0, e1))
        return
    if world != 1:
        _all_reduce_sums(sums)
    _C.call(_finalize_name() if world != 1 else "vspw_bn_finalize", _p(sums), ctypes.c_double(count), _p(gamma), _p(beta),
            _p(running_mean), _p(running_var), momentum, eps, _p(mean), _p(invstd), _p(scale), _p(shift), c, st)


def _finalize_name():
    return "vspw_bn_finalize_clamped" if _sync_group["clamp_var"] else "vspw_bn_finalize"


# Decision tap (parity tests): ReLU and max-pool are the only non-smooth steps of the path.  When a list is installed
# with record_decisions(), every training-path node that takes such a decision appends (kind, key tensor, output):
# ("relu", the BatchNorm weight of the node, z) - z > 0 is the mask, read AFTER the forward pass has completed (a
# deferred z is written by its consumer) - or ("maxpool", None, tap indices uint8 [n, oh, ow, c], ky*3+kx).  The test
# injects them into the float64 oracle so that both differentiate the same branch of the network.
_decisions = {"store": None}  # (a holder: the max-pool node of ops.pool reads it too)


def record_decisions(store):
    _decisions["store"] = store


class BatchNormActFn(torch.autograd.Function):
    """z = [relu](BN(x) [+ residual]) [* dropout2d mask]; training or eval statistics (csrc/bn.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, mask, training, momentum, eps, relu,
                stat_part):
        _require_gpu(x, "batch_norm")
        x = to_nhwc(x)
        n, c, h, w = x.shape
        rows = n * h * w
        dev = x.device
        st = _stream()
        coef = torch.empty((4, c), device=dev, dtype=torch.float32)  # mean, invstd, scale, shift
        mean, invstd, scale, shift = coef[0], coef[1], coef[2], coef[3]
        count = float(rows)
        world = 1
        if training:
            if rows * max(_sync_world(), 1) <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                                 % (tuple(x.shape),))
            _infer_fold["gen"] += 1  # running statistics are about to be rewritten in place
            world = _sync_world()
            small = rows <= _BN_SMALL_ROWS
            if small and world == 1:
                _C.call("vspw_bn_small_finalize", _p(x), rows, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                        momentum, eps, _p(mean), _p(invstd), _p(scale), _p(shift), None, c, st)
            else:
                sums = torch.empty((2, c), device=dev, dtype=torch.float64)
                if small:
                    _C.call("vspw_bn_small_finalize", _p(x), rows, None, None, None, None, momentum, eps, None, None,
                            None, None, _p(sums), c, st)
                elif stat_part is not None:
                    _C.call("vspw_bn_reduce_partials_f32", _p(stat_part), stat_part.shape[0], c, _p(sums), st)
                else:
                    nbytes = _C.query("vspw_bn_stats_workspace", rows, c)
                    ws = _ws(nbytes, dev)
                    _C.call("vspw_bn_stats", _p(x), rows, c, _p(sums), _p(ws), nbytes, st)
                if world != 1:
                    count = float(rows * max(world, 1))
                _sync_finalize(sums, world, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd,
                               scale, shift, c, st)
        else:
            _C.call("vspw_bn_eval_coeffs", _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(mean),
                    _p(invstd), _p(scale), _p(shift), c, st)
        if residual is not None:
            residual = to_nhwc(residual)
        z = empty_nhwc(n, c, h, w, dev)
        _C.call("vspw_bn_apply", _p(x), _p(scale), _p(shift), _p(residual), _p(mask), _p(z), rows, c, h * w,
                1 if relu else 0, st)
        ctx.training = training
        ctx.relu = relu
        ctx.count = count
        ctx.world = world
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, z if relu else None, gamma, coef, mask)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, z, gamma, coef, mask = ctx.saved_tensors
        dz = to_nhwc(dz)
        n, c, h, w = x.shape
        rows = n * h * w
        dev = x.device
        st = _stream()
        mean, invstd = coef[0], coef[1]
        sums = torch.empty((2, c), device=dev, dtype=torch.float64)
        nbytes = _C.query("vspw_bn_bwd_workspace", rows, c)
        ws = _ws(nbytes, dev)
        relu = 1 if ctx.relu else 0
        dgamma = torch.empty(c, device=dev, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        dbeta = torch.empty(c, device=dev, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        # reduction + the LOCAL parameter gradients (taken before any cross-rank exchange)
        _C.call("vspw_bn_bwd_reduce_pg", _p(dz), _p(z), _p(x), _p(mean), _p(invstd), _p(mask), rows, c, h * w, relu,
                _p(sums), _p(dgamma), _p(dbeta), _p(ws), nbytes, st)
        if ctx.training and ctx.world != 1:
            _all_reduce_sums(sums)
        dx = empty_nhwc(n, c, h, w, dev) if ctx.needs_input_grad[0] else None
        dres = empty_nhwc(n, c, h, w, dev) if (ctx.has_res and ctx.needs_input_grad[5]) else None
        if dx is not None or dres is not None:
            _C.call("vspw_bn_bwd_apply", _p(dz), _p(z), _p(x), _p(mean), _p(invstd), _p(gamma), _p(sums),
                    ctypes.c_double(ctx.count), _p(mask), rows, c, h * w, relu, 1 if ctx.training else 0, _p(dx),
                    _p(dres), None, None, st)
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, residual=None, mask=None, training=True, momentum=0.1,
                   eps=1e-5, relu=False, stat_part=None):
    z = BatchNormActFn.apply(x, gamma, beta, running_mean, running_var, residual, mask, training, momentum, eps,
                             relu, stat_part)
    if _decisions["store"] is not None and relu:
        _decisions["store"].append(("relu", gamma, z))
    return z


_infer_fold = {"enabled": os.environ.get("VSPW_NO_INFER_FOLD", "0") != "1", "cache": {}, "gen": 0}


def set_inference_folding(enabled):
    _infer_fold["enabled"] = bool(enabled)


def invalidate_inference_cache():
    """Parameters / running statistics were rewritten through raw pointers (the fused SGD step, a training-mode
    BatchNorm finalize): tensor._version does not see those writes, so the folded conv+BN weights cached for
    inference are keyed on this generation counter as well."""
    _infer_fold["gen"] += 1
    _wt_cache["gen"] += 1  # the transposed copies used by the data-gradient GEMMs are stale too
    if len(_infer_fold["cache"]) > 4096:
        _infer_fold["cache"].clear()


def _conv_bn_folded(x, w, cbias, gamma, beta, running_mean, running_var, residual, stride, pad, dil, eps, relu):
    """relu?(conv(x, w*scale) + (cbias*scale + shift) [+ residual]) with scale/shift from the running statistics.  The
    folded weights are cached per weight tensor and rebuilt when any of the tensors they derive from changes."""
    if not is_nhwc(w):
        w = w.contiguous(memory_format=torch.channels_last)
    k, c, kh, kw = w.shape
    if c != x.shape[1]:
        raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (x.shape[1], c))
    srcs = (w, cbias, gamma, beta, running_mean, running_var)
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in srcs) + (float(eps), _infer_fold["gen"])
    ent = _infer_fold["cache"].get(id(w))
    st = _stream()
    if ent is None or ent[0] != key:
        coef = torch.empty((4, k), device=x.device, dtype=torch.float32)
        _C.call("vspw_bn_eval_coeffs", _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(coef[0]),
                _p(coef[1]), _p(coef[2]), _p(coef[3]), k, st)
        wf = torch.empty((k, kh, kw, c), device=x.device, dtype=torch.float32)
        bf = torch.empty(k, device=x.device, dtype=torch.float32)
        _C.call("vspw_bn_fold_weights", _p(w), _p(cbias), _p(coef[2]), _p(coef[3]), _p(wf), _p(bf), k, kh * kw * c, st)
        ent = [key, wf, bf, None]
        _infer_fold["cache"][id(w)] = ent
    _, wf, bf = ent[0], ent[1], ent[2]
    d = _conv_desc(x, k, kh, kw, stride, pad, dil)
    z = empty_nhwc(d.n, k, d.oh, d.ow, x.device)
    if residual is not None:
        residual = to_nhwc(residual)
        if tuple(residual.shape) != tuple(z.shape):
            raise RuntimeError("conv_bn_act: residual %s vs output %s" % (tuple(residual.shape), tuple(z.shape)))
    if _wino_ok(d):  # stride-1 3x3: Winograd on the folded weights (their transform is cached with them)
        if ent[3] is None:
            ent[3] = torch.empty((16, k, c), device=x.device, dtype=torch.float32)
            _C.call("vspw_wino_weights", _p(wf), _p(ent[3]), k, c, 0, st)
        _wino_conv(d, x, None, k, c, False, bf, z, what="fwd-fold", u=ent[3], addend=residual, act=1 if relu else 0,
                   fuse=_wino["fuse_dgrad"])  # no weight gradient will want V: let the GEMM stage the transform
        return z
    with _Timed("igemm_nt_kernel", _conv_flops(d), _conv_tag(d, "fwd")):
        _C.call("vspw_conv2d_fwd_ex", ctypes.byref(d), _p(x), c, _p(wf), _p(bf), _p(residual), 1 if relu else 0, _p(z),
                k, st)
    return z


class BNLink(object):
    """Side channel between a conv+BN+ReLU node (owner) and the ONE conv node that consumes its output z.  In backward
    the consumer's data-gradient GEMM already touches every element of dL/dz; given the owner's pre-BN activations and
    batch statistics it applies the owner's ReLU mask and produces the owner's two batch-norm-backward reductions in
    its epilogue (vspw_conv2d_bwd_data_bn), so the owner skips its reduction pass and the mask read.  Only valid when
    the consumer is the sole user of z - the model code asserts that by passing fuse_input=True."""

    __slots__ = ("y", "mean", "invstd", "rows", "c", "partials", "g", "pending")

    def __init__(self):
        self.y = self.mean = self.invstd = self.partials = self.g = self.pending = None
        self.rows = self.c = 0


_bn_fusion = {"enabled": os.environ.get("VSPW_NO_BN_FUSION", "0") != "1", "fused_nodes": 0,
              "affine": os.environ.get("VSPW_NO_BN_AFFINE", "0") != "1", "affine_nodes": 0,
              # narrow outputs (conv1 of a bottleneck: dy is 1/4 the size of its input gradient) gain nothing: the pass
              # saved is as cheap as the second operand stream it costs (measured: 256 ch +-0, 1024 ch -70 us / block)
              "affine_min_c": int(os.environ.get("VSPW_AFFINE_MINC", "512"))}


# Forward apply deferred into the consumer (residual blocks): a conv+BN+residual+ReLU node whose output z has exactly
# one next reader - a pointwise conv - leaves z unwritten and hands (y, scale/shift, residual) to that conv, whose GEMM
# evaluates z while staging its A operand and writes it for everyone else (vspw_conv2d_fwd_apply).  Saves the separate
# read-read-write pass of vspw_bn_apply plus the GEMM's own read of z.
_fwd_apply = {"enabled": os.environ.get("VSPW_NO_FWD_APPLY", "0") != "1", "nodes": 0,
              # ... and into the input transform of a Winograd 3x3 reader (conv1 -> conv2 of a bottleneck)
              "wino": os.environ.get("VSPW_NO_FWD_APPLY_WINO", "0") != "1", "wino_nodes": 0}


def materialize(x):
    """Write a deferred node output (see _fwd_apply) with the plain apply kernel; no-op for ordinary tensors."""
    pend = getattr(x, "_vspw_pending", None)
    if pend is not None:
        py, pss, pres = pend
        n, c, h, w = x.shape
        _C.call("vspw_bn_apply", _p(py), _p(pss[0]), _p(pss[1]), _p(pres), None, _p(x), n * h * w, c, h * w, 1,
                _stream())
        x._vspw_pending = None
    return x


def set_bn_backward_fusion(enabled):
    _bn_fusion["enabled"] = bool(enabled)


class ConvBNActFn(torch.autograd.Function):
    """conv2d -> BN(train/eval) -> [+residual] -> [ReLU] -> [Dropout2d mask] as ONE autograd node, with the
    BatchNorm statistics accumulated in the convolution epilogue (no separate pass over the conv output)."""

    @staticmethod
    def forward(ctx, x, w, cbias, gamma, beta, running_mean, running_var, residual, mask, stride, pad, dil, training,
                momentum, eps, relu, skip_out=False, in_link=None, out_link=None, pending=None, defer=False):
        _require_gpu(x, "conv_bn_act")
        x = to_nhwc(x)
        dd = _conv_desc(x, w.shape[0], w.shape[2], w.shape[3], stride, pad, dil)
        small = training and dd.n * dd.oh * dd.ow <= _BN_SMALL_ROWS  # exact two-pass statistics (vspw_bn_small_finalize)
        fuse_stats = training and not small
        y, part, d = conv2d_forward(x, w, cbias, stride, pad, dil, want_stats=fuse_stats, pending=pending,
                                    wgrad=ctx.needs_input_grad[1])
        n, c, h, wd = y.shape
        rows = n * h * wd
        dev = x.device
        st = _stream()
        coef = torch.empty((4, c), device=dev, dtype=torch.float32)
        mean, invstd, scale, shift = coef[0], coef[1], coef[2], coef[3]
        count = float(rows)
        world = 1
        if training:
            if rows * max(_sync_world(), 1) <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                                 % (tuple(y.shape),))
            _infer_fold["gen"] += 1  # running statistics are about to be rewritten in place
            world = _sync_world()
            if small and world == 1:
                _C.call("vspw_bn_small_finalize", _p(y), rows, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                        momentum, eps, _p(mean), _p(invstd), _p(scale), _p(shift), None, c, st)
            elif part is not None and world == 1:  # single rank: reduce the epilogue partials and finalise in one launch
                _C.call("vspw_bn_finalize_partials_f32", _p(part), part.shape[0], ctypes.c_double(count), _p(gamma),
                        _p(beta), _p(running_mean), _p(running_var), momentum, eps, _p(mean), _p(invstd), _p(scale),
                        _p(shift), c, st)
            else:
                sums = torch.empty((2, c), device=dev, dtype=torch.float64)
                if small:
                    _C.call("vspw_bn_small_finalize", _p(y), rows, None, None, None, None, momentum, eps, None, None,
                            None, None, _p(sums), c, st)
                elif part is not None:
                    _C.call("vspw_bn_reduce_partials_f32", _p(part), part.shape[0], c, _p(sums), st)
                else:
                    nbytes = _C.query("vspw_bn_stats_workspace", rows, c)
                    ws = _ws(nbytes, dev)
                    _C.call("vspw_bn_stats", _p(y), rows, c, _p(sums), _p(ws), nbytes, st)
                if world != 1:
                    count = float(rows * max(world, 1))
                _sync_finalize(sums, world, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd,
                               scale, shift, c, st)
        else:
            _C.call("vspw_bn_eval_coeffs", _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(mean),
                    _p(invstd), _p(scale), _p(shift), c, st)
        if residual is not None:
            residual = to_nhwc(residual)
        z = empty_nhwc(n, c, h, wd, dev)
        if defer and out_link is not None and mask is None and relu and c % 32 == 0:
            # z is written by its one reader (see _fwd_apply); everything that touches it later (this node's backward,
            # the reader's weight gradient, the next skip connection) runs after that reader on the same stream
            out_link.pending = (y, coef[2:], residual)
        else:
            _C.call("vspw_bn_apply", _p(y), _p(scale), _p(shift), _p(residual), _p(mask), _p(z), rows, c, h * wd,
                    1 if relu else 0, st)
        ctx.d = d
        ctx.wino_v = getattr(y, "_vspw_wino_v", None)
        y._vspw_wino_v = None
        ctx.training = training
        ctx.relu = relu
        ctx.count = count
        ctx.world = world
        ctx.has_res = residual is not None
        ctx.has_cbias = cbias is not None
        ctx.save_for_backward(x, w, y, z if relu else None, gamma, coef, mask)
        ctx.skip_out = bool(skip_out)
        ctx.out_link = None
        if out_link is not None and training and relu and mask is None:
            out_link.y, out_link.mean, out_link.invstd, out_link.rows, out_link.c = y, mean, invstd, rows, c
            ctx.out_link = out_link
        ctx.in_link = None
        if in_link is not None and in_link.y is not None and in_link.c == x.shape[1] and \
                in_link.rows == x.shape[0] * x.shape[2] * x.shape[3] and \
                _C.query("vspw_conv2d_bwd_data_bn_partials", ctypes.byref(d)) > 0:
            ctx.in_link = in_link
        if skip_out:
            # second output = the input itself (autograd turns it into a view with this node as grad_fn): the block's
            # skip connection is routed through here so that its gradient is added in this conv's dgrad epilogue
            return z, x
        return z

    @staticmethod
    def backward(ctx, dz, dskip=None):
        x, w, y, z, gamma, coef, mask = ctx.saved_tensors
        d = ctx.d
        dz = to_nhwc(dz)
        n, c, h, wd = y.shape
        rows = n * h * wd
        dev = y.device
        st = _stream()
        mean, invstd = coef[0], coef[1]
        relu = 1 if ctx.relu else 0
        train = 1 if ctx.training else 0
        sums = torch.empty((2, c), device=dev, dtype=torch.float64)
        dgamma = torch.empty(c, device=dev, dtype=torch.float32) if ctx.needs_input_grad[3] else None
        dbeta = torch.empty(c, device=dev, dtype=torch.float32) if ctx.needs_input_grad[4] else None
        link = ctx.out_link
        fused = link is not None and link.partials is not None and link.g is not None and \
            link.g.data_ptr() == dz.data_ptr() and tuple(link.g.shape) == tuple(dz.shape)
        dy = None
        aff = None
        if fused:
            # the consumer's data gradient already masked dz with this node's ReLU and left the two reductions behind
            _bn_fusion["fused_nodes"] += 1
            part = link.partials
            _C.call("vspw_bn_bwd_reduce_partials_f32", _p(part), part.shape[0], c, _p(sums), _p(dgamma), _p(dbeta), st)
            if ctx.training and ctx.world != 1:
                _all_reduce_sums(sums)
            pointwise = d.kh == 1 and d.kw == 1 and d.stride == 1 and d.pad == 0 and d.pad_w == 0
            if (_bn_fusion["affine"] and pointwise and not ctx.has_cbias and c >= _bn_fusion["affine_min_c"]
                    and _C.query("vspw_conv2d_bwd_aff_supported", ctypes.byref(d)) == 1):
                # pointwise conv: BatchNorm's backward apply becomes an affine map staged by the two gradient GEMMs of
                # this conv - dy (the gradient w.r.t. the conv output) is never written
                coef = torch.empty((3, c), device=dev, dtype=torch.float32)
                _C.call("vspw_bn_bwd_affine_coeffs", _p(sums), ctypes.c_double(ctx.count), _p(gamma), _p(mean),
                        _p(invstd), _p(coef), c, train, st)
                aff = (y, coef)
                dy = dz
                _bn_fusion["affine_nodes"] += 1
            else:
                dy = empty_nhwc(n, c, h, wd, dev)
                _C.call("vspw_bn_bwd_apply", _p(dz), None, _p(y), _p(mean), _p(invstd), _p(gamma), _p(sums),
                        ctypes.c_double(ctx.count), None, rows, c, h * wd, 0, train, _p(dy), None, None, None, st)
            dres = dz if (ctx.has_res and ctx.needs_input_grad[7]) else None  # dres = g, which dz already is
        else:
            nbytes = _C.query("vspw_bn_bwd_workspace", rows, c)
            ws = _ws(nbytes, dev)
            # reduction + the LOCAL parameter gradients (dgamma/dbeta are taken before any cross-rank exchange)
            _C.call("vspw_bn_bwd_reduce_pg", _p(dz), _p(z), _p(y), _p(mean), _p(invstd), _p(mask), rows, c, h * wd,
                    relu, _p(sums), _p(dgamma), _p(dbeta), _p(ws), nbytes, st)
            if ctx.training and ctx.world != 1:
                _all_reduce_sums(sums)
            dres = empty_nhwc(n, c, h, wd, dev) if (ctx.has_res and ctx.needs_input_grad[7]) else None
            dy = empty_nhwc(n, c, h, wd, dev)
            _C.call("vspw_bn_bwd_apply", _p(dz), _p(z), _p(y), _p(mean), _p(invstd), _p(gamma), _p(sums),
                    ctypes.c_double(ctx.count), _p(mask), rows, c, h * wd, relu, train, _p(dy), _p(dres), None, None,
                    st)
        if link is not None:
            link.partials = link.g = link.y = link.mean = link.invstd = None  # one backward per forward
        if not is_nhwc(w):
            w = w.contiguous(memory_format=torch.channels_last)
        dx = dw = dcb = None
        if ctx.needs_input_grad[0]:
            front = None
            if ctx.in_link is not None and ctx.in_link.y is not None and _bn_fusion["enabled"]:
                front = (x, ctx.in_link)
            dx = conv2d_backward_data(dy, w, d, addend=dskip if ctx.skip_out else None, bn_front=front, aff=aff)
        if ctx.needs_input_grad[1]:
            dw = conv2d_backward_weight(dy, x, d, aff=aff, wino_v=ctx.wino_v)
        ctx.wino_v = None
        if ctx.has_cbias and ctx.needs_input_grad[2]:
            dcb = colsum(rows, c, dy)
        return (dx, dw, dcb, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None, None, None,
                None, None, None, None)


def conv_bn_act(x, w, cbias, gamma, beta, running_mean, running_var, residual=None, mask=None, stride=1, pad=0,
                dil=1, training=True, momentum=0.1, eps=1e-5, relu=True, skip_out=False, fuse_input=False,
                defer_apply=False):
    """skip_out: also return the input as a second output (see ConvBNActFn.forward) - use THAT tensor for the skip
    connection of a residual block and its gradient is folded into this convolution's data-gradient epilogue.
    fuse_input: the caller guarantees this conv is the ONLY consumer of x; if x came out of a conv+BN+ReLU node, that
    node's batch-norm backward reductions are then produced by this conv's data gradient (see BNLink).
    defer_apply: the caller guarantees that the NEXT thing done with the output is a conv_bn_act(fuse_input=True) call
    on it (or ops.materialize): the output may come back unwritten, to be evaluated by that call (see _fwd_apply)."""
    pending = getattr(x, "_vspw_pending", None)
    if residual is not None:
        materialize(residual)
    if not training and mask is None and not torch.is_grad_enabled() and _infer_fold["enabled"]:
        # inference: BatchNorm is an affine map per output channel - fold its scale into the weights, pass its shift as
        # the bias, add the residual and apply the ReLU in the GEMM epilogue: one launch, no pass over y
        _require_gpu(x, "conv_bn_act")
        x = to_nhwc(materialize(x))
        z = _conv_bn_folded(x, w, cbias, gamma, beta, running_mean, running_var, residual, stride, pad, dil, eps, relu)
        return (z, x) if skip_out else z
    grad = torch.is_grad_enabled() and _bn_fusion["enabled"]
    if pending is not None:
        ok = fuse_input and is_nhwc(x) and w.shape[2] == 1 and w.shape[3] == 1 and stride == 1 and pad == 0
        if ok:
            ok = _C.query("vspw_conv2d_fwd_apply_supported",
                          ctypes.byref(_conv_desc(x, w.shape[0], 1, 1, stride, pad, dil))) == 1
        elif fuse_input and is_nhwc(x) and w.shape[2] == 3 and w.shape[3] == 3:
            # stride-1 3x3 on the Winograd path: its input transform evaluates the deferred apply
            dq = _conv_desc(x, w.shape[0], 3, 3, stride, pad, dil)
            ok = _wino_ok(dq) and _wino_takes_pending(dq, pending, torch.is_grad_enabled() and w.requires_grad)
            if ok:
                _fwd_apply["wino_nodes"] += 1
        if not ok:
            materialize(x)
            pending = None
    in_link = getattr(x, "_vspw_link", None) if (fuse_input and grad and x.requires_grad) else None
    out_link = BNLink() if (grad and training and relu and mask is None) else None
    defer = bool(defer_apply and _fwd_apply["enabled"] and out_link is not None)
    out = ConvBNActFn.apply(x, w, cbias, gamma, beta, running_mean, running_var, residual, mask, stride, pad, dil,
                            training, momentum, eps, relu, skip_out, in_link, out_link, pending, defer)
    if pending is not None:
        x._vspw_pending = None  # written by the GEMM just launched
        _fwd_apply["nodes"] += 1
    if _decisions["store"] is not None and relu:
        _decisions["store"].append(("relu", gamma, out[0] if skip_out else out))
    if out_link is not None and out_link.y is not None:
        z = out[0] if skip_out else out
        z._vspw_link = out_link
        if out_link.pending is not None:
            z._vspw_pending, out_link.pending = out_link.pending, None
    return out


# --------------------------------------------------------------------------------------------------- pooling
class MaxPool3x3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, "max_pool")
        x = to_nhwc(x)
        n, c, h, w = x.shape
        oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        y = empty_nhwc(n, c, oh, ow, x.device)
        idx = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.uint8)
        _C.call("vspw_maxpool3x3s2_fwd", _p(x), _p(y), _p(idx), n, h, w, c, oh, ow, _stream())
        ctx.shape = (n, c, h, w, oh, ow)
        ctx.save_for_backward(idx)
        if _decisions["store"] is not None:
            _decisions["store"].append(("maxpool", None, idx))
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w, oh, ow = ctx.shape
        dy = to_nhwc(dy)
        dx = empty_nhwc(n, c, h, w, dy.device)
        _C.call("vspw_maxpool3x3s2_bwd", _p(dy), _p(idx), _p(dx), n, h, w, c, oh, ow, _stream())
        return dx


def max_pool3x3s2(x):
    return MaxPool3x3s2Fn.apply(x)


class PyramidPoolFn(torch.autograd.Function):
    """AdaptiveAvgPool2d at every pyramid scale over all frames, then (optionally) the Temporal-Context-Blending
    mean over the T frames of each clip (models/clip_psp.py:157-188).  Returns one [B,C,s,s] tensor per scale."""

    @staticmethod
    def forward(ctx, x, scales, T, wts, tail=0):
        """tail > 0: additionally return the last `tail` frames of x (the current frames of the clips, which the heads
        read next to the blended pools, models/clip_psp.py:154-189) as a zero-copy slab; in backward their gradient is
        added into the pooled gradient in place - no zero-filled full-size tensor, no layout change, no separate
        accumulation pass over conv5's gradient."""
        _require_gpu(x, "pyramid_pool")
        x = to_nhwc(x)
        n, c, h, w = x.shape
        if n % T != 0:
            raise RuntimeError("pyramid_pool: batch %d is not a multiple of T=%d" % (n, T))
        B = n // T
        st = _stream()
        outs = []
        keep = []
        if wts is not None:
            wts = wts.contiguous()
            if tuple(wts.shape) != (B, T):
                raise RuntimeError("pyramid_pool: temporal weights must be [B, T] = [%d, %d]" % (B, T))
        pooled_all = [empty_nhwc(n, c, s, s, x.device) for s in scales]
        fused = c % 4 == 0 and 0 < len(scales) <= 4
        if fused:  # every scale from one pass over x
            svec = (ctypes.c_int * len(scales))(*scales)
            ptrs = (ctypes.c_void_p * len(scales))(*[t.data_ptr() for t in pooled_all])
            nbytes = _C.query("vspw_pyramid_pool_fwd_workspace", svec, len(scales), n, h, c)
            ws = _ws(nbytes, x.device)
            _C.call("vspw_pyramid_pool_fwd", _p(x), svec, len(scales), ptrs, n, h, w, c, _p(ws), nbytes, st)
        for s, pooled in zip(scales, pooled_all):
            if not fused:
                _C.call("vspw_adaptive_avgpool_fwd", _p(x), _p(pooled), n, h, w, c, s, st)
            if T > 1:
                blended = empty_nhwc(B, c, s, s, x.device)
                _C.call("vspw_temporal_mean_fwd", _p(pooled), _p(wts), _p(blended), T, B, s * s * c, st)
                outs.append(blended)
                if wts is not None:
                    keep.append(pooled)  # tiny ([n,c,s,s]); needed for the gradient of the temporal weights
            else:
                outs.append(pooled)
        ctx.meta = (n, c, h, w, tuple(scales), T, int(tail))
        ctx.save_for_backward(wts, *keep)
        if tail:
            outs.append(x[n - tail:])  # NHWC memory: the last frames are one contiguous slab
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        wts = ctx.saved_tensors[0]
        pooled_all = ctx.saved_tensors[1:]
        n, c, h, w, scales, T, tail = ctx.meta
        B = n // T
        st = _stream()
        gtail = None
        if tail:
            gtail, grads = grads[-1], grads[:-1]
        dev = next(g for g in list(grads) + [gtail] if g is not None).device
        dx = empty_nhwc(n, c, h, w, dev)

        def add_tail(res):
            if gtail is not None:
                gt = to_nhwc(gtail)
                dxt = res[n - tail:]
                _C.call("vspw_axpby", _p(gt), _p(dxt), gt.numel(), 1.0, 1.0, st)
            return res

        live = [(s, to_nhwc(g)) for s, g in zip(scales, grads) if g is not None]
        if not live:
            return add_tail(dx.zero_()), None, None, None, None
        if wts is None and c % 4 == 0 and len(live) <= 4:  # one fused pass: all scales + the temporal-mean adjoint
            ptrs = (ctypes.c_void_p * len(live))(*[g.data_ptr() for _, g in live])
            svec = (ctypes.c_int * len(live))(*[s for s, _ in live])
            _C.call("vspw_pyramid_pool_bwd", ptrs, svec, len(live), _p(dx), n, h, w, c, T, st)
            return add_tail(dx), None, None, None, None
        first = True
        dwts = None
        if wts is not None and ctx.needs_input_grad[3]:
            dwts = torch.empty_like(wts)
        pooled_of = dict(zip(scales, pooled_all)) if pooled_all else {}
        for s, g in live:
            if T > 1:
                gp = empty_nhwc(n, c, s, s, dev)
                _C.call("vspw_temporal_mean_bwd", _p(g), _p(wts), _p(gp), T, B, s * s * c, st)
                if dwts is not None:
                    _C.call("vspw_temporal_mean_wgrad", _p(g), _p(pooled_of[s]), _p(dwts), T, B, s * s * c,
                            0 if first else 1, st)
            else:
                gp = g
            _C.call("vspw_adaptive_avgpool_bwd", _p(gp), _p(dx), n, h, w, c, s, 0 if first else 1, st)
            first = False
        return add_tail(dx), None, None, dwts, None


def pyramid_pool(x, scales, T=1, wts=None, tail=0):
    """tail=0: tuple of pooled maps (one per scale); tail=k: (pooled maps..., x[-k:])."""
    return PyramidPoolFn.apply(x, tuple(scales), T, wts, int(tail))


class TailFramesFn(torch.autograd.Function):
    """x[n - count:] of an NHWC-memory tensor (zero-copy slab); the gradient is written straight into an NHWC buffer
    (zeros in front) instead of autograd's NCHW-ordered slice-backward tensor that every consumer would re-lay-out."""

    @staticmethod
    def forward(ctx, x, count):
        _require_gpu(x, "tail_frames")
        x = to_nhwc(x)
        ctx.meta = (tuple(x.shape), int(count))
        return x[x.shape[0] - count:]

    @staticmethod
    def backward(ctx, g):
        (n, c, h, w), count = ctx.meta
        g = to_nhwc(g)
        dx = empty_nhwc(n, c, h, w, g.device)
        st = _stream()
        head = dx[:n - count]
        if head.numel():
            head.zero_()
        _C.call("vspw_axpby", _p(g), _p(dx[n - count:]), g.numel(), 1.0, 0.0, st)
        return dx, None


def tail_frames(x, count):
    if not (torch.is_grad_enabled() and x.requires_grad):
        return to_nhwc(x)[x.shape[0] - count:]
    return TailFramesFn.apply(x, count)


class SplitBatchFn(torch.autograd.Function):
    """(x[:b], x[b:]) of an NHWC-memory tensor as zero-copy slabs (NetWarp: [current; previous] stacked on the batch,
    reference models/netwarp.py:196-203).  The gradient of both halves lands in ONE NHWC buffer by two contiguous copies;
    autograd's own slice-backward builds two zero-filled full-size tensors, fills them through strided element-wise
    kernels and adds them (measured on the NetWarp step: 21 + 8 launches, 1.05 ms)."""

    @staticmethod
    def forward(ctx, x, b):
        _require_gpu(x, "split_batch")
        x = to_nhwc(x)
        ctx.meta = (tuple(x.shape), int(b))
        return x[:b], x[b:]

    @staticmethod
    def backward(ctx, ga, gb):
        (n, c, h, w), b = ctx.meta
        dx = empty_nhwc(n, c, h, w, (ga if ga is not None else gb).device)
        st = _stream()
        for part, g in ((dx[:b], ga), (dx[b:], gb)):
            if g is None:
                part.zero_()
            else:
                g = to_nhwc(g)
                _C.call("vspw_axpby", _p(g), _p(part), g.numel(), 1.0, 0.0, st)
        return dx, None


def split_batch(x, b):
    if not (torch.is_grad_enabled() and x.requires_grad):
        x = to_nhwc(x)
        return x[:b], x[b:]
    return SplitBatchFn.apply(x, b)


class PPMConcatFn(torch.autograd.Function):
    """torch.cat([conv5] + [bilinear_up(branch_i)], dim=1) written straight into one NHWC buffer
    (models/clip_psp.py:45-53)."""

    @staticmethod
    def forward(ctx, x, *branches):
        _require_gpu(x, "ppm_concat")
        x = to_nhwc(x)
        n, c, h, w = x.shape
        st = _stream()
        ctot = c + sum(b.shape[1] for b in branches)
        out = empty_nhwc(n, ctot, h, w, x.device)
        _C.call("vspw_copy_channels", _p(x), _p(out), n * h * w, c, c, 0, ctot, 0, st)
        off = c
        metas = []
        for b in branches:
            b = to_nhwc(b)
            bn_, bc, bh, bw = b.shape
            if bn_ != n:
                raise RuntimeError("ppm_concat: batch mismatch")
            _C.call("vspw_bilinear_fwd", _p(b), _p(out), n, bh, bw, h, w, bc, bc, 0, ctot, off, st)
            metas.append((bc, bh, bw, off))
            off += bc
        ctx.meta = (n, c, h, w, ctot, metas)
        return out

    @staticmethod
    def backward(ctx, g):
        n, c, h, w, ctot, metas = ctx.meta
        g = to_nhwc(g)
        st = _stream()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(n, c, h, w, g.device)
            _C.call("vspw_copy_channels", _p(g), _p(dx), n * h * w, c, ctot, 0, c, 0, st)
        outs = [dx]
        for i, (bc, bh, bw, off) in enumerate(metas):
            if not ctx.needs_input_grad[1 + i]:
                outs.append(None)
                continue
            db = empty_nhwc(n, bc, bh, bw, g.device)
            _C.call("vspw_bilinear_bwd", _p(g), _p(db), n, bh, bw, h, w, bc, bc, 0, ctot, off, st)
            outs.append(db)
        return tuple(outs)


def ppm_concat(x, branches):
    return PPMConcatFn.apply(x, *branches)


class BilinearFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False)."""

    @staticmethod
    def forward(ctx, x, size):
        _require_gpu(x, "interpolate")
        x = to_nhwc(x)
        n, c, h, w = x.shape
        oh, ow = int(size[0]), int(size[1])
        y = empty_nhwc(n, c, oh, ow, x.device)
        _C.call("vspw_bilinear_fwd", _p(x), _p(y), n, h, w, oh, ow, c, c, 0, c, 0, _stream())
        ctx.meta = (n, c, h, w, oh, ow)
        return y

    @staticmethod
    def backward(ctx, g):
        n, c, h, w, oh, ow = ctx.meta
        g = to_nhwc(g)
        dx = empty_nhwc(n, c, h, w, g.device)
        _C.call("vspw_bilinear_bwd", _p(g), _p(dx), n, h, w, oh, ow, c, c, 0, c, 0, _stream())
        return dx, None


def interpolate_bilinear(x, size):
    return BilinearFn.apply(x, tuple(size))


class AvgPool2x2Fn(torch.autograd.Function):
    """F.avg_pool2d(x, (2, 2)): the non-local decoders' `downsample` switch (models/non_local_models.py:30-32,136-137)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, "avg_pool2x2")
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h // 2, w // 2, x.device)
        _C.call("vspw_avgpool2x2_nhwc_fwd", _p(x), _p(y), n, h, w, c, _stream())
        ctx.meta = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = ctx.meta
        g = to_nhwc(g)
        dx = empty_nhwc(n, c, h, w, g.device)
        _C.call("vspw_avgpool2x2_nhwc_bwd", _p(g), _p(dx), n, h, w, c, _stream())
        return dx


def avg_pool2x2(x):
    return AvgPool2x2Fn.apply(x)


# --------------------------------------------------------------------------------------------------- softmax / loss
class ChannelSoftmaxFn(torch.autograd.Function):
    """(log_)softmax over dim=1 of an NCHW-logical / NHWC-memory tensor."""

    @staticmethod
    def forward(ctx, x, log):
        _require_gpu(x, "softmax")
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h, w, x.device)
        _C.call("vspw_softmax_lastdim_fwd", _p(x), _p(y), n * h * w, c, 1.0, 1 if log else 0, _stream())
        ctx.log = log
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = to_nhwc(g)
        n, c, h, w = y.shape
        dx = empty_nhwc(n, c, h, w, y.device)
        _C.call("vspw_softmax_lastdim_bwd", _p(g), _p(y), _p(dx), n * h * w, c, 1.0, 1 if ctx.log else 0, _stream())
        return dx, None


def log_softmax_channels(x):
    return ChannelSoftmaxFn.apply(x, True)


def softmax_channels(x):
    return ChannelSoftmaxFn.apply(x, False)


def _labels(label, n, H, W):
    """[n,1,H,W] or [n,H,W] labels -> (contiguous [n,H,W] tensor, is_float32).  fp32 labels - what the drivers hand
    over - stay fp32: the loss kernels apply the reference's label.squeeze(1).long() themselves (no cast pass over
    the label planes); anything else is converted to int64."""
    if label.dim() == 4:
        label = label.squeeze(1)
    if tuple(label.shape) != (n, H, W):
        raise RuntimeError("label shape %s does not match (%d,%d,%d)" % (tuple(label.shape), n, H, W))
    if label.dtype == torch.float32:
        return label.contiguous(), 1
    return label.long().contiguous(), 0


class SegNLLFn(torch.autograd.Function):
    """loss, acc = NLLLoss(ignore)(bilinear_up(logp), label), pixel_acc(...)  without materialising the up-sampled
    tensor.  `from_logits`: logp = log_softmax(x) is computed here and the backward returns d/d logits."""

    @staticmethod
    def forward(ctx, x, label, ignore_index, want_acc, from_logits):
        _require_gpu(x, "seg_nll")
        x = to_nhwc(x)
        n, k, h, w = x.shape
        H, W = label.shape[-2], label.shape[-1]
        lab, lab_f32 = _labels(label, n, H, W)
        st = _stream()
        if from_logits:
            logp = empty_nhwc(n, k, h, w, x.device)
            _C.call("vspw_softmax_lastdim_fwd", _p(x), _p(logp), n * h * w, k, 1.0, 1, st)
        else:
            logp = x
        out = torch.empty(4, device=x.device, dtype=torch.float64)
        _C.call("vspw_zero_f64", _p(out), 4, st)
        _C.call("vspw_seg_nll_fwd", _p(logp), _p(lab), lab_f32, _p(out), n, h, w, k, H, W, int(ignore_index),
                1 if want_acc else 0, st)
        loss = (out[0] * (1.0 / _NLL_FIXED) / out[1]).float()  # out[0] is fixed-point (include/vspw_hip.h)
        acc = (out[2] / (out[3] + 1e-10)).float()
        ctx.meta = (n, k, h, w, H, W, int(ignore_index), bool(from_logits), lab_f32)
        ctx.save_for_backward(logp, lab, out)
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, gloss, gacc):
        logp, lab, out = ctx.saved_tensors
        n, k, h, w, H, W, ignore, from_logits, lab_f32 = ctx.meta
        g = gloss.reshape(1).float().contiguous()
        dx = empty_nhwc(n, k, h, w, logp.device)
        _C.call("vspw_seg_nll_bwd", _p(logp), _p(lab), lab_f32, _p(out), _p(g), _p(dx), n, h, w, k, H, W, ignore,
                1 if from_logits else 0, _stream())
        return dx, None, None, None, None


def seg_nll(x, label, ignore_index=255, want_acc=True, from_logits=True):
    return SegNLLFn.apply(x, label, ignore_index, want_acc, from_logits)


def upsample_softmax(logits, size):
    """Inference head: softmax(bilinear_up(logits, size), dim=1) -> [n,K,H,W] (NHWC memory). No autograd."""
    _require_gpu(logits, "upsample_softmax")
    logits = to_nhwc(logits.detach())
    n, k, h, w = logits.shape
    H, W = int(size[0]), int(size[1])
    probs = empty_nhwc(n, k, H, W, logits.device)
    _C.call("vspw_upsample_softmax", _p(logits), _p(probs), n, h, w, k, H, W, _stream())
    return probs


# --------------------------------------------------------------------------------------------------- batched GEMMs
def _transpose(a):
    """[B,R,C] -> [B,C,R] contiguous."""
    B, R, C = a.shape
    out = torch.empty((B, C, R), device=a.device, dtype=torch.float32)
    _C.call("vspw_transpose_batched", _p(a), _p(out), B, R, C, _stream())
    return out


def _bmm_nt(a, bt):
    """a [B,M,K], bt [B,N,K] (contiguous) -> [B,M,N]; the batch is a grid dimension of the pointwise-conv GEMM kernel."""
    B, M, K = a.shape
    N = bt.shape[1]
    y = torch.empty((B, M, N), device=a.device, dtype=torch.float32)
    _C.call("vspw_bmm_nt", _p(a), _p(bt), _p(y), B, M, N, K, _stream())
    return y


def _bmm_tn(a, b):
    """a [B,R,M], b [B,R,N] (contiguous) -> a^T b [B,M,N]; batched split-R GEMM (the weight-gradient kernel)."""
    B, R, M = a.shape
    N = b.shape[2]
    y = torch.empty((B, M, N), device=a.device, dtype=torch.float32)
    nbytes = _C.query("vspw_bmm_tn_workspace", B, R, M, N)
    ws = _ws(nbytes, a.device) if nbytes else None
    _C.call("vspw_bmm_tn", _p(a), _p(b), _p(y), B, R, M, N, _p(ws), nbytes, _stream())
    return y


class BmmNTFn(torch.autograd.Function):
    """c[b] = a[b] @ bt[b]^T ; a [B,M,K], bt [B,N,K] (both contiguous, K fastest)."""

    @staticmethod
    def forward(ctx, a, bt):
        _require_gpu(a, "bmm_nt")
        a = a.contiguous()
        bt = bt.contiguous()
        ctx.save_for_backward(a, bt)
        return _bmm_nt(a, bt)

    @staticmethod
    def backward(ctx, g):
        a, bt = ctx.saved_tensors
        g = g.contiguous()
        da = dbt = None
        if ctx.needs_input_grad[0]:  # da = g @ bt : NT with (bt^T) [K,N] rows
            da = _bmm_nt(g, _transpose(bt))
        if ctx.needs_input_grad[1]:  # dbt = g^T @ a  [N,K]
            dbt = _bmm_tn(g, a)
        return da, dbt


class BmmTNFn(torch.autograd.Function):
    """c[b] = a[b]^T @ b[b] ; a [B,R,M], b [B,R,N] -> [B,M,N]."""

    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a, "bmm_tn")
        a = a.contiguous()
        b = b.contiguous()
        ctx.save_for_backward(a, b)
        return _bmm_tn(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()  # [B,M,N]
        da = db = None
        if ctx.needs_input_grad[0]:  # da [R,M] = b [R,N] @ g[M,N]^T
            da = _bmm_nt(b, g)
        if ctx.needs_input_grad[1]:  # db [R,N] = a [R,M] @ g [M,N] = NT(a, g^T [N,M])
            db = _bmm_nt(a, _transpose(g))
        return da, db


def bmm_nt(a, bt):
    return BmmNTFn.apply(a, bt)


def bmm_tn(a, b):
    return BmmTNFn.apply(a, b)


def _nl_dot(q, k, v, scale):
    """scale * (q k^T) v for contiguous [B,N,C] operands, without the N x N intermediate (csrc/nonlocal.hip)."""
    B, N, C = q.shape
    out = torch.empty((B, N, C), device=q.device, dtype=torch.float32)
    nbytes = _C.query("vspw_nl_dot_workspace", B, N, C)
    ws = _ws(nbytes, q.device) if nbytes else None
    with _Timed("nl_dot_kernel", 4.0 * B * N * N * C, "nl_dot b%d n%d c%d" % (B, N, C)):
        _C.call("vspw_nl_dot", _p(q), _p(k), _p(v), _p(out), B, N, C, float(scale), _p(ws), nbytes, _stream())
    return out


def nl_dot_supported(c):
    return c in (32, 64, 128)


class NonLocalDotFn(torch.autograd.Function):
    """y = (theta phi^T / N) g of NLBlockND mode 'dot' (reference models/non_local.py:116-133), theta/phi/g [B,N,C]:
    the affinity f = theta phi^T is streamed through registers tile by tile and never written to memory; the three
    gradients are the same kernel with the operands permuted (f is linear in everything: no softmax)."""

    @staticmethod
    def forward(ctx, theta, phi, g, scale):
        _require_gpu(theta, "non_local_dot")
        theta, phi, g = theta.contiguous(), phi.contiguous(), g.contiguous()
        if theta.shape != phi.shape or theta.shape != g.shape or theta.dim() != 3:
            raise RuntimeError("non_local_dot: theta, phi, g must be [B,N,C] of equal shape")
        ctx.scale = float(scale)
        ctx.save_for_backward(theta, phi, g)
        return _nl_dot(theta, phi, g, scale)

    @staticmethod
    def backward(ctx, dy):
        theta, phi, g = ctx.saved_tensors
        dy = dy.contiguous()
        s = ctx.scale
        dth = _nl_dot(dy, g, phi, s) if ctx.needs_input_grad[0] else None   # (dy g^T) phi
        dph = _nl_dot(g, dy, theta, s) if ctx.needs_input_grad[1] else None  # (g dy^T) theta
        dg = _nl_dot(phi, theta, dy, s) if ctx.needs_input_grad[2] else None  # (phi theta^T) dy
        return dth, dph, dg, None


def non_local_dot(theta, phi, g, scale):
    return NonLocalDotFn.apply(theta, phi, g, scale)


class TransposeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        _require_gpu(a, "transpose")
        return _transpose(a.contiguous())

    @staticmethod
    def backward(ctx, g):
        return _transpose(g.contiguous())


def transpose_last2(a):
    return TransposeFn.apply(a)


class RowSoftmaxFn(torch.autograd.Function):
    """softmax(alpha * x, dim=-1) for a contiguous [..., K] tensor."""

    @staticmethod
    def forward(ctx, x, alpha):
        _require_gpu(x, "row_softmax")
        x = x.contiguous()
        k = x.shape[-1]
        rows = x.numel() // k
        y = torch.empty_like(x)
        _C.call("vspw_softmax_lastdim_fwd", _p(x), _p(y), rows, k, float(alpha), 0, _stream())
        ctx.alpha = float(alpha)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        k = y.shape[-1]
        dx = torch.empty_like(y)
        _C.call("vspw_softmax_lastdim_bwd", _p(g), _p(y), _p(dx), y.numel() // k, k, ctx.alpha, 0, _stream())
        return dx, None


def row_softmax(x, alpha=1.0):
    return RowSoftmaxFn.apply(x, alpha)


class PixelSoftmaxFn(torch.autograd.Function):
    """softmax over the pixel axis of [B,HW,K] (dim=1)."""

    @staticmethod
    def forward(ctx, x, alpha):
        _require_gpu(x, "pixel_softmax")
        x = x.contiguous()
        B, HW, K = x.shape
        y = torch.empty_like(x)
        _C.call("vspw_softmax_pixels_fwd", _p(x), _p(y), B, HW, K, float(alpha), _stream())
        ctx.alpha = float(alpha)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        B, HW, K = y.shape
        dx = torch.empty_like(y)
        _C.call("vspw_softmax_pixels_bwd", _p(g), _p(y), _p(dx), B, HW, K, ctx.alpha, _stream())
        return dx, None


def pixel_softmax(x, alpha=1.0):
    return PixelSoftmaxFn.apply(x, alpha)


class TemporalMeanFn(torch.autograd.Function):
    """x [T*B, ...] stacked frame-major -> mean over T, reference order [current(last chunk), others...]."""

    @staticmethod
    def forward(ctx, x, T):
        _require_gpu(x, "temporal_mean")
        x = x.contiguous()
        n = x.shape[0]
        B = n // T
        inner = x.numel() // n
        y = torch.empty((B,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
        _C.call("vspw_temporal_mean_fwd", _p(x), None, _p(y), T, B, inner, _stream())
        ctx.meta = (T, B, inner, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        T, B, inner, shape = ctx.meta
        g = g.contiguous()
        dx = torch.empty(shape, device=g.device, dtype=torch.float32)
        _C.call("vspw_temporal_mean_bwd", _p(g), None, _p(dx), T, B, inner, _stream())
        return dx, None


def temporal_mean(x, T):
    return TemporalMeanFn.apply(x, T)


# --------------------------------------------------------------------------------------------------- netwarp pieces
class FlowWarpFn(torch.autograd.Function):
    """flowwarp(x, flo) of models/netwarp.py:12-37; flo is [B,2,H,W] (x-displacement, y-displacement)."""

    @staticmethod
    def forward(ctx, x, flo):
        _require_gpu(x, "flowwarp")
        x = to_nhwc(x)
        flo = to_nhwc(flo)
        n, c, h, w = x.shape
        if tuple(flo.shape) != (n, 2, h, w):
            raise RuntimeError("flowwarp: flow shape %s does not match input %s" % (tuple(flo.shape), tuple(x.shape)))
        y = empty_nhwc(n, c, h, w, x.device)
        _C.call("vspw_flowwarp_fwd", _p(x), _p(flo), _p(y), n, h, w, c, _stream())
        ctx.save_for_backward(x, flo)
        return y

    @staticmethod
    def backward(ctx, g):
        x, flo = ctx.saved_tensors
        g = to_nhwc(g)
        n, c, h, w = x.shape
        dx = empty_nhwc(n, c, h, w, x.device) if ctx.needs_input_grad[0] else None
        dflo = empty_nhwc(n, 2, h, w, x.device) if ctx.needs_input_grad[1] else None
        _C.call("vspw_flowwarp_bwd", _p(g), _p(x), _p(flo), _p(dx), _p(dflo), n, h, w, c, _stream())
        return dx, dflo


def flowwarp(x, flo):
    return FlowWarpFn.apply(x, flo)


class ChanBlendFn(torch.autograd.Function):
    """w0[c]*a + w1[c]*b (models/netwarp.py:201,216-217)."""

    @staticmethod
    def forward(ctx, a, b, w0, w1):
        _require_gpu(a, "chan_blend")
        a = to_nhwc(a)
        b = to_nhwc(b)
        n, c, h, w = a.shape
        out = empty_nhwc(n, c, h, w, a.device)
        _C.call("vspw_chan_blend_fwd", _p(a), _p(b), _p(w0), _p(w1), _p(out), n * h * w, c, _stream())
        ctx.save_for_backward(a, b, w0, w1)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, w0, w1 = ctx.saved_tensors
        g = to_nhwc(g)
        n, c, h, w = a.shape
        rows = n * h * w
        st = _stream()
        da = db = dw0 = dw1 = None
        if ctx.needs_input_grad[0]:
            da = empty_nhwc(n, c, h, w, a.device)
            _C.call("vspw_chan_scale", _p(g), _p(w0), _p(da), rows, c, st)
        if ctx.needs_input_grad[1]:
            db = empty_nhwc(n, c, h, w, a.device)
            _C.call("vspw_chan_scale", _p(g), _p(w1), _p(db), rows, c, st)
        if ctx.needs_input_grad[2]:
            dw0 = colsum(rows, c, g, a)
        if ctx.needs_input_grad[3]:
            dw1 = colsum(rows, c, g, b)
        return da, db, dw0, dw1


def chan_blend(a, b, w0, w1):
    return ChanBlendFn.apply(a, b, w0, w1)


class ChannelCatFn(torch.autograd.Function):
    """torch.cat(tensors, dim=1) for NHWC-memory tensors (strided channel-slice copies)."""

    @staticmethod
    def forward(ctx, *xs):
        _require_gpu(xs[0], "channel_cat")
        xs = [to_nhwc(x) for x in xs]
        n, _, h, w = xs[0].shape
        ctot = sum(x.shape[1] for x in xs)
        out = empty_nhwc(n, ctot, h, w, xs[0].device)
        st = _stream()
        off = 0
        widths = []
        for x in xs:
            if (x.shape[0], x.shape[2], x.shape[3]) != (n, h, w):
                raise RuntimeError("channel_cat: shape mismatch")
            c = x.shape[1]
            _C.call("vspw_copy_channels", _p(x), _p(out), n * h * w, c, c, 0, ctot, off, st)
            widths.append(c)
            off += c
        ctx.meta = (n, h, w, ctot, widths)
        return out

    @staticmethod
    def backward(ctx, g):
        n, h, w, ctot, widths = ctx.meta
        g = to_nhwc(g)
        st = _stream()
        outs = []
        off = 0
        for i, c in enumerate(widths):
            if ctx.needs_input_grad[i]:
                d = empty_nhwc(n, c, h, w, g.device)
                _C.call("vspw_copy_channels", _p(g), _p(d), n * h * w, c, ctot, off, c, 0, st)
                outs.append(d)
            else:
                outs.append(None)
            off += c
        return tuple(outs)


def channel_cat(xs):
    return ChannelCatFn.apply(*xs)


class _AsNHWCFn(torch.autograd.Function):
    """Layout change NCHW-memory -> NHWC-memory that stays on the autograd graph (identity for gradients)."""

    @staticmethod
    def forward(ctx, x):
        return to_nhwc(x)

    @staticmethod
    def backward(ctx, g):
        return g


def as_nhwc(x):
    """Differentiable to_nhwc: use this (not to_nhwc) on tensors that may require grad outside a Function."""
    return x if is_nhwc(x) else _AsNHWCFn.apply(x)


def pixels_view(x):
    """NHWC-memory [N,C,H,W] -> zero-copy [N, H*W, C] view (rows = pixels)."""
    x = as_nhwc(x)
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n, h * w, c)


def from_pixels(p, h, w):
    """[N, H*W, C] contiguous -> logical [N,C,H,W] in NHWC memory (zero-copy)."""
    n, hw, c = p.shape
    return p.reshape(n, h, w, c).permute(0, 3, 1, 2)


class ScaleFn(torch.autograd.Function):
    """y = a * x for a dense tensor of any layout (elementwise on the underlying storage order)."""

    @staticmethod
    def forward(ctx, x, a):
        _require_gpu(x, "scale")
        if not x.is_contiguous():
            x = x.contiguous()
        y = torch.empty_like(x)
        _C.call("vspw_axpby", _p(x), _p(y), x.numel(), float(a), 0.0, _stream())
        ctx.a = float(a)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dx = torch.empty_like(g)
        _C.call("vspw_axpby", _p(g), _p(dx), g.numel(), ctx.a, 0.0, _stream())
        return dx, None


def scale(x, a):
    return ScaleFn.apply(x, a)


# --------------------------------------------------------------------------------------------------- flow plumbing
class NearestResizeFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='nearest') on NCHW planes (the flow field of the NetWarp heads, netwarp.py:199,214)."""

    @staticmethod
    def forward(ctx, x, size):
        _require_gpu(x, "nearest_resize")
        x = x.contiguous()
        n, c, h, w = x.shape
        oh, ow = int(size[0]), int(size[1])
        y = torch.empty((n, c, oh, ow), device=x.device, dtype=torch.float32)
        _C.call("vspw_nearest_resize_fwd", _p(x), _p(y), n * c, h, w, oh, ow, _stream())
        ctx.meta = (n, c, h, w, oh, ow)
        return y

    @staticmethod
    def backward(ctx, g):
        n, c, h, w, oh, ow = ctx.meta
        g = g.contiguous()
        dx = torch.empty((n, c, h, w), device=g.device, dtype=torch.float32)
        _C.call("vspw_nearest_resize_bwd", _p(g), _p(dx), n * c, h, w, oh, ow, _stream())
        return dx, None


def nearest_resize(x, size):
    return NearestResizeFn.apply(x, tuple(size))


def plane_shift(x, out_hw, top, left):
    """out[..., y, x] = x[..., y - top, x - left], zero outside (no autograd: image / frozen-flow plumbing): constant
    padding for top, left >= 0, a crop for negative offsets."""
    _require_gpu(x, "plane_shift")
    x = x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty((n, c, int(out_hw[0]), int(out_hw[1])), device=x.device, dtype=torch.float32)
    _C.call("vspw_plane_shift", _p(x), _p(y), n * c, h, w, int(out_hw[0]), int(out_hw[1]), int(top), int(left), _stream())
    return y


def unnormalize_rgb(x, std, mean, post=255.0):
    """(x * std[c] + mean[c]) * post for an NCHW RGB batch (no autograd)."""
    _require_gpu(x, "unnormalize_rgb")
    x = x.contiguous()
    n, c, h, w = x.shape
    if c != 3:
        raise RuntimeError("unnormalize_rgb: 3 channels expected, got %d" % c)
    y = torch.empty_like(x)
    _C.call("vspw_unnormalize_rgb", _p(x), _p(y), n, h * w, float(std[0]), float(std[1]), float(std[2]), float(mean[0]),
            float(mean[1]), float(mean[2]), float(post), _stream())
    return y
