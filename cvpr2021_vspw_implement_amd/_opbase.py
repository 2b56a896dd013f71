"""Shared helpers of the host-side operators (ops.py): pointer / stream plumbing, the NHWC layout helpers, the
convolution descriptor and the optional per-launch timers behind bench.py's roofline figures."""
import ctypes
import os  # noqa: F401

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
    T = int(_C.query({25.0: "vspw_wino3_tiles", 36.0: "vspw_wino4_tiles", 49.0: "vspw_wino5_tiles"}.get(planes, "vspw_wino_tiles"), ctypes.byref(d)))
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
    # F(3x3,3x3): 25 planes
    "vspw_wino3_input": lambda a: _wino_bytes(a, 2, 1, planes=25.0),
    "vspw_wino3_dy": lambda a: _wino_bytes(a, 2, 1, planes=25.0),
    "vspw_wino3_output": lambda a: _wino_bytes(a, 2, 1 + _nn(a[5], a[6], a[10]), planes=25.0),
    # F(4x4,3x3): 36 planes
    "vspw_wino4_input": lambda a: _wino_bytes(a, 2, 1, planes=36.0),
    "vspw_wino4_dy": lambda a: _wino_bytes(a, 2, 1, planes=36.0),
    "vspw_wino4_output": lambda a: _wino_bytes(a, 2, 1 + _nn(a[5], a[6], a[10]), planes=36.0),
    # F(5x5,3x3): 49 planes
    "vspw_wino5_input": lambda a: _wino_bytes(a, 2, 1, planes=49.0),
    "vspw_wino5_dy": lambda a: _wino_bytes(a, 2, 1, planes=49.0),
    "vspw_wino5_output": lambda a: _wino_bytes(a, 2, 1 + _nn(a[5], a[6], a[10]), planes=49.0),
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
