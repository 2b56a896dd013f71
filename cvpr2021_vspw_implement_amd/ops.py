"""Host-side operators of the VSPW hot path: thin autograd.Functions that launch the hand-written HIP kernels of
libvspw_hip.so through its C ABI (include/vspw_hip.h).  PyTorch is used for device memory, streams and autograd
bookkeeping only; every FLOP of these operators runs in csrc/*.hip.

Tensors keep the reference's logical NCHW shapes but live in NHWC memory (torch channels_last), so `x.size()` reads
like the reference while the kernels see [pixel rows][channel columns].

There is no CPU fallback: a CPU tensor raises (SURVEY.md §8b "Errors").

Layout of the module family: _opbase.py (pointer / layout helpers, launch timers), _ops_conv.py (convolutions, Winograd
dispatch, derived-weight caches), _ops_bn.py (BatchNorm, conv+BN+activation nodes); this file holds the remaining
operators (pooling, interpolation, losses, batched GEMMs, non-local, flow) and re-exports the others, so callers keep
using `ops.<name>`.
"""
import ctypes
import os

import torch

from . import _C
from ._C import ConvDesc
from ._opbase import (  # noqa: F401 (re-exported: callers use ops.<name>)
    _HBM_BYTES, _HbmTimed, _NLL_FIXED, _NO_TRACE, _NoTrace, _Timed, _conv_desc, _conv_flops, _conv_tag,
    _hbm_trace, _ktimer, _nn, _p, _require_gpu, _stream, _vp, _wino_bytes, _ws, empty_nhwc, hbm_timer_records,
    is_nhwc, kernel_timer, kernel_timer_records, kernel_timer_reset, to_nhwc)
from ._ops_conv import (  # noqa: F401 (re-exported: callers use ops.<name>)
    Conv2dFn, _DerivedWeights, _fwd_apply, _transposed_weight, _wgrad_launch, _wgrad_side, _wino, _wino_conv,
    _wino_ok, _wino_takes_pending, _wino_weights, _wino_wgrad, _wt_alloc, _wt_cache, _wt_copies, _wt_key,
    _wt_single, _wu_alloc, _wu_copies, _wu_single, colsum, conv2d, conv2d_backward_data, conv2d_backward_weight,
    conv2d_forward, drop_weight_transpose_cache, join_side_streams, set_accum_chunk, set_wgrad_side_stream, set_winograd,
    _wino3_conv, _wino3_weights, _wino_f3, _wu3_copies, accum_chunk_supported, set_winograd_f3, set_winograd_tile, winograd_tile_hint)
from ._ops_bn import (  # noqa: F401 (re-exported: callers use ops.<name>)
    BNLink, BatchNormActFn, ConvBNActFn, _BN_SMALL_ROWS, _all_reduce_sums, _bn_fusion, _conv_bn_folded,
    _decisions, _finalize_name, _infer_fold, _sync_finalize, _sync_group, _sync_world, batch_norm_act,
    conv_bn_act, invalidate_inference_cache, materialize, record_decisions, set_bn_backward_fusion,
    set_inference_folding, set_sync_bn, sync_bn_timer)


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


def flowwarp_nearest(x, flo):
    """flowwarp of TC_cal.py:12-38 (grid_sample mode='nearest'): label maps carried along the flow; no gradient."""
    _require_gpu(x, "flowwarp_nearest")
    x, flo = to_nhwc(x.detach()), to_nhwc(flo.detach())
    n, c, h, w = x.shape
    if tuple(flo.shape) != (n, 2, h, w):
        raise RuntimeError("flowwarp_nearest: flow shape %s does not match input %s" % (tuple(flo.shape), tuple(x.shape)))
    y = empty_nhwc(n, c, h, w, x.device)
    _C.call("vspw_flowwarp_nearest", _p(x), _p(flo), _p(y), n, h, w, c, _stream())
    return y


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
    if x.dtype != torch.float32:
        raise TypeError("plane_shift: float32 planes only (got %s); callers cast like the reference's .float() call sites" % x.dtype)
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
