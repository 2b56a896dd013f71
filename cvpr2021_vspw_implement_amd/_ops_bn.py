"""BatchNorm operators of the hot path (see ops.py): synchronised statistics, the BatchNorm(+residual+ReLU+Dropout2d)
node, inference folding and the fused conv+BN+activation node with its backward fusions (BNLink, deferred apply)."""
import ctypes
import os
import weakref

import torch

from . import _C
from ._C import ConvDesc
from ._opbase import (_Timed, _conv_desc, _conv_flops, _conv_tag, _p, _require_gpu, _stream, _ws, empty_nhwc, is_nhwc,
                      to_nhwc)
from ._ops_conv import (_fwd_apply, _wino, _wino3_conv, _wino_conv, _wino_f3, _wino_ok, _wino_takes_pending, _wt_cache, colsum,
                        conv2d_backward_data, conv2d_backward_weight, conv2d_forward)

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
    # (address, version) identify a tensor only while it is ALIVE: a model that was dropped hands its storage - and the
    # ids of its Python objects - to the next one of the same shapes (seen in the GPU suite: ResNet-18's layer1 weights
    # served to ResNet-101's layer1).  An entry is valid only for the very tensor objects it was derived from.
    if ent is not None and any((r() if r is not None else None) is not t for r, t in zip(ent[4], srcs)):
        ent = None
    if ent is None or ent[0] != key:
        coef = torch.empty((4, k), device=x.device, dtype=torch.float32)
        _C.call("vspw_bn_eval_coeffs", _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(coef[0]),
                _p(coef[1]), _p(coef[2]), _p(coef[3]), k, st)
        wf = torch.empty((k, kh, kw, c), device=x.device, dtype=torch.float32)
        bf = torch.empty(k, device=x.device, dtype=torch.float32)
        _C.call("vspw_bn_fold_weights", _p(w), _p(cbias), _p(coef[2]), _p(coef[3]), _p(wf), _p(bf), k, kh * kw * c, st)
        ent = [key, wf, bf, None, tuple(weakref.ref(t) if t is not None else None for t in srcs)]
        _infer_fold["cache"][id(w)] = ent
        if len(_infer_fold["cache"]) > 512:  # entries of weights that no longer exist
            for k_ in [k_ for k_, e_ in _infer_fold["cache"].items() if e_[4][0]() is None]:
                del _infer_fold["cache"][k_]
    _, wf, bf = ent[0], ent[1], ent[2]
    d = _conv_desc(x, k, kh, kw, stride, pad, dil)
    z = empty_nhwc(d.n, k, d.oh, d.ow, x.device)
    if residual is not None:
        residual = to_nhwc(residual)
        if tuple(residual.shape) != tuple(z.shape):
            raise RuntimeError("conv_bn_act: residual %s vs output %s" % (tuple(residual.shape), tuple(z.shape)))
    fm = _wino_f3(d) if _wino_ok(d) else 0
    if fm:  # stride-1 3x3: Winograd F(3x3,3x3) / F(4x4,3x3) on the folded weights (transform cached with them)
        planes = (fm + 2) * (fm + 2)
        if ent[3] is None or ent[3].shape[0] != planes:
            ent[3] = torch.empty((planes, k, c), device=x.device, dtype=torch.float32)
            _C.call("vspw_wino%d_weights" % fm, _p(wf), _p(ent[3]), k, c, 0, st)
        _wino3_conv(d, x, None, k, c, False, bf, z, what="fwd-fold", u=ent[3], addend=residual, act=1 if relu else 0, m=fm)
        return z
    if _wino_ok(d):  # ... F(2x2,3x3) (VSPW_WINO_F3=0)
        if ent[3] is None or ent[3].shape[0] != 16:
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
            pointwise = d.kh == 1 and d.kw == 1 and d.stride == 1 and d.pad == 0 and d.pad_w == 0
            affine = (_bn_fusion["affine"] and pointwise and not ctx.has_cbias and c >= _bn_fusion["affine_min_c"]
                      and _C.query("vspw_conv2d_bwd_aff_supported", ctypes.byref(d)) == 1)
            exchange = ctx.training and ctx.world != 1
            if affine and not exchange:  # reduction and coefficients in one launch (nothing to exchange in between)
                coef = torch.empty((3, c), device=dev, dtype=torch.float32)
                _C.call("vspw_bn_bwd_reduce_partials_coeffs_f32", _p(part), part.shape[0], c, ctypes.c_double(ctx.count),
                        _p(gamma), _p(mean), _p(invstd), train, _p(sums), _p(dgamma), _p(dbeta), _p(coef), st)
            else:
                _C.call("vspw_bn_bwd_reduce_partials_f32", _p(part), part.shape[0], c, _p(sums), _p(dgamma), _p(dbeta), st)
                if exchange:
                    _all_reduce_sums(sums)
            if affine:
                # pointwise conv: BatchNorm's backward apply becomes an affine map staged by the two gradient GEMMs of
                # this conv - dy (the gradient w.r.t. the conv output) is never written
                if exchange:
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
