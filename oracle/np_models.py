"""ORACLE (test infrastructure): numpy restatement of the reference's model graphs, composed from np_ops.

Each builder follows one reference forward() and cites it.  Parameters come from a {state_dict key: ndarray} dict
with the REFERENCE key names; BatchNorm running statistics are updated in place in that dict.
"""
import numpy as np

from . import np_ops as O
from .np_ops import Var

RESNET_LAYERS = {"resnet18": ("basic", [2, 2, 2, 2]), "resnet50": ("bottle", [3, 4, 6, 3]),
                 "resnet101": ("bottle", [3, 4, 23, 3])}


class Params:
    def __init__(self, sd, train_params=True):
        self.sd = sd
        self.vars = {}
        self.train_params = train_params

    def p(self, key):
        if key not in self.vars:
            self.vars[key] = Var(self.sd[key], needs=self.train_params)
        return self.vars[key]

    def has(self, key):
        return key in self.sd

    def grads(self):
        return {k: v.g for k, v in self.vars.items() if v.g is not None}


def _conv(P, x, pre, stride=1, pad=0, dil=1):
    b = P.p(pre + ".bias") if P.has(pre + ".bias") else None
    return O.conv2d(x, P.p(pre + ".weight"), b, stride, pad, dil)


def _bn(P, x, pre, training):
    return O.batch_norm(x, P.p(pre + ".weight"), P.p(pre + ".bias"), P.sd[pre + ".running_mean"],
                        P.sd[pre + ".running_var"], training)


def _cbr(P, x, conv, bn, training, stride=1, pad=0, dil=1, relu=True):
    y = _bn(P, _conv(P, x, conv, stride, pad, dil), bn, training)
    return O.relu(y, key=bn) if relu else y


def resnet_dilated(P, x, arch, pre, training):
    """reference models/models.py:752-767 over the ResNet of models/resnet.py:95-141 after _nostride_dilate
    (models/models.py:737-750): layer3 dil 2 (its stride-2 conv -> stride 1, dil 1), layer4 dil 4 (-> dil 2)."""
    kind, layers = RESNET_LAYERS[arch]
    x = _cbr(P, x, pre + "conv1", pre + "bn1", training, 2, 1)
    x = _cbr(P, x, pre + "conv2", pre + "bn2", training, 1, 1)
    x = _cbr(P, x, pre + "conv3", pre + "bn3", training, 1, 1)
    x = O.max_pool3x3s2(x, key=pre + "maxpool")
    outs = []
    for li, nblocks in enumerate(layers):
        stage_stride = 1 if li == 0 else 2
        dilate = {2: 2, 3: 4}.get(li)  # layer3 / layer4
        for bi in range(nblocks):
            bp = "%slayer%d.%d." % (pre, li + 1, bi)
            stride = stage_stride if bi == 0 else 1
            if dilate is None:
                s3, d3, p3 = stride, 1, 1
                sds = stride
            elif stride == 2:  # the de-strided conv: stride 1, dilation dilate//2
                s3, d3, p3 = 1, dilate // 2, dilate // 2
                sds = 1
            else:
                s3, d3, p3 = 1, dilate, dilate
                sds = 1
            res = x
            if P.has(bp + "downsample.0.weight"):
                res = _cbr(P, x, bp + "downsample.0", bp + "downsample.1", training, sds, 0, 1, relu=False)
            if kind == "bottle":
                y = _cbr(P, x, bp + "conv1", bp + "bn1", training)
                y = _cbr(P, y, bp + "conv2", bp + "bn2", training, s3, p3, d3)
                y = _cbr(P, y, bp + "conv3", bp + "bn3", training, relu=False)
                last = bp + "bn3"
            else:
                # BasicBlock: conv1 carries the stride (and is the one de-strided); conv2 is a plain 3x3 -> dilated
                y = _cbr(P, x, bp + "conv1", bp + "bn1", training, s3, p3, d3)
                d2 = dilate if dilate is not None else 1
                y = _cbr(P, y, bp + "conv2", bp + "bn2", training, 1, d2, d2, relu=False)
                last = bp + "bn2"
            x = O.relu(O.add(y, res), key=last)  # the block's output ReLU, keyed by the BatchNorm that feeds it
        outs.append(x)
    return outs


def _ppm_concat(P, conv5, pooled, pre, conv_idx, bn_idx, training):
    h, w = conv5.shape[2:]
    outs = [conv5]
    for i, pf in enumerate(pooled):
        b = _cbr(P, pf, "%s%d.%d" % (pre, i, conv_idx), "%s%d.%d" % (pre, i, bn_idx), training)
        outs.append(O.interpolate_bilinear(b, (h, w)))
    return O.cat(outs, 1)


def _head(P, x, pre, training, mask=None, last=4):
    y = _cbr(P, x, pre + ".0", pre + ".1", training, 1, 1)
    y = O.dropout2d_mask(y, mask)
    return _conv(P, y, pre + ".%d" % last)


def ppm_deepsup(P, feats, pre, training, seg_size=None, scales=(1, 2, 3, 6)):
    """reference models/models.py:938-995 (PPMDeepsup)."""
    conv5 = feats[-1]
    pooled = [O.adaptive_avg_pool2d(conv5, s) for s in scales]
    x = _head(P, _ppm_concat(P, conv5, pooled, pre + "ppm.", 1, 2, training), pre + "conv_last_", training)
    if seg_size is not None:
        return O.softmax(O.interpolate_bilinear(x, seg_size), 1)
    ds = _cbr(P, feats[-2], pre + "cbr_deepsup.0", pre + "cbr_deepsup.1", training, 1, 1)
    ds = _conv(P, ds, pre + "conv_last_deepsup_")
    return O.log_softmax(x, 1), O.log_softmax(ds, 1)


def segmentation_module(P, arch, img, label, training, deep_sup_scale=0.4, seg_size=None, decoder="ppm_deepsup"):
    """reference models/models.py:82-111 (SegmentationModule.forward)."""
    feats = resnet_dilated(P, Var(img), arch, "encoder.", training)
    dec = {"ppm_deepsup": ppm_deepsup, "ocrnet_deepsup": ocrnet, "nonlocal2d": nonlocal2d}[decoder]
    if seg_size is not None:
        return dec(P, feats, "decoder.", training, seg_size)
    out = dec(P, feats, "decoder.", training)
    h, w = label.shape[2:]
    if isinstance(out, tuple):
        pred, ds = out
    else:
        pred, ds = out, None
    pu = O.interpolate_bilinear(pred, (h, w))
    loss = O.nll_loss(pu, label)
    if ds is not None and deep_sup_scale is not None:
        loss = O.add(loss, O.scale(O.nll_loss(O.interpolate_bilinear(ds, (h, w)), label), deep_sup_scale))
    return loss, O.pixel_acc(pu.v, label)


def clip_psp(P, arch, frames, labels, training, deep_sup_scale=0.4, seg_size=None, scales=(1, 2, 3, 6)):
    """reference models/clip_psp.py:136-217.  frames/labels: lists with the CURRENT frame LAST (the reference
    appends it, :142,197)."""
    T = len(frames)
    B = frames[0].shape[0]
    feats = resnet_dilated(P, Var(np.concatenate(frames, 0)), arch, "encoder.", training)
    chunks = O.split_batch(feats[-1], B)
    cur, others = chunks[-1], chunks[:-1]
    p_fs = []
    for s in scales:
        feats_s = [O.adaptive_avg_pool2d(cur, s)] + [O.adaptive_avg_pool2d(o, s) for o in others]
        p_fs.append(O.mean_stack(feats_s))
    x = _head(P, _ppm_concat(P, cur, p_fs, "ppm_conv.ppm.", 0, 1, training), "ppm_conv.conv_last_", training)
    if seg_size is not None:
        return O.softmax(O.interpolate_bilinear(x, seg_size), 1), x
    label = labels[-1]
    h, w = label.shape[2:]
    pu = O.interpolate_bilinear(O.log_softmax(x, 1), (h, w))
    loss = O.nll_loss(pu, label)
    if deep_sup_scale is not None:
        ds = _head(P, feats[-2], "deepsup", training)
        du = O.interpolate_bilinear(O.log_softmax(ds, 1), (h, w))
        loss = O.add(loss, O.scale(O.nll_loss(du, np.concatenate(labels, 0)), deep_sup_scale))
    return loss, O.pixel_acc(pu.v, label)


def _gather(feats, probs):
    """reference spatial_ocr_block.py:100-105: softmax over HW of the class maps, times the features."""
    b, k, h, w = probs.shape
    c = feats.shape[1]
    p = O.softmax(O.reshape(probs, (b, k, h * w)), 2)
    f = O.transpose(O.reshape(feats, (b, c, h * w)), (0, 2, 1))
    ctx = O.matmul(p, f)  # b x k x c
    return O.reshape(O.transpose(ctx, (0, 2, 1)), (b, c, k, 1))


def _ocr_attention(P, x, proxy, pre, training):
    """reference spatial_ocr_block.py:247-289 (_ObjectAttentionBlock) + :358-381 (SpatialOCR_Module)."""
    b, c, h, w = x.shape
    ob = pre + "object_context_block."
    q = _cbr(P, _cbr(P, x, ob + "f_pixel.0", ob + "f_pixel.1", training), ob + "f_pixel.3", ob + "f_pixel.4", training)
    k = _cbr(P, _cbr(P, proxy, ob + "f_object.0", ob + "f_object.1", training), ob + "f_object.3", ob + "f_object.4",
             training)
    v = _cbr(P, proxy, ob + "f_down.0", ob + "f_down.1", training)
    kc = q.shape[1]
    qm = O.transpose(O.reshape(q, (b, kc, h * w)), (0, 2, 1))
    km = O.reshape(k, (k.shape[0], kc, -1))
    vm = O.transpose(O.reshape(v, (v.shape[0], kc, -1)), (0, 2, 1))
    sim = O.softmax(O.scale(O.matmul(qm, km), kc ** -0.5), -1)
    ctx = O.reshape(O.transpose(O.matmul(sim, vm), (0, 2, 1)), (b, kc, h, w))
    ctx = _cbr(P, ctx, ob + "f_up.0", ob + "f_up.1", training)
    return _cbr(P, O.cat([ctx, x], 1), pre + "conv_bn_dropout.0", pre + "conv_bn_dropout.1", training)


def ocrnet(P, feats, pre, training, seg_size=None):
    """reference models/ocrnet.py:56-72 (SpatialOCRNet.forward)."""
    x_dsn = _head(P, feats[-2], pre + "dsn_head", training)
    x = _cbr(P, feats[-1], pre + "conv_3x3.0", pre + "conv_3x3.1", training, 1, 1)
    context = _gather(x, x_dsn)
    x = _conv(P, _ocr_attention(P, x, context, pre + "spatial_ocr_head.", training), pre + "head")
    if seg_size is not None:
        return O.softmax(O.interpolate_bilinear(x, seg_size), 1)
    return O.log_softmax(x, 1), O.log_softmax(x_dsn, 1)


def clip_ocr(P, arch, frames, labels, training, deep_sup_scale=0.4, seg_size=None):
    """reference models/clip_ocr.py:106-198 with clipocr_all=False, use_memory=False."""
    T = len(frames)
    B = frames[0].shape[0]
    feats = resnet_dilated(P, Var(np.concatenate(frames, 0)), arch, "encoder.", training)
    x_dsn = _head(P, feats[-2], "dsn_head", training)
    out = _cbr(P, feats[-1], "conv_3x3.0", "conv_3x3.1", training, 1, 1)
    ctxs = [_gather(f, p) for f, p in zip(O.split_batch(out, B), O.split_batch(x_dsn, B))]
    context = O.mean_stack(ctxs)
    x = O.split_batch(out, B)[-1]
    x = _conv(P, _ocr_attention(P, x, context, "spatial_ocr_head.", training), "head")
    if seg_size is not None:
        return O.softmax(O.interpolate_bilinear(x, seg_size), 1), x
    label = labels[-1]
    h, w = label.shape[2:]
    pu = O.interpolate_bilinear(O.log_softmax(x, 1), (h, w))
    loss = O.nll_loss(pu, label)
    du = O.interpolate_bilinear(O.log_softmax(x_dsn, 1), (h, w))
    loss = O.add(loss, O.scale(O.nll_loss(du, np.concatenate(labels, 0)), deep_sup_scale))
    return loss, O.pixel_acc(pu.v, label)


def nl_block(P, x, pre, training):
    """reference models/non_local.py:82-151, mode 'dot': z = W_z(theta^T phi / N . g) + x; x is [B,C,N,1]."""
    b, c, n, _ = x.shape
    g = _conv(P, x, pre + "g")
    th = _conv(P, x, pre + "theta")
    ph = _conv(P, x, pre + "phi")
    ci = g.shape[1]
    gm = O.transpose(O.reshape(g, (b, ci, n)), (0, 2, 1))
    tm = O.transpose(O.reshape(th, (b, ci, n)), (0, 2, 1))
    pm = O.reshape(ph, (b, ci, n))
    f = O.scale(O.matmul(tm, pm), 1.0 / n)
    y = O.reshape(O.transpose(O.matmul(f, gm), (0, 2, 1)), (b, ci, n, 1))
    wy = _bn(P, _conv(P, y, pre + "W_z.0"), pre + "W_z.1", training)
    return O.add(wy, x)


def _flat_w(sd, keys):
    """Conv3d 1x1x1 weights [O,I,1,1,1] -> [O,I,1,1] so conv2d applies (dimension=3 block)."""
    for k in keys:
        if k in sd and sd[k].ndim == 5:
            sd[k] = sd[k].reshape(sd[k].shape[:2] + (1, 1))


def nonlocal2d(P, feats, pre, training, seg_size=None, downsample=False):
    """reference models/non_local_models.py:124-151 (Non_local2d.forward); downsample (:135-138): the block runs on
    the 2x2-average-pooled embedding and its output is bilinearly resized back."""
    emb = _conv(P, feats[-1], pre + "emb")
    b, c, h, w = emb.shape
    src = O.avg_pool2x2(emb) if downsample else emb
    hs, ws = src.shape[2:]
    z = O.reshape(nl_block(P, O.reshape(src, (b, c, hs * ws, 1)), pre + "nonlocalblock.", training), (b, c, hs, ws))
    if downsample:
        z = O.interpolate_bilinear(z, (h, w))
    pred = _conv(P, O.cat([emb, z], 1), pre + "last_layer")
    if seg_size is not None:
        return O.softmax(O.interpolate_bilinear(pred, seg_size), 1)
    return O.log_softmax(pred, 1)


def nonlocal3d(P, arch, frames, labels, training, seg_size=None, downsample=False):
    """reference models/non_local_models.py:19-72 (Non_local3d.forward): one non-local block over the T*h*w positions of
    a clip ([B,C,T,h,w], Conv3d 1x1x1 = per-position linear maps); per-frame losses / accuracies are averaged."""
    T = len(frames)
    B = frames[0].shape[0]
    emb = _conv(P, resnet_dilated(P, Var(np.concatenate(frames, 0)), arch, "encoder.", training)[-1], "emb")
    n, c, h, w = emb.shape
    src = O.avg_pool2x2(emb) if downsample else emb
    hs, ws = src.shape[2:]
    # [T*B,C,hs,ws] -> per clip [B, C, T*hs*ws, 1]
    x = O.transpose(O.reshape(src, (T, B, c, hs * ws)), (1, 2, 0, 3))
    z = nl_block(P, O.reshape(x, (B, c, T * hs * ws, 1)), "nonlocalblock.", training)
    z = O.reshape(O.transpose(O.reshape(z, (B, c, T, hs * ws)), (2, 0, 1, 3)), (n, c, hs, ws))
    if downsample:
        z = O.interpolate_bilinear(z, (h, w))
    pred = _conv(P, O.cat([emb, z], 1), "last_layer")
    preds = O.split_batch(pred, B)
    if seg_size is not None:
        return [O.softmax(O.interpolate_bilinear(p, seg_size), 1) for p in preds]
    loss, acc = None, 0.0
    for p, lab in zip(preds, labels):
        pu = O.interpolate_bilinear(O.log_softmax(p, 1), lab.shape[2:])
        l_ = O.nll_loss(pu, lab)
        loss = l_ if loss is None else O.add(loss, l_)
        acc += O.pixel_acc(pu.v, lab)
    return O.scale(loss, 1.0 / T), acc / T
