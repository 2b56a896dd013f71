"""Object-Contextual-Representation blocks on the HIP kernels.

Mirrors reference models/ocr_modules/spatial_ocr_block.py: SpatialGather_Module (:39-68),
SpatialTemporalGather_Module (:70-129, incl. the inference memory bank), _ObjectAttentionBlock / ObjectAttentionBlock2D
(:176-308), SpatialOCR_Module (:310-381) — same constructor arguments and state_dict keys
(object_context_block.{f_pixel,f_object,f_down,f_up}.*, conv_bn_dropout.*).  In NHWC memory the reference's
view/permute gymnastics disappear: [B,C,H,W] *is* the [B, HW, C] matrix the matmuls want, so
  gather    : softmax over pixels (column softmax) + P^T F  (TN GEMM on the conv weight-gradient kernel)
  attention : Q K^T (NT GEMM on the conv forward kernel) -> row softmax -> (.) V
"""
import torch
import torch.nn as nn

from ... import nn as vnn
from ... import ops

BatchNorm2d = vnn.SynchronizedBatchNorm2d


def _gather_context(feats, probs, scale):
    """[n,C,h,w], [n,K,h,w] -> [n,K,C] = softmax_HW(scale*probs) @ feats."""
    P = ops.pixel_softmax(ops.pixels_view(probs), scale)  # [n,HW,K]
    return ops.bmm_tn(P, ops.pixels_view(feats))  # [n,K,C]


def _as_context_map(ctx_kc):
    """[B,K,C] -> logical [B,C,K,1] (the reference's .permute(0,2,1).unsqueeze(3)), NHWC memory, zero-copy."""
    b, k, c = ctx_kc.shape
    return ctx_kc.reshape(b, k, 1, c).permute(0, 3, 1, 2)


class SpatialGather_Module(nn.Module):
    def __init__(self, cls_num=0, scale=1, use_gt=False):
        super().__init__()
        self.cls_num = cls_num
        self.scale = scale
        self.use_gt = use_gt
        self.relu = nn.ReLU(inplace=True)

    def forward(self, feats, probs, gt_probs=None):
        if self.use_gt and gt_probs is not None:
            raise NotImplementedError("ground-truth OCR gather is not on the VSPW hot path")
        return _as_context_map(_gather_context(feats, probs, float(self.scale)))


class SpatialTemporalGather_Module(nn.Module):
    def __init__(self, cls_num=0, scale=1, use_gt=False):
        super().__init__()
        self.cls_num = cls_num
        self.scale = scale
        self.use_gt = use_gt
        self.relu = nn.ReLU(inplace=True)

    def forward(self, feats, probs, clip_num, memory=None, memory_num=None):
        assert probs.size(0) == feats.size(0)
        T = clip_num + 1
        ctx_all = _gather_context(feats, probs, float(self.scale))  # [T*B,K,C], frame-major
        B = ctx_all.shape[0] // T
        if memory is None:
            return _as_context_map(ops.temporal_mean(ctx_all, T))
        # Inference memory bank, statement-for-statement as spatial_ocr_block.py:110-125 (including the fact that a
        # non-empty bank is copied, so only contexts appended to an EMPTY bank persist across calls).
        if len(memory) > 0:
            memory = [m.detach() for m in memory]
        for t in range(T):
            while len(memory) > memory_num:
                memory.pop(0)
            memory.append(ctx_all[t * B:(t + 1) * B].detach())
        stacked = torch.cat(memory, dim=0).contiguous()
        return _as_context_map(ops.temporal_mean(stacked, len(memory)))


def _conv_bn_relu_1x1(cin, cout):
    return [vnn.Conv2d(in_channels=cin, out_channels=cout, kernel_size=1, stride=1, padding=0),
            BatchNorm2d(cout), nn.ReLU(inplace=True)]


class _ObjectAttentionBlock(nn.Module):
    def __init__(self, in_channels, key_channels, scale=1, use_gt=False, use_bg=False, fetch_attention=False):
        super().__init__()
        if scale != 1 or use_gt or use_bg or fetch_attention:
            raise NotImplementedError("only the scale=1, no-gt OCR attention of the VSPW heads is implemented")
        self.scale = scale
        self.in_channels = in_channels
        self.key_channels = key_channels
        self.use_gt = use_gt
        self.use_bg = use_bg
        self.fetch_attention = fetch_attention
        self.pool = nn.MaxPool2d(kernel_size=(scale, scale))
        self.f_pixel = vnn.FusedSequential(*(_conv_bn_relu_1x1(in_channels, key_channels)
                                             + _conv_bn_relu_1x1(key_channels, key_channels)))
        self.f_object = vnn.FusedSequential(*(_conv_bn_relu_1x1(in_channels, key_channels)
                                              + _conv_bn_relu_1x1(key_channels, key_channels)))
        self.f_down = vnn.FusedSequential(*_conv_bn_relu_1x1(in_channels, key_channels))
        self.f_up = vnn.FusedSequential(*_conv_bn_relu_1x1(key_channels, in_channels))

    def forward(self, x, proxy, gt_label=None):
        b, _, h, w = x.shape
        q = ops.pixels_view(self.f_pixel(x))  # [B,HW,Ck]
        key = ops.pixels_view(self.f_object(proxy))  # [B,K,Ck]  (proxy is [B,C,K,1])
        val = ops.pixels_view(self.f_down(proxy))  # [B,K,Ck]
        if key.shape[0] != q.shape[0]:
            if key.shape[0] != 1:
                raise RuntimeError("The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton "
                                   "dimension 0" % (q.shape[0], key.shape[0]))
            key = key.expand(q.shape[0], -1, -1)
            val = val.expand(q.shape[0], -1, -1)
        sim = ops.row_softmax(ops.bmm_nt(q, key), self.key_channels ** -0.5)  # [B,HW,K]
        ctx = ops.bmm_nt(sim, ops.transpose_last2(val))  # [B,HW,Ck]
        return self.f_up(ops.from_pixels(ctx, h, w))


class ObjectAttentionBlock2D(_ObjectAttentionBlock):
    pass


class SpatialOCR_Module(nn.Module):
    def __init__(self, in_channels, key_channels, out_channels, scale=1, dropout=0.1, use_gt=False, use_bg=False,
                 use_oc=True, fetch_attention=False):
        super().__init__()
        self.use_gt = use_gt
        self.use_bg = use_bg
        self.use_oc = use_oc
        self.fetch_attention = fetch_attention
        self.object_context_block = ObjectAttentionBlock2D(in_channels, key_channels, scale, use_gt, use_bg,
                                                           fetch_attention)
        self.conv_bn_dropout = vnn.FusedSequential(
            vnn.Conv2d(2 * in_channels, out_channels, kernel_size=1, padding=0),
            BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Dropout2d(dropout),
        )

    def forward(self, feats, proxy_feats, gt_label=None):
        context = self.object_context_block(feats, proxy_feats)
        return self.conv_bn_dropout(ops.channel_cat([context, feats]))
