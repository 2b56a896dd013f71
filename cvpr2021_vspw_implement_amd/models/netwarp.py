"""NetWarp (optical-flow feature warping), mirroring reference models/netwarp.py:12-239.

`flowwarp` and the per-channel blend run on the HIP kernels (csrc/misc.hip).  The frozen flow provider is RAFT on the
HIP kernels as well (models/raft.py + csrc/raft.hip); like the reference (netwarp.py:71-77) it
is built in the constructor and filled from `./RAFT_core/raft-things.pth-no-zip`.  `args.raft_weights` overrides the
checkpoint path ('' / None: keep the random initialisation - benchmarks, tests); `args.flow_net` substitutes any
nn.Module with RAFT's forward(img1, img2, iters, test_mode) -> (low, up) signature.
"""
import torch
import torch.nn as nn

from .. import nn as vnn
from .. import ops
from .lr_groups import LrGroupsMixin
from .models import conv3x3_bn_relu, nll_ignore_index
from ._metrics import pixel_accuracy

BatchNorm2d = vnn.SynchronizedBatchNorm2d


def flowwarp(x, flo):
    """Warp x [B,C,H,W] by flo [B,2,H,W]: grid_sample(x, 2*(grid+flo)/(dim-1)-1, bilinear, zeros,
    align_corners=False) — the (dim-1)/align_corners=False mismatch of the reference is preserved."""
    return ops.flowwarp(x, flo)


class FlowCNN(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = conv3x3_bn_relu(11, 16)
        self.conv2 = conv3x3_bn_relu(16, 32)
        self.conv3 = conv3x3_bn_relu(32, 2)
        self.conv4 = conv3x3_bn_relu(4, 2)

    def forward(self, img1, img2, flow):
        x = ops.channel_cat([flow, img1, img2, img2 - img1])
        x = self.conv3(self.conv2(self.conv1(x)))
        return self.conv4(ops.channel_cat([flow, x]))


def _load_flow_net(args):
    net = getattr(args, "flow_net", None)
    if net is not None:
        return net
    from collections import OrderedDict

    from .raft import RAFT

    raft = RAFT()
    weights = getattr(args, "raft_weights", "./RAFT_core/raft-things.pth-no-zip")
    if weights:
        to_load = torch.load(weights, map_location="cpu")
        raft.load_state_dict(OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in to_load.items()))
    return raft


def _pad_to_8(x):
    """RAFT's InputPadder('sintel') (RAFT_core/utils/utils.py:7-25): zero-pad H,W up to multiples of 8 (the
    reference's replicate mode is commented out, utils.py:19-20: mode='constant')."""
    from ..RAFT_core.utils.utils import margins_to_multiple_of_8

    h, w = x.shape[-2:]
    top, bottom, left, right = margins_to_multiple_of_8(h, w)
    return ops.plane_shift(x, (h + top + bottom, w + left + right), top, left), [left, right, top, bottom]


class _NetWarpBase(LrGroupsMixin, nn.Module):
    """Shared flow / warp / blend plumbing of NetWarp (models/netwarp.py:66-239) and NetWarp_ocr
    (models/netwarp_ocr.py:121-299)."""

    def _init_common(self, net_enc, net_dec, crit, args, deep_sup_scale, blend2_channels, head=None):
        # sub-module registration order = the reference's (it fixes the order of state_dict keys)
        self.raft = _load_flow_net(args)
        self.mean = torch.FloatTensor([0.485, 0.456, 0.406])
        self.std = torch.FloatTensor([0.229, 0.224, 0.225])
        self.encoder = net_enc
        self.decoder = net_dec
        if head is not None:
            self.head = head
        self.crit = crit
        self.deep_sup_scale = deep_sup_scale
        self.args = args
        assert self.args.clip_num == 2
        self.flowcnn = FlowCNN()
        self.w0_0 = nn.Parameter(torch.ones(2048))
        self.w0_1 = nn.Parameter(torch.zeros(2048))
        self.w1_0 = nn.Parameter(torch.ones(blend2_channels))
        self.w1_1 = nn.Parameter(torch.zeros(blend2_channels))

    def get_10x_lr_params(self):
        for p in super().get_10x_lr_params():
            yield p
        for w in [self.w0_0, self.w1_0, self.w1_1, self.w0_1]:  # reference order
            yield w

    def pixel_acc(self, pred, label):
        return pixel_accuracy(pred, label)

    def _flow(self, cur255, prev255):
        with torch.no_grad():
            self.raft.eval()
            a, pad = _pad_to_8(cur255)
            b, _ = _pad_to_8(prev255)
            _, flow = self.raft(a, b, iters=20, test_mode=True)
            hh, ww = flow.shape[-2:]
            return ops.plane_shift(flow, (hh - pad[2] - pad[3], ww - pad[0] - pad[1]), -pad[2], -pad[0])  # unpad

    def _refined_flow(self, feed_dict):
        c_img = feed_dict["img_data"]
        clip_imgs = feed_dict["clipimgs_data"]
        assert len(clip_imgs) == 1
        c_pre_img = clip_imgs[0]
        # image un-normalisation (x * std + mean) * 255: input plumbing for the flow net
        c_img_f = ops.unnormalize_rgb(c_img, self.std.tolist(), self.mean.tolist(), 255.0)
        c_pre_img_f = ops.unnormalize_rgb(c_pre_img, self.std.tolist(), self.mean.tolist(), 255.0)
        flow = feed_dict["flow"] if "flow" in feed_dict else self._flow(c_img_f, c_pre_img_f)
        return c_img, c_pre_img, self.flowcnn(c_img_f, c_pre_img_f, flow)

    def _warp_blend(self, feats, flow, w_cur, w_warp):
        """feats = [current; previous] stacked on the batch: blend the current half with the flow-warped previous."""
        B = feats.shape[0] // 2
        cur, prev = ops.split_batch(feats, B)
        flow_s = ops.nearest_resize(flow, cur.shape[-2:])  # nearest, magnitudes NOT rescaled (quirk)
        return ops.chan_blend(cur, ops.flowwarp(prev, flow_s), w_cur, w_warp), prev


class NetWarp(_NetWarpBase):
    def __init__(self, net_enc, net_dec, crit, args, deep_sup_scale=None):
        super().__init__()
        self._init_common(net_enc, net_dec, crit, args, deep_sup_scale, 4096)
        self.conv_last_ = vnn.FusedSequential(
            vnn.Conv2d(2048 + 4 * 512, 512, kernel_size=3, padding=1, bias=False),
            BatchNorm2d(512),
            nn.ReLU(inplace=True),
            nn.Dropout2d(0.1),
            vnn.Conv2d(512, args.num_class, kernel_size=1),
        )

    def _lr_10x_roots(self):
        return [self.decoder, self.flowcnn, self.conv_last_]

    def forward(self, feed_dict, *, segSize=None):
        if feed_dict is None:
            return torch.zeros((0, self.args.num_class, 480, 720)).cuda()
        label = feed_dict["seg_label"]
        c_img, c_pre_img, flow = self._refined_flow(feed_dict)
        feats = self.encoder(torch.cat([c_img, c_pre_img], 0), return_feature_maps=True)
        B = c_img.shape[0]
        new_cur1, prev1 = self._warp_blend(feats[-1], flow, self.w0_0, self.w0_1)
        feats[-1] = torch.cat([new_cur1, prev1], 0)
        pred_deepsup_s, _, ppm_cat = self.decoder(feats)
        new_feat, _ = self._warp_blend(ppm_cat, flow, self.w1_0, self.w1_1)
        pred_ = self.conv_last_(new_feat)
        if segSize is not None:
            return ops.upsample_softmax(pred_, segSize)
        ignore = nll_ignore_index(self.crit)
        loss, acc = ops.seg_nll(pred_, label, ignore, want_acc=True, from_logits=True)
        if self.deep_sup_scale is not None:
            loss_deepsup, _ = ops.seg_nll(ops.split_batch(pred_deepsup_s, B)[0], label, ignore, want_acc=False,
                                          from_logits=False)
            loss = loss + loss_deepsup * self.deep_sup_scale
        return loss, acc


class SpatialOCRNetasDec(nn.Module):
    """OCR decoder without the class head (models/netwarp_ocr.py:65-117): returns (512-ch OCR features, dsn logits)."""

    def __init__(self, num_class):
        self.inplanes = 128
        super().__init__()
        from .ocr_modules.spatial_ocr_block import SpatialGather_Module, SpatialOCR_Module
        from .ocrnet import ocr_heads

        self.num_classes = num_class
        self.conv_3x3, _, dsn_head = ocr_heads(num_class)
        self.spatial_context_head = SpatialGather_Module(self.num_classes)
        self.spatial_ocr_head = SpatialOCR_Module(in_channels=512, key_channels=256, out_channels=512, scale=1,
                                                  dropout=0.05)
        self.dsn_head = dsn_head

    def forward(self, x):
        x_dsn = self.dsn_head(x[-2])
        x = self.conv_3x3(x[-1])
        context = self.spatial_context_head(x, x_dsn)
        return self.spatial_ocr_head(x, context), x_dsn


class NetWarp_ocr(_NetWarpBase):
    def __init__(self, net_enc, crit, args, deep_sup_scale=None):
        super().__init__()
        self._init_common(net_enc, SpatialOCRNetasDec(args.num_class), crit, args, deep_sup_scale, 512,
                          head=vnn.Conv2d(512, args.num_class, kernel_size=1, stride=1, padding=0, bias=True))

    def _lr_10x_roots(self):
        return [self.decoder, self.flowcnn, self.head]

    def forward(self, feed_dict, *, segSize=None):
        if feed_dict is None:
            return torch.zeros((0, self.args.num_class, 480, 720)).cuda()
        label = feed_dict["seg_label"]
        c_img, c_pre_img, flow = self._refined_flow(feed_dict)
        feats = self.encoder(torch.cat([c_img, c_pre_img], 0), return_feature_maps=True)
        new_cur1, prev1 = self._warp_blend(feats[-1], flow, self.w0_0, self.w0_1)
        feats[-1] = torch.cat([new_cur1, prev1], 0)
        ocr_feats, x_dsn = self.decoder(feats)
        new_feat, _ = self._warp_blend(ocr_feats, flow, self.w1_0, self.w1_1)
        pred_ = self.head(new_feat)
        if segSize is not None:
            return ops.upsample_softmax(pred_, segSize)
        ignore = nll_ignore_index(self.crit)
        loss, acc = ops.seg_nll(pred_, label, ignore, want_acc=True, from_logits=True)
        clip_label = feed_dict["cliplabels_data"]
        clip_label.append(feed_dict["seg_label"])  # deepsup covers [previous..., current] = all 2B frames
        loss_deepsup, _ = ops.seg_nll(x_dsn, torch.cat(clip_label, dim=0), ignore, want_acc=False, from_logits=True)
        loss = loss + loss_deepsup * self.deep_sup_scale
        return loss, acc
