"""TCB-PSP: PSPNet head with Temporal Context Blending of the pyramid-pooled features across the T frames of a clip.

Mirrors reference models/clip_psp.py:23-217 (PPM_conv, Clip_PSP): same constructor, feed_dict keys, side effects on
the caller's lists, state_dict keys (ppm_conv.ppm.i.{0,1}.*, ppm_conv.conv_last_.{0,1,4}.*, deepsup.{0,1,4}.*) and
(loss, acc) / softmax-probability outputs.  Execution is a short chain of fused HIP kernels:
encoder (conv+BN+ReLU chains) -> one pyramid-pool + temporal-mean pass -> 4x (1x1 conv+BN+ReLU) -> bilinear-up+concat
-> 3x3 conv+BN+ReLU+Dropout2d -> 1x1 conv -> fused log-softmax/bilinear/NLL/pixel-acc loss.
"""
import torch
import torch.nn as nn

from .. import nn as vnn
from .. import ops
from .lr_groups import LrGroupsMixin
from .models import nll_ignore_index
from ._metrics import pixel_accuracy

BatchNorm2d = vnn.SynchronizedBatchNorm2d


class PPM_conv(nn.Module):
    def __init__(self, fc_dim=2048, num_class=None, pool_scales=(1, 2, 3, 6)):
        super().__init__()
        self.ppm = nn.ModuleList([
            vnn.FusedSequential(vnn.Conv2d(fc_dim, 512, kernel_size=1, bias=False), BatchNorm2d(512),
                                nn.ReLU(inplace=True))
            for _ in pool_scales
        ])
        self.conv_last_ = vnn.FusedSequential(
            vnn.Conv2d(fc_dim + len(pool_scales) * 512, 512, kernel_size=3, padding=1, bias=False),
            BatchNorm2d(512),
            nn.ReLU(inplace=True),
            nn.Dropout2d(0.1),
            vnn.Conv2d(512, num_class, kernel_size=1),
        )

    def forward(self, x, xs):
        branches = [seq(x_) for seq, x_ in zip(self.ppm, xs)]
        return self.conv_last_(ops.ppm_concat(x, branches))


class Clip_PSP(LrGroupsMixin, nn.Module):
    def __init__(self, net_enc, crit, args, pool_scales=(1, 2, 3, 6), deep_sup_scale=None):
        super().__init__()
        self.encoder = net_enc
        self.crit = crit
        self.deep_sup_scale = deep_sup_scale
        self.args = args
        fc_dim = 2048
        self.pool_scales = pool_scales
        self.ppm_conv = PPM_conv(fc_dim, args.num_class, pool_scales=pool_scales)
        self.deepsup = vnn.FusedSequential(
            vnn.Conv2d(fc_dim // 2, fc_dim // 4, kernel_size=3, stride=1, padding=1, bias=False),
            BatchNorm2d(fc_dim // 4),
            nn.ReLU(inplace=True),
            nn.Dropout2d(0.1),
            vnn.Conv2d(fc_dim // 4, args.num_class, 1, 1, 0),
        )
        if self.args.psp_weight:
            self.pspweight_conv = nn.Sequential(vnn.Conv2d(fc_dim, 1, kernel_size=1, bias=False),
                                                vnn.AdaptiveAvgPool2d((1, 1)))
        self.ppm_pool = nn.ModuleList([vnn.AdaptiveAvgPool2d(scale) for scale in pool_scales])

    def _lr_10x_roots(self):
        roots = [self.ppm_conv]
        if self.deep_sup_scale is not None:
            roots.append(self.deepsup)
        if self.args.psp_weight:
            roots.append(self.pspweight_conv)
        return roots

    def _lr_10x_bias_roots(self):  # the reference's bias list omits pspweight_conv (clip_psp.py:127-135)
        roots = [self.ppm_conv]
        if self.deep_sup_scale is not None:
            roots.append(self.deepsup)
        return roots

    def pixel_acc(self, pred, label):
        return pixel_accuracy(pred, label)

    def _temporal_weights(self, conv5, T):
        """softmax over the T frames of the pooled 1x1-conv score (clip_psp.py:147-152) -> [B, T]."""
        score = self.pspweight_conv(conv5)  # [T*B,1,1,1]
        B = score.shape[0] // T
        return ops.row_softmax(score.reshape(T, B).t().contiguous(), 1.0).contiguous()

    def forward(self, feed_dict, segSize=None):
        c_img = feed_dict["img_data"]
        clip_imgs = feed_dict["clipimgs_data"]
        label = feed_dict["seg_label"]
        clip_num = len(clip_imgs)
        T = clip_num + 1
        clip_imgs.append(c_img)  # same side effect as the reference (clip_psp.py:142)
        frames = torch.cat(clip_imgs, dim=0)  # current frame is the LAST chunk
        feats = self.encoder(frames, return_feature_maps=True)
        conv5 = feats[-1]
        B = conv5.shape[0] // T
        wts = self._temporal_weights(conv5, T) if self.args.psp_weight else None
        # blended pools of all T frames + the current frames (the last B of the batch) from one node: see PyramidPoolFn
        *blended, cur = ops.pyramid_pool(conv5, self.pool_scales, T, wts, tail=B)
        pred_ = self.ppm_conv(cur, blended)
        if segSize is not None:
            return ops.upsample_softmax(pred_, segSize)
        ignore = nll_ignore_index(self.crit)
        clip_labels = feed_dict["cliplabels_data"]
        clip_labels.append(label)
        loss, acc = ops.seg_nll(pred_, label, ignore, want_acc=True, from_logits=True)
        if self.deep_sup_scale is not None:
            alllabel = torch.cat(clip_labels, dim=0)
            loss_deepsup, _ = ops.seg_nll(self.deepsup(feats[-2]), alllabel, ignore, want_acc=False, from_logits=True)
            loss = loss + loss_deepsup * self.deep_sup_scale
        return loss, acc
