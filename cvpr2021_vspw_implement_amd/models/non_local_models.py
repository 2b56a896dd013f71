"""Non-local decoders: per-frame `nonlocal2d` and the clip-level Non_local3d, mirroring reference
models/non_local_models.py:9-151 (constructor, feed_dict protocol, state_dict keys emb.*, nonlocalblock.*, last_layer.*).
"""
import torch
import torch.nn as nn

from .. import nn as vnn
from .. import ops
from .lr_groups import LrGroupsMixin
from .models import nll_ignore_index
from .non_local import NLBlockND
from ._metrics import pixel_accuracy


class Non_local3d(LrGroupsMixin, nn.Module):
    def __init__(self, args, net_enc, crit, downsample=False):
        super().__init__()
        self.encoder = net_enc
        self.downsample = downsample
        self.crit = crit
        self.emb = vnn.Conv2d(2048, 256, 1, 1)
        self.nonlocalblock = NLBlockND(in_channels=256, mode="dot", dimension=3, bn_layer=True)
        self.last_layer = vnn.Conv2d(512, args.num_class, kernel_size=1, stride=1)

    def _lr_10x_roots(self):
        return [self.emb, self.nonlocalblock, self.last_layer]

    def pixel_acc(self, pred, label):
        return pixel_accuracy(pred, label)

    def forward(self, feed_dict, segSize=None):
        clip_imgs = feed_dict["clipimgs_data"]
        labels = feed_dict["cliplabels_data"]
        clip_num = len(clip_imgs)
        frames = torch.cat(clip_imgs, dim=0)
        emb = self.emb(self.encoder(frames, return_feature_maps=True)[-1])  # [T*B,256,h,w]
        n, c, h, w = emb.shape
        B = n // clip_num
        # [T*B,C,h,w] -> [B,C,T,h,w]: data movement only (torch views + one gather copy)
        # `downsample` (reference :30-32,43-44): affinity on the 2x2-average-pooled embedding, bilinear back up
        emb_ = ops.avg_pool2x2(emb) if self.downsample else emb
        hs, ws = emb_.shape[-2:]
        x = emb_.reshape(clip_num, B, c, hs, ws).permute(1, 2, 0, 3, 4)
        x = self.nonlocalblock(x)
        x = x.permute(2, 0, 1, 3, 4).reshape(n, c, hs, ws)
        if self.downsample:
            x = ops.interpolate_bilinear(x, (h, w))
        x = self.last_layer(ops.channel_cat([emb, x]))
        preds = torch.split(x, B, dim=0)
        if segSize is None:
            ignore = nll_ignore_index(self.crit)
            losses, accs = [], []
            for p, lab in zip(preds, labels):
                l_, a_ = ops.seg_nll(p, lab, ignore, want_acc=True, from_logits=True)
                losses.append(l_)
                accs.append(a_)
            return sum(losses) / len(losses), sum(accs) / len(accs)
        return [ops.upsample_softmax(p, segSize) for p in preds]


class Non_local2d(nn.Module):
    def __init__(self, num_class=None, downsample=False):
        super().__init__()
        self.downsample = downsample
        self.emb = vnn.Conv2d(2048, 256, 1, 1)
        self.nonlocalblock = NLBlockND(in_channels=256, mode="dot", dimension=2, bn_layer=True)
        self.last_layer = vnn.Conv2d(512, num_class, kernel_size=1, stride=1)

    def forward(self, input, segSize=None):
        emb = self.emb(input[-1])
        if self.downsample:  # reference :135-138
            x = ops.interpolate_bilinear(self.nonlocalblock(ops.avg_pool2x2(emb)), emb.shape[-2:])
        else:
            x = self.nonlocalblock(emb)
        pred = self.last_layer(ops.channel_cat([emb, x]))
        if segSize is None:
            return ops.log_softmax_channels(pred)
        return ops.upsample_softmax(pred, segSize)
