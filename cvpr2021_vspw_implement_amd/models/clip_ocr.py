"""TCB-OCR: OCRNet head whose object contexts are blended (mean) over the T frames of a clip.

Mirrors reference models/clip_ocr.py:23-198 (ClipOCRNet): constructor, feed_dict protocol, `args` flags
(num_class, use_memory, memory_num, clipocr_all), state_dict keys (conv_3x3.*, spatial_ocr_head.*, head.*, dsn_head.*)
and outputs.
"""
import torch
import torch.nn as nn

from .. import ops
from .lr_groups import LrGroupsMixin
from .models import nll_ignore_index
from .ocr_modules.spatial_ocr_block import SpatialOCR_Module, SpatialTemporalGather_Module
from .ocrnet import ocr_heads
from ._metrics import pixel_accuracy


class ClipOCRNet(LrGroupsMixin, nn.Module):
    def __init__(self, net_enc, crit, args, deep_sup_scale=None):
        super().__init__()
        self.args = args
        if self.args.use_memory:
            self.memory = []
        self.crit = crit
        self.deep_sup_scale = deep_sup_scale
        self.encoder = net_enc
        self.inplanes = 128
        self.num_classes = args.num_class
        self.conv_3x3, head, dsn_head = ocr_heads(self.num_classes)
        self.spatial_context_head = SpatialTemporalGather_Module(self.num_classes)
        self.spatial_ocr_head = SpatialOCR_Module(in_channels=512, key_channels=256, out_channels=512, scale=1,
                                                  dropout=0.05)
        self.head = head
        self.dsn_head = dsn_head

    def _lr_10x_roots(self):
        return [self.conv_3x3, self.spatial_context_head, self.spatial_ocr_head, self.head, self.dsn_head]

    def pixel_acc(self, pred, label):
        return pixel_accuracy(pred, label)

    def forward(self, feed_dict, segSize=None):
        c_img = feed_dict["img_data"]
        clip_imgs = feed_dict["clipimgs_data"]
        label = feed_dict["seg_label"]
        clip_num = len(clip_imgs)
        T = clip_num + 1
        clip_imgs.append(c_img)
        frames = torch.cat(clip_imgs, dim=0)
        feats = self.encoder(frames, return_feature_maps=True)

        x_dsn = self.dsn_head(feats[-2])
        out_tmp = self.conv_3x3(feats[-1])
        if segSize is not None and self.args.use_memory:
            if feed_dict["is_clean_memory"]:
                self.memory = []
            context = self.spatial_context_head(out_tmp, x_dsn, clip_num, self.memory, self.args.memory_num)
        else:
            context = self.spatial_context_head(out_tmp, x_dsn, clip_num)
        B = out_tmp.shape[0] // T

        all_frames = bool(self.args.clipocr_all)
        if all_frames and T > 1:
            # the reference cannot run this flag: spatial_ocr_head(out_tmp, context) pairs B*T frames of pixels with B
            # object contexts and _ObjectAttentionBlock's view() rejects it (clip_ocr.py:136-137,
            # spatial_ocr_block.py:263: "shape '[B*T, 256, -1]' is invalid for input of size ..."); same error class here
            raise RuntimeError("clipocr_all with %d frames per clip: shape '[%d, 256, -1]' is invalid for the %d object "
                               "contexts (the reference fails identically)" % (T, out_tmp.shape[0], B))
        x = out_tmp if all_frames else ops.tail_frames(out_tmp, B)
        x = self.head(self.spatial_ocr_head(x, context))
        if segSize is not None:
            if all_frames:
                x = x[(T - 1) * B:]
            return ops.upsample_softmax(x, segSize)

        ignore = nll_ignore_index(self.crit)
        clip_labels = feed_dict["cliplabels_data"]
        clip_labels.append(label)
        alllabel = torch.cat(clip_labels, dim=0)
        loss, acc = ops.seg_nll(x, alllabel if all_frames else label, ignore, want_acc=True, from_logits=True)
        loss_deepsup, _ = ops.seg_nll(x_dsn, alllabel, ignore, want_acc=False, from_logits=True)
        loss = loss + loss_deepsup * self.deep_sup_scale
        return loss, acc
