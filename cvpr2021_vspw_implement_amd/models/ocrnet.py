"""Per-frame OCRNet decoder (`ocrnet_deepsup`), mirroring reference models/ocrnet.py:22-72."""
import torch.nn as nn

from .. import nn as vnn
from .. import ops
from .ocr_modules.spatial_ocr_block import SpatialGather_Module, SpatialOCR_Module

BatchNorm2d = vnn.SynchronizedBatchNorm2d


def ocr_heads(num_classes, in_channels=(1024, 2048)):
    conv_3x3 = vnn.FusedSequential(
        vnn.Conv2d(in_channels[1], 512, kernel_size=3, stride=1, padding=1),
        BatchNorm2d(512),
        nn.ReLU(inplace=True),
    )
    head = vnn.Conv2d(512, num_classes, kernel_size=1, stride=1, padding=0, bias=True)
    dsn_head = vnn.FusedSequential(
        vnn.Conv2d(in_channels[0], 512, kernel_size=3, stride=1, padding=1),
        BatchNorm2d(512),
        nn.ReLU(inplace=True),
        nn.Dropout2d(0.05),
        vnn.Conv2d(512, num_classes, kernel_size=1, stride=1, padding=0, bias=True),
    )
    return conv_3x3, head, dsn_head


class SpatialOCRNet(nn.Module):
    def __init__(self, num_class):
        self.inplanes = 128
        super().__init__()
        self.num_classes = num_class
        self.conv_3x3, head, dsn_head = ocr_heads(num_class)
        self.spatial_context_head = SpatialGather_Module(self.num_classes)
        self.spatial_ocr_head = SpatialOCR_Module(in_channels=512, key_channels=256, out_channels=512, scale=1,
                                                  dropout=0.05)
        self.head = head
        self.dsn_head = dsn_head

    def forward(self, x, segSize=None):
        x_dsn = self.dsn_head(x[-2])
        x = self.conv_3x3(x[-1])
        context = self.spatial_context_head(x, x_dsn)
        x = self.head(self.spatial_ocr_head(x, context))
        if segSize is not None:
            return ops.upsample_softmax(x, segSize)
        return ops.log_softmax_channels(x), ops.log_softmax_channels(x_dsn)
