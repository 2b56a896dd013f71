"""Deep-stem ResNet-18/50/101 backbones on the HIP kernels.

Same constructor surface, attribute names and state_dict keys as the reference's models/resnet.py:95-205
(3x conv3x3 stem 3->64(s2)->64->128, max-pool, four stages, avgpool/fc_1 kept for key parity), but every block is
executed as fused conv+BN(+residual)+ReLU kernel chains instead of separate ATen ops.
"""
import math
import os

import torch
import torch.nn as nn

from .. import nn as vnn
from .. import ops

__all__ = ["ResNet", "resnet18", "resnet50", "resnet101", "BasicBlock", "Bottleneck"]

BatchNorm2d = vnn.SynchronizedBatchNorm2d


def conv3x3(in_planes, out_planes, stride=1):
    return vnn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def _residual_branch(block, x):
    if block.downsample is None:
        return x
    return vnn.conv_bn_act(x, block.downsample[0], block.downsample[1], relu=False)


# conv2 -> conv3 hand-over of bn2 + ReLU (same mechanism as the block-output hand-over, no residual stream).  Measured
# +-0 on the bench step (102.51 vs 102.58 ms: the 19 us bn_apply pass it removes costs as much as the 8-fold
# re-evaluation in conv3's eight column tiles), so it is off by default; bit-identical either way (tests).
_DEFER_CONV2 = os.environ.get("VSPW_FWD_APPLY_CONV2", "0") == "1"
# conv1's BatchNorm apply + ReLU evaluated by conv2's Winograd input transform (ops._wino_takes_pending; anything else
# that conv2 turns out to be materialises it first).  Measured on the bench step: +-0 (83.2-83.4 ms either way - the
# transform, which touches every pixel four times, slows down by what the 20 us apply pass cost): off by default
_DEFER_CONV1 = os.environ.get("VSPW_FWD_APPLY_CONV1", "0") == "1"


class _BlockSequential(nn.Sequential):
    """nn.Sequential of residual blocks (same child names, hence the reference's state_dict keys).  Block i's output is
    read by block i+1 and nothing else, which is what Bottleneck.forward(sole_consumer=True) needs to know; the first
    block's input and the last block's output may have other readers (feature maps handed to the decoders)."""

    def forward(self, x):
        blocks = list(self)
        for i, blk in enumerate(blocks):
            if isinstance(blk, Bottleneck):
                # the next bottleneck's conv1 (pointwise) may evaluate this block's final BN + skip + ReLU itself
                # (not when a forward hook would look at the block's output before that conv1 has written it)
                nxt = (i + 1 < len(blocks) and isinstance(blocks[i + 1], Bottleneck) and not blk._forward_hooks
                       and not blocks[i + 1]._forward_pre_hooks)
                x = blk(x, sole_consumer=i > 0, defer_output=nxt)
            else:
                x = blk(x)
        return ops.materialize(x)


class BasicBlock(nn.Module):
    """reference models/resnet.py:24-53"""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = vnn.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        return vnn.conv_bn_act(out, self.conv2, self.bn2, relu=True, residual=_residual_branch(self, x))


class Bottleneck(nn.Module):
    """reference models/resnet.py:56-92 (stride sits on the 3x3)"""

    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = vnn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = vnn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = vnn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, sole_consumer=False, defer_output=False):
        """sole_consumer: nothing but this block reads x (true for every block but the first of a layer, see
        _BlockSequential) - lets conv1's data gradient carry the previous block's batch-norm backward reductions."""
        # the skip connection is taken from conv1's node (skip_out): in backward its gradient is added in conv1's
        # data-gradient epilogue instead of by a separate accumulation pass over the block input
        skip = x
        if x.requires_grad and torch.is_grad_enabled():
            out, skip = vnn.conv_bn_act(x, self.conv1, self.bn1, relu=True, skip_out=True, fuse_input=sole_consumer,
                                        defer_apply=_DEFER_CONV1)
        else:
            out = vnn.conv_bn_act(x, self.conv1, self.bn1, relu=True, defer_apply=_DEFER_CONV1)
        # conv1's output is only read by conv2, conv2's only by conv3 (pointwise: it may evaluate bn2 + ReLU itself)
        # (a block late in its stage may ask for a larger Winograd tile: see ResnetDilated / ops.winograd_tile_hint)
        with ops.winograd_tile_hint(getattr(self, "_vspw_wino_tile", 0)):
            out = vnn.conv_bn_act(out, self.conv2, self.bn2, relu=True, fuse_input=True, defer_apply=_DEFER_CONV2)
        return vnn.conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=_residual_branch(self, skip),
                               fuse_input=True, defer_apply=defer_output)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=146):
        self.inplanes = 128
        super().__init__()
        self.conv1 = conv3x3(3, 64, stride=2)
        self.bn1 = BatchNorm2d(64)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(64, 64)
        self.bn2 = BatchNorm2d(64)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv3 = conv3x3(64, 128)
        self.bn3 = BatchNorm2d(128)
        self.relu3 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc_1 = nn.Linear(512 * block.expansion, num_classes)

        for m in self.modules():  # reference models/resnet.py:118-124
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                vnn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return _BlockSequential(*layers)

    def stem(self, x):
        x = vnn.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        x = vnn.conv_bn_act(x, self.conv2, self.bn2, relu=True, fuse_input=True)  # each stem output has one reader
        x = vnn.conv_bn_act(x, self.conv3, self.bn3, relu=True, fuse_input=True)
        return ops.max_pool3x3s2(x)

    def forward(self, x):
        raise NotImplementedError("the ImageNet classifier head (avgpool + fc_1) is not on the VSPW hot path")


def resnet18(pretrained=False, **kwargs):
    return ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet50(pretrained=False, **kwargs):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)


def resnet101(pretrained=False, **kwargs):
    return ResNet(Bottleneck, [3, 4, 23, 3], **kwargs)
