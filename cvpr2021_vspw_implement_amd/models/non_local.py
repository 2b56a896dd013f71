"""Non-local block (mode 'dot') on the HIP kernels, mirroring reference models/non_local.py:7-151.

y = (theta^T phi / N) g is evaluated by ONE fused kernel (csrc/nonlocal.hip, ops.non_local_dot) over the [B, N, C]
pixel-row matrices that NHWC memory provides for free (N = H*W, or T*H*W for the spatio-temporal block): 32-key tiles of
phi / g stream through LDS, the affinity tile lives in MFMA accumulators and feeds the second product directly, so the
N x N affinity (2.5 GB per sample at T = 7) never exists in memory - forward or backward.
"""
import torch.nn as nn

from .. import nn as vnn
from .. import ops


class NLBlockND(nn.Module):
    def __init__(self, in_channels, inter_channels=None, mode="embedded", dimension=3, bn_layer=True):
        super().__init__()
        assert dimension in [1, 2, 3]
        if mode not in ["gaussian", "embedded", "dot", "concatenate"]:
            raise ValueError("`mode` must be one of `gaussian`, `embedded`, `dot` or `concatenate`")
        if mode != "dot":
            raise NotImplementedError("only mode='dot' (the one the VSPW heads use) is on the HIP path")
        if not bn_layer or dimension == 1:
            raise NotImplementedError("only the bn_layer=True, 2-D / 3-D block of the VSPW heads is on the HIP path")
        self.mode = mode
        self.dimension = dimension
        self.in_channels = in_channels
        self.inter_channels = inter_channels
        if self.inter_channels is None:
            self.inter_channels = max(in_channels // 2, 1)
        if dimension == 3:
            conv_nd = lambda cin, cout: vnn.Conv3d1x1(in_channels=cin, out_channels=cout, kernel_size=1)  # noqa: E731
        else:
            conv_nd = lambda cin, cout: vnn.Conv2d(in_channels=cin, out_channels=cout, kernel_size=1)  # noqa: E731
        # the reference instantiates SynchronizedBatchNorm3d for both the 2-D and the 3-D block (non_local.py:42-46)
        bn = vnn.SynchronizedBatchNorm3d
        self.g = conv_nd(self.in_channels, self.inter_channels)
        self.W_z = nn.Sequential(conv_nd(self.inter_channels, self.in_channels), bn(self.in_channels))
        nn.init.constant_(self.W_z[1].weight, 0)
        nn.init.constant_(self.W_z[1].bias, 0)
        self.theta = conv_nd(self.in_channels, self.inter_channels)
        self.phi = conv_nd(self.in_channels, self.inter_channels)

    @staticmethod
    def _w2d(conv):
        return conv.weight.reshape(conv.out_channels, conv.in_channels, 1, 1)

    def forward(self, x):
        shape = x.shape
        b, c = shape[0], shape[1]
        n_pos = 1
        for s in shape[2:]:
            n_pos *= s
        xr = ops.as_nhwc(x.reshape(b, c, n_pos, 1))  # [B,C,N,1]: rows = positions
        g = ops.pixels_view(ops.conv2d(xr, self._w2d(self.g), self.g.bias))  # [B,N,Ci]
        th = ops.pixels_view(ops.conv2d(xr, self._w2d(self.theta), self.theta.bias))
        ph = ops.pixels_view(ops.conv2d(xr, self._w2d(self.phi), self.phi.bias))
        if ops.nl_dot_supported(self.inter_channels):
            # fused: theta phi^T is streamed through registers, never written (csrc/nonlocal.hip); f / N happens on the
            # affinity tile, where the reference applies it
            y = ops.from_pixels(ops.non_local_dot(th, ph, g, 1.0 / n_pos), n_pos, 1)  # [B,Ci,N,1]
        else:  # other channel counts: two dense GEMMs over a materialised [B,N,N] affinity
            f = ops.bmm_nt(th, ph)  # [B,N,N]
            gT = ops.transpose_last2(ops.scale(g, 1.0 / n_pos))  # [B,Ci,N]
            y = ops.from_pixels(ops.bmm_nt(f, gT), n_pos, 1)  # [B,Ci,N,1]
        wz, bnz = self.W_z[0], self.W_z[1]
        z = ops.conv_bn_act(y, self._w2d(wz), wz.bias, bnz.weight, bnz.bias, bnz.running_mean, bnz.running_var,
                            residual=xr, mask=None, stride=1, pad=0, dil=1, training=bnz.training,
                            momentum=bnz.momentum, eps=bnz.eps, relu=False)
        return z.reshape(shape)
