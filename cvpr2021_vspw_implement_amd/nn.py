"""nn.Module building blocks whose forward runs on the HIP kernels (ops.py).

They subclass the torch modules the reference uses (nn.Conv2d, _BatchNorm) so constructor signatures, default
initialisation, attribute names (`stride`, `dilation`, `padding` are rewritten in place by ResnetDilated, reference
models/models.py:737-750) and state_dict keys are the reference's; only `forward` differs.
"""
import torch
import torch.nn as nn

from . import ops


def _sq(v, what):
    a, b = (v, v) if isinstance(v, int) else (v[0], v[1])
    if a != b:
        raise NotImplementedError("non-square %s %s is not on the VSPW hot path" % (what, v))
    return int(a)


class Conv2d(nn.Conv2d):
    """nn.Conv2d with the weight held in channels_last memory ([Cout][KH][KW][Cin], the kernels' layout)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("grouped / non-zero-padded convolutions are not on the VSPW hot path")
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def geometry(self):
        return _sq(self.stride, "stride"), _sq(self.padding, "padding"), _sq(self.dilation, "dilation")

    def forward(self, x):
        s, p, d = self.geometry()
        return ops.conv2d(x, self.weight, self.bias, s, p, d)


class Conv3d1x1(nn.Conv3d):
    """nn.Conv3d(kernel_size=1) (NLBlockND with dimension=3): a 1x1 conv over [B,C,T,H,W] folded to [B,C,T*H,W]."""

    def forward(self, x):
        if self.kernel_size != (1, 1, 1):
            raise NotImplementedError("only 1x1x1 Conv3d is on the VSPW hot path")
        b, c, t, h, w = x.shape
        y = ops.conv2d(x.reshape(b, c, t * h, w), self.weight.reshape(self.out_channels, c, 1, 1), self.bias, 1, 0, 1)
        return y.reshape(b, self.out_channels, t, h, w)


class _SynchronizedBatchNorm(nn.modules.batchnorm._BatchNorm):
    """SynchronizedBatchNorm{1,2,3}d of models/sync_batchnorm/batchnorm.py: F.batch_norm semantics on one device,
    statistics all-reduced across ranks when ops.set_sync_bn(True) (one process per GPU replaces DataParallel)."""

    def _check_input_dim(self, input):
        pass

    def bn_args(self):
        return (self.weight, self.bias, self.running_mean, self.running_var)

    def forward(self, x, residual=None, relu=False, mask=None):
        shape = x.shape
        if x.dim() == 5:  # [B,C,T,H,W] -> [B,C,T*H,W]
            x = x.reshape(shape[0], shape[1], shape[2] * shape[3], shape[4])
        elif x.dim() == 3:
            x = x.unsqueeze(-1)
        elif x.dim() == 2:
            x = x.unsqueeze(-1).unsqueeze(-1)
        y = ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, residual, mask,
                               self.training, self.momentum, self.eps, relu)
        return y.reshape(shape)


class SynchronizedBatchNorm1d(_SynchronizedBatchNorm):
    pass


class SynchronizedBatchNorm2d(_SynchronizedBatchNorm):
    pass


class SynchronizedBatchNorm3d(_SynchronizedBatchNorm):
    pass


BatchNorm2d = SynchronizedBatchNorm2d


def dropout2d_mask(module, x_shape, device):
    """Per-(image, channel) keep mask scaled by 1/(1-p) for nn.Dropout2d in training mode, else None.
    `module._forced_mask` (tests) overrides the RNG draw."""
    if module is None or not module.training or module.p == 0.0:
        return None
    forced = getattr(module, "_forced_mask", None)
    if forced is not None:
        return forced.to(device=device, dtype=torch.float32).contiguous()
    n, c = x_shape[0], x_shape[1]
    keep = 1.0 - module.p
    return (torch.rand((n, c), device=device) < keep).float().div_(keep)


def conv_bn_act(x, conv, bn, relu=True, residual=None, dropout=None, skip_out=False, fuse_input=False,
                defer_apply=False):
    """conv -> bn -> (+residual) -> relu -> dropout2d as one fused autograd node.  skip_out=True returns (out, x'):
    feed x' to the block's skip path and its gradient is added inside this conv's data-gradient GEMM.
    fuse_input=True asserts that this conv is the only consumer of x (see ops.BNLink)."""
    s, p, d = conv.geometry()
    mask = dropout2d_mask(dropout, (x.shape[0], conv.out_channels), x.device)
    return ops.conv_bn_act(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual,
                           mask, s, p, d, bn.training, bn.momentum, bn.eps, relu, skip_out, fuse_input, defer_apply)


class AdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    def forward(self, x):
        s = self.output_size
        if isinstance(s, (tuple, list)):
            if s[0] != s[1]:
                raise NotImplementedError("non-square adaptive pooling")
            s = s[0]
        return ops.pyramid_pool(x, (int(s),), 1, None)[0]


class FusedSequential(nn.Sequential):
    """nn.Sequential (same child indices, hence same state_dict keys) whose forward fuses
    Conv2d -> BatchNorm -> ReLU -> Dropout2d runs into single kernels chains."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, Conv2d) and isinstance(nxt, _SynchronizedBatchNorm):
                j = i + 2
                relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
                if relu:
                    j += 1
                drop = mods[j] if j < len(mods) and isinstance(mods[j], nn.Dropout2d) else None
                if drop is not None:
                    j += 1
                x = conv_bn_act(x, m, nxt, relu=relu, dropout=drop)
                i = j
            elif isinstance(m, _SynchronizedBatchNorm):
                j = i + 1
                relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
                if relu:
                    j += 1
                drop = mods[j] if j < len(mods) and isinstance(mods[j], nn.Dropout2d) else None
                if drop is not None:
                    j += 1
                mask = dropout2d_mask(drop, x.shape, x.device)
                x = m(x, relu=relu, mask=mask)
                i = j
            elif isinstance(m, (nn.ReLU, nn.Dropout2d)):
                raise NotImplementedError("bare %s outside a BatchNorm chain is not on the VSPW hot path"
                                          % type(m).__name__)
            else:
                x = m(x)
                i += 1
        return x
