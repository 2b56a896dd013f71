"""NetWarp (optical-flow feature warping), mirroring reference models/netwarp.py:12-239.

`flowwarp` and the per-channel blend run on the HIP kernels (csrc/misc.hip).  RAFT itself is the frozen flow provider
(SURVEY.md §2 "◐", §8f rank 1): it stays on stock PyTorch-ROCm ops and is NOT part of this package — NetWarp takes
it from `args.flow_net` (any nn.Module with RAFT's forward(img1, img2, iters, test_mode) -> (low, up) signature), or
imports the reference's own RAFT_core when that package and its checkpoint are on the path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import nn as vnn
from .. import ops
from .lr_groups import LrGroupsMixin
from .models import conv3x3_bn_relu, nll_ignore_index

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
    try:
        from RAFT_core.raft import RAFT  # the reference's vendored RAFT (stock PyTorch ops), if on sys.path
    except Exception as e:  # pragma: no cover - depends on the deployment
        raise NotImplementedError(
            "NetWarp needs an optical-flow network: pass args.flow_net or put the reference's RAFT_core (with "
            "raft-things.pth-no-zip) on sys.path (%s)" % (e,))
    from collections import OrderedDict

    raft = RAFT()
    to_load = torch.load("./RAFT_core/raft-things.pth-no-zip")
    raft.load_state_dict(OrderedDict((k[7:], v) for k, v in to_load.items()))
    return raft


def _pad_to_8(x):
    """RAFT's InputPadder('sintel') (RAFT_core/utils/utils.py:7-25): replicate-pad H,W up to multiples of 8."""
    h, w = x.shape[-2:]
    ph, pw = (((h // 8) + 1) * 8 - h) % 8, (((w // 8) + 1) * 8 - w) % 8
    pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    return F.pad(x, pad, mode="replicate"), pad


class NetWarp(LrGroupsMixin, nn.Module):
    def __init__(self, net_enc, net_dec, crit, args, deep_sup_scale=None):
        super().__init__()
        self.raft = _load_flow_net(args)
        self.mean = torch.FloatTensor([0.485, 0.456, 0.406])
        self.std = torch.FloatTensor([0.229, 0.224, 0.225])
        self.encoder = net_enc
        self.decoder = net_dec
        self.crit = crit
        self.deep_sup_scale = deep_sup_scale
        self.args = args
        assert self.args.clip_num == 2
        self.flowcnn = FlowCNN()
        self.conv_last_ = vnn.FusedSequential(
            vnn.Conv2d(2048 + 4 * 512, 512, kernel_size=3, padding=1, bias=False),
            BatchNorm2d(512),
            nn.ReLU(inplace=True),
            nn.Dropout2d(0.1),
            vnn.Conv2d(512, args.num_class, kernel_size=1),
        )
        self.w0_0 = nn.Parameter(torch.ones(2048))
        self.w0_1 = nn.Parameter(torch.zeros(2048))
        self.w1_0 = nn.Parameter(torch.ones(4096))
        self.w1_1 = nn.Parameter(torch.zeros(4096))

    def _lr_10x_roots(self):
        return [self.decoder, self.flowcnn, self.conv_last_]

    def pixel_acc(self, pred, label):
        _, preds = torch.max(pred, dim=1)
        valid = (label >= 0).long()
        acc_sum = torch.sum(valid * (preds == label).long())
        pixel_sum = torch.sum(valid)
        return acc_sum.float() / (pixel_sum.float() + 1e-10)

    def _flow(self, cur255, prev255):
        with torch.no_grad():
            self.raft.eval()
            a, pad = _pad_to_8(cur255)
            b, _ = _pad_to_8(prev255)
            _, flow = self.raft(a, b, iters=20, test_mode=True)
            hh, ww = flow.shape[-2:]
            return flow[..., pad[2]:hh - pad[3], pad[0]:ww - pad[1]].contiguous()

    def forward(self, feed_dict, *, segSize=None):
        if feed_dict is None:
            return torch.zeros((0, self.args.num_class, 480, 720)).cuda()
        c_img = feed_dict["img_data"]
        clip_imgs = feed_dict["clipimgs_data"]
        label = feed_dict["seg_label"]
        assert len(clip_imgs) == 1
        c_pre_img = clip_imgs[0]
        mean = self.mean.to(c_img.device).view(1, 3, 1, 1)
        std = self.std.to(c_img.device).view(1, 3, 1, 1)
        c_img_f = (c_img * std + mean) * 255.0  # image un-normalisation: input plumbing for the flow net
        c_pre_img_f = (c_pre_img * std + mean) * 255.0
        flow = feed_dict["flow"] if "flow" in feed_dict else self._flow(c_img_f, c_pre_img_f)
        flow = self.flowcnn(c_img_f, c_pre_img_f, flow)

        feats = self.encoder(torch.cat([c_img, c_pre_img], 0), return_feature_maps=True)
        B = c_img.shape[0]
        conv5 = feats[-1]
        cur1, prev1 = conv5[:B], conv5[B:]
        flow_1 = F.interpolate(flow, cur1.shape[-2:], mode="nearest")  # nearest, magnitudes NOT rescaled (quirk)
        new_cur1 = ops.chan_blend(cur1, ops.flowwarp(prev1, flow_1), self.w0_0, self.w0_1)
        feats[-1] = torch.cat([new_cur1, prev1], 0)
        pred_deepsup_s, _, ppm_cat = self.decoder(feats)
        cur2, prev2 = ppm_cat[:B], ppm_cat[B:]
        flow_2 = F.interpolate(flow, cur2.shape[-2:], mode="nearest")
        new_feat = ops.chan_blend(cur2, ops.flowwarp(prev2, flow_2), self.w1_0, self.w1_1)
        pred_ = self.conv_last_(new_feat)
        if segSize is not None:
            return ops.upsample_softmax(pred_, segSize)
        ignore = nll_ignore_index(self.crit)
        loss, acc = ops.seg_nll(pred_, label, ignore, want_acc=True, from_logits=True)
        if self.deep_sup_scale is not None:
            loss_deepsup, _ = ops.seg_nll(pred_deepsup_s[:B], label, ignore, want_acc=False, from_logits=False)
            loss = loss + loss_deepsup * self.deep_sup_scale
        return loss, acc
