"""RAFT (basic, frozen) on the HIP kernels: the flow network NetWarp runs before its warps.

Mirror of the reference's RAFT_core/raft.py:26-127 as models/netwarp.py:71-77,170-176 uses it: `RAFT()` with the
state_dict keys of `raft-things.pth` (after the `module.` prefix is stripped), eval mode, no gradients,
`forward(image1, image2, iters=20, test_mode=True) -> (flow_low [N,2,H/8,W/8], flow_up [N,2,H,W])`, images in [0, 255].

The torch modules below only HOLD the parameters under the reference's names; the forward pass is a fixed schedule of
C-ABI launches (csrc/raft.hip + vspw_conv2d_fwd_ex) on NHWC buffers:
  * every torch.cat of the update block (update.py:24,29,44,47,94,129) is a channel slot of one [pixels][384] buffer,
    written in place by the producing convolution (row stride ldy) and read as a strided slice (ldx) by the consumers;
  * the z and r gates of each GRU half share one convolution (filters stacked along Cout, sigmoid in the epilogue);
  * conv + bias + relu/sigmoid/tanh are one launch; norm -> relu -> (+skip) -> relu of the encoders are one launch;
  * the convex-upsampling mask is only evaluated for the last iteration (test_mode returns only that one).
"""
import ctypes
import os
import math

import torch
import torch.nn as nn

from .. import _C
from ..ops import ConvDesc, _p, _require_gpu, _stream, _ws

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3


# ------------------------------------------------------------------------------------------- parameter containers
# the update block's Winograd convolutions through the row-fused GEMM where the grid fills the chip (see conv() below)
_RAFT_WROWS = os.environ.get("VSPW_RAFT_WROWS", "1") == "1"
_RAFT_PADK = os.environ.get("VSPW_RAFT_PADK", "1") == "1"  # convc1's 324-long reduction zero-padded to 352 (a multiple of 32)
_RAFT_THIN = os.environ.get("VSPW_RAFT_THIN", "1") == "1"  # vspw_conv2d_thin for the 256 -> 2 convolution of the flow head

class _ResidualBlock(nn.Module):
    """RAFT_core/extractor.py:6-56 (norm_fn 'instance' or 'batch'); norm3 is also downsample[1], as there."""

    def __init__(self, in_planes, planes, norm_fn, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        mk = (lambda: nn.BatchNorm2d(planes)) if norm_fn == "batch" else (lambda: nn.InstanceNorm2d(planes))
        self.norm1, self.norm2 = mk(), mk()
        self.stride = stride
        if stride == 1:
            self.downsample = None
        else:
            self.norm3 = mk()
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)


class _BasicEncoder(nn.Module):
    """RAFT_core/extractor.py:116-190 (dropout 0)."""

    def __init__(self, output_dim, norm_fn):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = nn.BatchNorm2d(64) if norm_fn == "batch" else nn.InstanceNorm2d(64)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3)
        self.layer1 = nn.Sequential(_ResidualBlock(64, 64, norm_fn, 1), _ResidualBlock(64, 64, norm_fn, 1))
        self.layer2 = nn.Sequential(_ResidualBlock(64, 96, norm_fn, 2), _ResidualBlock(96, 96, norm_fn, 1))
        self.layer3 = nn.Sequential(_ResidualBlock(96, 128, norm_fn, 2), _ResidualBlock(128, 128, norm_fn, 1))
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class _FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)


class _SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=256):
        super().__init__()
        c = hidden_dim + input_dim
        self.convz1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))


class _BasicMotionEncoder(nn.Module):
    def __init__(self, corr_levels, corr_radius):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)


class _BasicUpdateBlock(nn.Module):
    def __init__(self, corr_levels, corr_radius, hidden_dim=128):
        super().__init__()
        self.encoder = _BasicMotionEncoder(corr_levels, corr_radius)
        self.gru = _SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = _FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 64 * 9, 1))


# ------------------------------------------------------------------------------------------------ launch helpers
_THIN_OK = {}
GEMM_FLOPS = {"total": 0.0}  # algorithmic FLOPs of the MFMA launches issued so far (tools/raft_bench.py reads / resets it)


def _conv(x, n, h, w, c, ldx, wt, bias, kh, kw, stride, pad, act, y, ldy):
    """One vspw_conv2d_fwd_ex launch.  x: device pointer holder of an NHWC buffer [n][h][w][ldx] read from its first
    `c` channels on; wt [K][KH][KW][C] packed; y written with row stride ldy.  Returns (oh, ow)."""
    ph, pw = pad
    k = wt.shape[0]
    oh = (h + 2 * ph - (kh - 1) - 1) // stride + 1
    ow = (w + 2 * pw - (kw - 1) - 1) // stride + 1
    d = ConvDesc(n, h, w, c, oh, ow, k, kh, kw, stride, ph, 1, pw)
    GEMM_FLOPS["total"] += 2.0 * n * oh * ow * k * kh * kw * c
    thin_key = (n, h, w, c, k, kh, kw, stride, ph, pw, ldx, ldy)
    thin = _THIN_OK.get(thin_key)
    if thin is None:  # one query per geometry, not one per launch (20 iterations x every convolution of the update block)
        thin = _THIN_OK[thin_key] = bool(_RAFT_THIN and int(_C.query("vspw_conv2d_thin_supported", ctypes.byref(d), ldx, ldy)))
    if thin and ((x.value or 0) | wt.data_ptr()) % 16 == 0:  # (the kernel loads float4s: 16-byte aligned slots only)
        # 256 -> 2 channels: a direct kernel instead of an MFMA tile that is 97 % padding
        _C.call("vspw_conv2d_thin", ctypes.byref(d), x, ldx, _p(wt), _p(bias), act, y, ldy, _stream())
        return oh, ow
    _C.call("vspw_conv2d_fwd_ex", ctypes.byref(d), x, ldx, _p(wt), _p(bias), None, act, y, ldy, _stream())
    return oh, ow


def _off(t, elems):
    """Device pointer `elems` floats into tensor t (a channel slot of an NHWC buffer)."""
    return ctypes.c_void_p(t.data_ptr() + 4 * int(elems))


def _pack(conv):
    """nn.Conv2d weight [K][C][KH][KW] -> [K][KH][KW][C] contiguous (the kernels' layout)."""
    return conv.weight.detach().permute(0, 2, 3, 1).contiguous()


class RAFT(nn.Module):
    """RAFT_core/raft.py:26-127."""

    def __init__(self, requires_grad=False):
        super().__init__()
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        self.corr_levels = 4
        self.corr_radius = 4
        self.fnet = _BasicEncoder(output_dim=256, norm_fn="instance")
        self.cnet = _BasicEncoder(output_dim=hdim + cdim, norm_fn="batch")
        self.update_block = _BasicUpdateBlock(self.corr_levels, self.corr_radius, hidden_dim=hdim)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False
        self._packed = None

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    # ------------------------------------------------------------------------------------------ weight packing
    def _pack_all(self):
        """Kernel-layout copies of the (frozen) weights, rebuilt when any parameter changes version or device."""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + tuple(
            (b.data_ptr(), b._version) for b in self.buffers())
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        P = {}
        for name, m in self.named_modules(remove_duplicate=False):  # norm3 is also downsample.1
            if isinstance(m, nn.Conv2d):
                P[name] = (_pack(m), m.bias.detach().contiguous())
            elif isinstance(m, nn.BatchNorm2d):
                c = m.num_features
                dev = m.weight.device
                buf = torch.empty((4, c), device=dev, dtype=torch.float32)
                _C.call("vspw_bn_eval_coeffs", _p(m.weight), _p(m.bias), _p(m.running_mean), _p(m.running_var),
                        float(m.eps), _p(buf[0]), _p(buf[1]), _p(buf[2]), _p(buf[3]), c, _stream())
                P[name] = (buf[2], buf[3])  # scale, shift
        g = self.update_block.gru
        for sfx in ("1", "2"):
            z, r = getattr(g, "convz" + sfx), getattr(g, "convr" + sfx)
            P["update_block.gru.convzr" + sfx] = (torch.cat([_pack(z), _pack(r)], 0).contiguous(),
                                                  torch.cat([z.bias.detach(), r.bias.detach()], 0).contiguous())
        # convc1 reduces over the 324 correlation taps: 324 % 32 != 0 sends it to the generic gather kernel (31 us per
        # iteration).  Zero-padded to 352 columns (the lookup buffer carries 28 zero columns) it is a plain pointwise GEMM.
        if _RAFT_PADK:
            wc, bc = P["update_block.encoder.convc1"]
            wp = torch.zeros((wc.shape[0], 1, 1, 352), device=wc.device, dtype=torch.float32)
            wp[..., : wc.shape[3]] = wc
            P["update_block.encoder.convc1"] = (wp.contiguous(), bc)
        c2 = self.cnet.conv2
        w2, b2 = _pack(c2), c2.bias.detach()
        P["cnet.conv2.net"] = (w2[: self.hidden_dim].contiguous(), b2[: self.hidden_dim].contiguous())
        P["cnet.conv2.inp"] = (w2[self.hidden_dim:].contiguous(), b2[self.hidden_dim:].contiguous())
        # Winograd F(2x2,3x3) weights (csrc/winograd.hip) of the update block's wide stride-1 3x3 convolutions: 38 % of
        # an iteration's FLOPs at 4/9 of the multiplications.  `encoder.conv` has 126 outputs: two zero rows pad it to 128
        # (the two surplus channels land in HX's flow slot, which is rewritten right after, update.py:95-96)
        import os

        if os.environ.get("VSPW_RAFT_WINOGRAD", "1") == "1":
            for name, pad_k in (("update_block.encoder.convc2", 0), ("update_block.encoder.conv", 2),
                                ("update_block.flow_head.conv1", 0), ("update_block.mask.0", 0)):
                w, b = P[name]
                if pad_k:
                    w = torch.cat([w, torch.zeros((pad_k,) + tuple(w.shape[1:]), device=w.device)], 0).contiguous()
                    b = torch.cat([b, torch.zeros(pad_k, device=b.device)], 0).contiguous()
                k, c = w.shape[0], w.shape[3]
                u = torch.empty((16, k, c), device=w.device, dtype=torch.float32)
                _C.call("vspw_wino_weights", _p(w), _p(u), k, c, 0, _stream())
                P["wino:" + name] = (u, b, k, c)
        self._packed = (key, P)
        return P

    # ------------------------------------------------------------------------------------------------ encoders
    def _norm_act(self, P, enc, name, x, n, hw, c, residual, relu_in, relu_out, y):
        if enc.norm_fn == "batch":
            scale, shift = P[name]
            stride = 0
        else:
            coef = torch.empty((2, n, c), device=x.device, dtype=torch.float32)
            nbytes = _C.query("vspw_instance_norm_workspace", n, hw, c)
            ws = _ws(nbytes, x.device)
            _C.call("vspw_instance_norm_coeffs", _p(x), n, hw, c, 1e-5, _p(coef[0]), _p(coef[1]), _p(ws), nbytes,
                    _stream())
            scale, shift, stride = coef[0], coef[1], c
        _C.call("vspw_affine_act", _p(x), _p(scale), _p(shift), stride, _p(residual), relu_in, relu_out, _p(y), n, hw, c,
                _stream())

    def _encoder(self, enc, prefix, P, img, out_convs):
        """img: NHWC buffer [n][H][W][3] (2-D view [n*H*W, 3]).  out_convs: [(packed-key, act, dst ptr, ldy)] for the
        final 1x1 conv (one entry for fnet, two for cnet: tanh half and relu half).  Returns (h, w) of the output."""
        dev = img.device
        n, H, W = img.shape[0], img.shape[1], img.shape[2]

        def buf(rows, c):
            return torch.empty((rows, c), device=dev, dtype=torch.float32)

        wt, b = P[prefix + ".conv1"]
        h, w = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        t = buf(n * h * w, 64)
        _conv(_p(img), n, H, W, 3, 3, wt, b, 7, 7, 2, (3, 3), ACT_NONE, _p(t), 64)
        x = buf(n * h * w, 64)
        self._norm_act(P, enc, prefix + ".norm1", t, n, h * w, 64, None, 1, 0, x)
        c = 64
        for lname in ("layer1", "layer2", "layer3"):
            for bi, blk in enumerate(getattr(enc, lname)):
                bp = "%s.%s.%d" % (prefix, lname, bi)
                planes = blk.conv1.out_channels
                s = blk.stride
                oh, ow = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
                rows = n * oh * ow
                wt, b = P[bp + ".conv1"]
                t1 = buf(rows, planes)
                _conv(_p(x), n, h, w, c, c, wt, b, 3, 3, s, (1, 1), ACT_NONE, _p(t1), planes)
                y1 = buf(rows, planes)
                self._norm_act(P, enc, bp + ".norm1", t1, n, oh * ow, planes, None, 1, 0, y1)
                wt, b = P[bp + ".conv2"]
                _conv(_p(y1), n, oh, ow, planes, planes, wt, b, 3, 3, 1, (1, 1), ACT_NONE, _p(t1), planes)
                if s != 1:
                    wt, b = P[bp + ".downsample.0"]
                    td = buf(rows, planes)
                    _conv(_p(x), n, h, w, c, c, wt, b, 1, 1, s, (0, 0), ACT_NONE, _p(td), planes)
                    skip = buf(rows, planes)
                    self._norm_act(P, enc, bp + ".downsample.1", td, n, oh * ow, planes, None, 0, 0, skip)
                else:
                    skip = x
                out = buf(rows, planes)
                # relu(skip + relu(norm2(conv2)))  (extractor.py:50-56)
                self._norm_act(P, enc, bp + ".norm2", t1, n, oh * ow, planes, skip, 1, 1, out)
                x, h, w, c = out, oh, ow, planes
        for key, act, dst, ldy in out_convs:
            wt, b = P[key]
            _conv(_p(x), n, h, w, c, c, wt, b, 1, 1, 1, (0, 0), act, dst, ldy)
        return h, w

    # ------------------------------------------------------------------------------------------------- forward
    @torch.no_grad()
    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        if not test_mode:
            raise NotImplementedError("the frozen flow network is only run with test_mode=True (models/netwarp.py:175)")
        if flow_init is not None:
            raise NotImplementedError("flow_init is not used on the NetWarp path")
        _require_gpu(image1, "RAFT")
        _require_gpu(image2, "RAFT")
        N, _, H, W = image1.shape
        if H % 8 or W % 8 or H < 128 or W < 128:
            raise ValueError("RAFT needs H, W divisible by 8 (InputPadder, models/netwarp.py:172-174) and >= 128, got "
                             "%dx%d" % (H, W))
        dev = image1.device
        P = self._pack_all()
        hd = self.hidden_dim

        # raft.py:78-82 (plumbing on the 3-channel images), then NHWC
        im = torch.cat([image1, image2], 0)
        im = (2 * (im / 255.0) - 1.0).permute(0, 2, 3, 1).contiguous()

        h8, w8 = H // 8, W // 8
        hw = h8 * w8
        rows = N * hw
        f32 = dict(device=dev, dtype=torch.float32)
        fmap = torch.empty((2 * N, hw, 256), **f32)
        self._encoder(self.fnet, "fnet", P, im, [("fnet.conv2", ACT_NONE, _p(fmap), 256)])

        # update-block buffers: HX = [h | inp | motion(126) flow(2)], RHX = [r*h | inp | motion flow]
        HX = torch.empty((rows, 384), **f32)
        RHX = torch.empty((rows, 384), **f32)
        self._encoder(self.cnet, "cnet", P, im[:N],
                      [("cnet.conv2.net", ACT_TANH, _p(HX), 384), ("cnet.conv2.inp", ACT_RELU, _off(HX, hd), 384)])
        _C.call("vspw_copy_channels", _p(HX), _p(RHX), rows, 128, 384, 128, 384, 128, _stream())

        # all-pairs correlation volume + pyramid (corr.py:17-29,54-62); 1/sqrt(256) = 2^-4 is folded into fmap1 (exact)
        f1 = fmap[:N]
        _C.call("vspw_axpby", _p(f1), _p(f1), f1.numel(), 1.0 / math.sqrt(256.0), 0.0, _stream())
        pyr = [torch.empty((rows, hw), **f32)]
        # corr[b] = fmap1[b] @ fmap2[b]^T for every pair in one batched launch
        GEMM_FLOPS["total"] += 2.0 * N * hw * hw * 256
        _C.call("vspw_bmm_nt", _p(fmap[:N]), _p(fmap[N:]), _p(pyr[0]), N, hw, hw, 256, _stream())
        lh, lw = h8, w8
        for _ in range(self.corr_levels - 1):
            nxt = torch.empty((rows, (lh // 2) * (lw // 2)), **f32)
            _C.call("vspw_avgpool2x2", _p(pyr[-1]), _p(nxt), rows, lh, lw, _stream())
            pyr.append(nxt)
            lh, lw = lh // 2, lw // 2

        flow = torch.zeros((rows, 2), **f32)
        ldc = 352 if _RAFT_PADK else 324
        corr = torch.zeros((rows, ldc), **f32)  # (columns 324 .. ldc stay zero: see _pack_all)
        cor1 = torch.empty((rows, 256), **f32)
        CF = torch.empty((rows, 256), **f32)  # [cor(192) | flo(64)]
        flo1 = torch.empty((rows, 128), **f32)
        ZR = torch.empty((rows, 256), **f32)
        Q = torch.empty((rows, 128), **f32)
        FH = torch.empty((rows, 256), **f32)
        delta = torch.empty((rows, 2), **f32)
        ub = "update_block."

        wino_m = {}

        def conv(x, c, ldx, key, kh, kw, pad, act, y, ldy):
            wk = P.get("wino:" + key)
            if wk is not None and act in (ACT_NONE, ACT_RELU):
                # M = (B^T d B) U^T for the 16 transform positions (input transform inside the GEMM's operand staging),
                # then y = act(A^T M A + bias) into the destination channel slot
                u, b, k, cc = wk
                d = ConvDesc(N, h8, w8, cc, h8, w8, k, 3, 3, 1, 1, 1, 1)
                T = int(_C.query("vspw_wino_tiles", ctypes.byref(d)))
                m = wino_m.get(k)
                if m is None:
                    m = wino_m[k] = torch.empty((16, T, k), **f32)
                GEMM_FLOPS["total"] += 2.0 * rows * k * 9 * cc       # direct-convolution FLOPs this replaces
                GEMM_FLOPS["executed"] = GEMM_FLOPS.get("executed", 0.0) + 2.0 * 16 * T * k * cc
                # row-fused form (csrc/wino_rows.hip) once there is a workgroup per CU (480 x 856 frames: 272; measured
                # 22.05 -> 21.76 ms per forward; NetWarp's 480 x 480 crops give 152 and stay with the 16 short GEMMs)
                tpad = 0
                if _RAFT_WROWS and k % 128 == 0 and 4 * ((T + 95) // 96) * (k // 128) >= 256:
                    tpad = int(_C.query("vspw_wino_rows_tpad", ctypes.byref(d), cc, k, 1))
                if tpad and tpad * 8 <= T * 16:  # (its 8 planes fit the 16-plane buffer)
                    _C.call("vspw_wino_gemm_fused_rows_ex", ctypes.byref(d), x, ldx, cc, _p(u), k, _p(m), _stream())
                    _C.call("vspw_wino_output_rows_ex", ctypes.byref(d), _p(m), tpad, k, _p(b), y, ldy,
                            1 if act == ACT_RELU else 0, _stream())
                    return
                _C.call("vspw_wino_gemm_fused_ex", ctypes.byref(d), x, ldx, cc, _p(u), k, _p(m), _stream())
                _C.call("vspw_wino_output_ex", ctypes.byref(d), _p(m), k, _p(b), y, ldy, 1 if act == ACT_RELU else 0,
                        _stream())
                return
            wt, b = P[key]
            _conv(x, N, h8, w8, c, ldx, wt, b, kh, kw, 1, pad, act, y, ldy)

        for _ in range(iters):
            _C.call("vspw_corr_lookup", _p(pyr[0]), _p(pyr[1]), _p(pyr[2]), _p(pyr[3]), _p(flow), 2, _p(corr), ldc, N,
                    h8, w8, _stream())
            # BasicMotionEncoder (update.py:88-96)
            conv(_p(corr), ldc, ldc, ub + "encoder.convc1", 1, 1, (0, 0), ACT_RELU, _p(cor1), 256)
            conv(_p(cor1), 256, 256, ub + "encoder.convc2", 3, 3, (1, 1), ACT_RELU, _p(CF), 256)
            conv(_p(flow), 2, 2, ub + "encoder.convf1", 7, 7, (3, 3), ACT_RELU, _p(flo1), 128)
            conv(_p(flo1), 128, 128, ub + "encoder.convf2", 3, 3, (1, 1), ACT_RELU, _off(CF, 192), 256)
            conv(_p(CF), 256, 256, ub + "encoder.conv", 3, 3, (1, 1), ACT_RELU, _off(HX, 256), 384)
            _C.call("vspw_copy_channels", _p(flow), _p(HX), rows, 2, 2, 0, 384, 382, _stream())
            _C.call("vspw_copy_channels", _p(HX), _p(RHX), rows, 128, 384, 256, 384, 256, _stream())
            # SepConvGRU (update.py:44-60): horizontal (1x5) then vertical (5x1)
            for sfx, kh, kw, pad in (("1", 1, 5, (0, 2)), ("2", 5, 1, (2, 0))):
                conv(_p(HX), 384, 384, ub + "gru.convzr" + sfx, kh, kw, pad, ACT_SIGMOID, _p(ZR), 256)
                _C.call("vspw_gru_rh", _p(ZR), 256, _p(HX), 384, _p(RHX), 384, rows, hd, _stream())
                conv(_p(RHX), 384, 384, ub + "gru.convq" + sfx, kh, kw, pad, ACT_TANH, _p(Q), 128)
                _C.call("vspw_gru_update", _p(ZR), 256, _p(Q), 128, _p(HX), 384, rows, hd, _stream())
            # FlowHead (update.py:13-14) on h = HX[:, :128]
            conv(_p(HX), hd, 384, ub + "flow_head.conv1", 3, 3, (1, 1), ACT_RELU, _p(FH), 256)
            conv(_p(FH), 256, 256, ub + "flow_head.conv2", 3, 3, (1, 1), ACT_NONE, _p(delta), 2)
            _C.call("vspw_axpby", _p(delta), _p(flow), flow.numel(), 1.0, 1.0, _stream())  # coords1 += delta_flow

        flow_low = flow.view(N, h8, w8, 2).permute(0, 3, 1, 2)
        # mask head (update.py:122-125,134) and convex upsampling (raft.py:57-68) for the returned prediction only
        M1 = torch.empty((rows, 256), **f32)
        MK = torch.empty((rows, 576), **f32)
        conv(_p(HX), hd, 384, ub + "mask.0", 3, 3, (1, 1), ACT_RELU, _p(M1), 256)
        conv(_p(M1), 256, 256, ub + "mask.2", 1, 1, (0, 0), ACT_NONE, _p(MK), 576)
        flow_up = torch.empty((N, 2, H, W), **f32)
        _C.call("vspw_convex_upsample", _p(flow), 2, _p(MK), 576, 0.25, _p(flow_up), N, h8, w8, _stream())
        return flow_low, flow_up
