"""The frozen RAFT flow network on the HIP kernels (cvpr2021_vspw_implement_amd/models/raft.py, csrc/raft.hip) against
(1) the vectors the reference's RAFT_core produced (tests/golden/raft_basic.npz) and (2) the numpy oracle run live on
other sizes; plus kernel-level checks of the correlation lookup, instance norm, convex upsampling and the extended
convolution entry point (per-axis padding, strided channel slices, fused activation) through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import np_raft
from oracle.det_init import det_input

from helpers import golden, raft_images, raft_state

pytestmark = pytest.mark.gpu


def _raft(dev, fx):
    from cvpr2021_vspw_implement_amd.models.raft import RAFT

    sd = raft_state(fx)
    m = RAFT()
    assert list(m.state_dict().keys()) == [str(k) for k in fx["sd_keys"]]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


def test_raft_matches_reference_vectors(dev):
    fx = golden("raft_basic")
    m, _ = _raft(dev, fx)
    a, b = raft_images("raft_basic", (1, 3, 128, 192))
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    low1, up1 = m(ta, tb, iters=1, test_mode=True)
    assert np.abs(low1.cpu().numpy() - fx["flow_low_it1"]).max() < 5e-4
    assert np.abs(up1.cpu().numpy() - fx["flow_up_it1"]).max() < 4e-3
    low, up = m(ta, tb, iters=4, test_mode=True)
    assert low.shape == (1, 2, 16, 24) and up.shape == (1, 2, 128, 192)
    # |flow| reaches 13.7 px; the oracle itself sits 8e-5 / 2.5e-4 from these vectors
    assert np.abs(low.cpu().numpy() - fx["flow_low_it4"]).max() < 2e-3
    assert np.abs(up.cpu().numpy() - fx["flow_up_it4"]).max() < 1.6e-2


def test_raft_matches_oracle_batch2_other_size(dev):
    fx = golden("raft_basic")
    m, sd = _raft(dev, fx)
    a, b = raft_images("raft_b2", (2, 3, 136, 160))
    low, up = m(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), iters=3, test_mode=True)
    rlow, rup = np_raft.raft_forward(sd, a, b, iters=3)
    assert np.abs(low.cpu().numpy() - rlow).max() < 2e-3 * max(1.0, np.abs(rlow).max() / 10)
    assert np.abs(up.cpu().numpy() - rup).max() < 1.6e-2 * max(1.0, np.abs(rup).max() / 80)


def test_corr_lookup_kernel(dev):
    from cvpr2021_vspw_implement_amd import _C
    from cvpr2021_vspw_implement_amd.ops import _p, _stream

    B, h, w = 2, 16, 24
    hw = h * w
    f1 = det_input("lk:f1", (B, 32, h, w))
    f2 = det_input("lk:f2", (B, 32, h, w))
    cb = np_raft.CorrBlock(f1, f2)
    flow = det_input("lk:flow", (B, 2, h, w), scale=3.0)
    flow[0, :, 0, 0] = (-30.0, 4.0)  # far outside: zero padding
    ref = cb(np_raft.coords_grid(B, h, w, np.float32) + flow)  # [B,324,h,w]
    pyr = [torch.from_numpy(np.ascontiguousarray(p.reshape(B * hw, -1))).to(dev) for p in cb.pyr]
    fl = torch.from_numpy(np.ascontiguousarray(flow.transpose(0, 2, 3, 1).reshape(B * hw, 2))).to(dev)
    out = torch.empty((B * hw, 324), device=dev)
    _C.call("vspw_corr_lookup", _p(pyr[0]), _p(pyr[1]), _p(pyr[2]), _p(pyr[3]), _p(fl), 2, _p(out), 324, B, h, w,
            _stream())
    got = out.cpu().numpy().reshape(B, h, w, 324).transpose(0, 3, 1, 2)
    assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    # pyramid pooling kernel
    nxt = torch.empty((B * hw, (h // 2) * (w // 2)), device=dev)
    _C.call("vspw_avgpool2x2", _p(pyr[0]), _p(nxt), B * hw, h, w, _stream())
    assert np.abs(nxt.cpu().numpy() - cb.pyr[1].reshape(B * hw, -1)).max() < 1e-5


def test_instance_norm_and_affine_act(dev):
    from cvpr2021_vspw_implement_amd import _C
    from cvpr2021_vspw_implement_amd.ops import _p, _stream

    n, h, w, c = 3, 37, 41, 96
    x = det_input("in:x", (n, c, h, w), scale=2.0) + 0.7
    res = det_input("in:res", (n, c, h, w))
    ref = np.maximum(res + np.maximum(np_raft.instance_norm(x), 0), 0)
    tx = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).to(dev)
    tr = torch.from_numpy(np.ascontiguousarray(res.transpose(0, 2, 3, 1))).to(dev)
    coef = torch.empty((2, n, c), device=dev)
    nbytes = _C.query("vspw_instance_norm_workspace", n, h * w, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _C.call("vspw_instance_norm_coeffs", _p(tx), n, h * w, c, 1e-5, _p(coef[0]), _p(coef[1]), _p(ws), nbytes, _stream())
    y = torch.empty_like(tx)
    _C.call("vspw_affine_act", _p(tx), _p(coef[0]), _p(coef[1]), c, _p(tr), 1, 1, _p(y), n, h * w, c, _stream())
    got = y.cpu().numpy().transpose(0, 3, 1, 2)
    assert np.abs(got - ref).max() < 2e-5


def test_convex_upsample_kernel(dev):
    from cvpr2021_vspw_implement_amd import _C
    from cvpr2021_vspw_implement_amd.ops import _p, _stream

    n, h, w = 2, 5, 7
    flow = det_input("up:flow", (n, 2, h, w), scale=3.0)
    mask = det_input("up:mask", (n, 576, h, w), scale=4.0)
    ref = np_raft.upsample_flow(flow, mask * np.float32(0.25))
    tf = torch.from_numpy(np.ascontiguousarray(flow.transpose(0, 2, 3, 1))).to(dev)
    tm = torch.from_numpy(np.ascontiguousarray(mask.transpose(0, 2, 3, 1))).to(dev)
    out = torch.empty((n, 2, 8 * h, 8 * w), device=dev)
    _C.call("vspw_convex_upsample", _p(tf), 2, _p(tm), 576, 0.25, _p(out), n, h, w, _stream())
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-4


@pytest.mark.parametrize("kh,kw,pad,c,k,act", [(1, 5, (0, 2), 64, 40, 2), (5, 1, (2, 0), 64, 40, 3),
                                                (7, 7, (3, 3), 2, 24, 1), (3, 3, (1, 1), 32, 2, 0)])
def test_conv_ex_slices_padding_activation(dev, kh, kw, pad, c, k, act):
    """vspw_conv2d_fwd_ex: the input is channels [8, 8+c) of a wider NHWC buffer, the output lands in channels
    [4, 4+k) of another one, padding differs per axis, activation fused."""
    from cvpr2021_vspw_implement_amd import _C
    from cvpr2021_vspw_implement_amd._C import ConvDesc
    from cvpr2021_vspw_implement_amd.ops import _p, _stream

    n, h, w = 2, 11, 13
    ldx, ldy = c + 16, k + 12
    xfull = det_input("cx:x%d%d" % (kh, c), (n, ldx, h, w))
    wt = det_input("cx:w%d%d" % (kh, c), (k, c, kh, kw), scale=0.2)
    bias = det_input("cx:b%d%d" % (kh, c), (k,))
    ref = np_raft.conv2d(xfull[:, 8:8 + c], wt, bias, 1, pad)
    ref = {0: lambda v: v, 1: np_raft.relu, 2: np_raft.sigmoid, 3: np.tanh}[act](ref)
    tx = torch.from_numpy(np.ascontiguousarray(xfull.transpose(0, 2, 3, 1))).to(dev)
    tw = torch.from_numpy(np.ascontiguousarray(wt.transpose(0, 2, 3, 1))).to(dev)
    tb = torch.from_numpy(bias).to(dev)
    y = torch.full((n, h, w, ldy), -7.0, device=dev)
    d = ConvDesc(n, h, w, c, h, w, k, kh, kw, 1, pad[0], 1, pad[1])
    _C.call("vspw_conv2d_fwd_ex", ctypes.byref(d), ctypes.c_void_p(tx.data_ptr() + 32), ldx, _p(tw), _p(tb), None, act,
            ctypes.c_void_p(y.data_ptr() + 16), ldy, _stream())
    got = y.cpu().numpy()
    assert np.abs(got[..., 4:4 + k].transpose(0, 3, 1, 2) - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    assert (got[..., :4] == -7.0).all() and (got[..., 4 + k:] == -7.0).all()  # neighbouring slots untouched


@pytest.mark.parametrize("kh,kw,pad,c,k,act,form", [
    (3, 3, (1, 1), 256, 2, 0, 1),   # FlowHead.conv2
    (3, 3, (1, 1), 64, 1, 1, 1),    # one output, lanes 16-63 idle
    (1, 1, (0, 0), 384, 4, 2, 1),   # c > 256: two passes over the channels, partial sums parked in y
    (3, 1, (1, 0), 128, 3, 3, 1),
    (7, 7, (3, 3), 2, 128, 1, 0),   # BasicMotionEncoder.convf1: stays an implicit GEMM (tools/diag/thin_time.py)
    (3, 3, (1, 1), 32, 8, 0, 0),    # not thin: the caller keeps vspw_conv2d_fwd_ex
])
def test_thin_convolutions_direct_forms(dev, kh, kw, pad, c, k, act, form):
    """vspw_conv2d_thin (csrc/raft.hip): the few-outputs direct kernel against the numpy convolution, with the input a
    channel slice of a wider NHWC buffer, the output a slot of another one, ragged pixel counts (the last wave is partial),
    every activation; vspw_conv2d_thin_supported says whether it runs."""
    from cvpr2021_vspw_implement_amd import _C
    from cvpr2021_vspw_implement_amd._C import ConvDesc
    from cvpr2021_vspw_implement_amd.ops import _p, _stream

    n, h, w = 2, 11, 13
    ldx, ldy = c + 16, k + 12
    tag = "thin:%d%d%d%d" % (kh, kw, c, k)
    xfull = det_input(tag + "x", (n, ldx, h, w))
    wt = det_input(tag + "w", (k, c, kh, kw), scale=0.2)
    bias = det_input(tag + "b", (k,))
    d = ConvDesc(n, h, w, c, h, w, k, kh, kw, 1, pad[0], 1, pad[1])
    assert int(_C.query("vspw_conv2d_thin_supported", ctypes.byref(d), ldx, ldy)) == form
    if form == 0:
        return
    ref = np_raft.conv2d(xfull[:, 8:8 + c].astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, pad)
    ref = {0: lambda v: v, 1: np_raft.relu, 2: np_raft.sigmoid, 3: np.tanh}[act](ref)
    tx = torch.from_numpy(np.ascontiguousarray(xfull.transpose(0, 2, 3, 1))).to(dev)
    tw = torch.from_numpy(np.ascontiguousarray(wt.transpose(0, 2, 3, 1))).to(dev)
    tb = torch.from_numpy(bias).to(dev)
    for b_ in (tb, None):
        y = torch.full((n, h, w, ldy), -7.0, device=dev)
        _C.call("vspw_conv2d_thin", ctypes.byref(d), ctypes.c_void_p(tx.data_ptr() + 32), ldx, _p(tw), _p(b_), act,
                ctypes.c_void_p(y.data_ptr() + 16), ldy, _stream())
        got = y.cpu().numpy()
        want = ref if b_ is not None else {0: lambda v: v, 1: np_raft.relu, 2: np_raft.sigmoid, 3: np.tanh}[act](
            np_raft.conv2d(xfull[:, 8:8 + c].astype(np.float64), wt.astype(np.float64), np.zeros(k), 1, pad))
        assert np.abs(got[..., 4:4 + k].transpose(0, 3, 1, 2) - want).max() < 2e-5 * max(1.0, np.abs(want).max())
        assert (got[..., :4] == -7.0).all() and (got[..., 4 + k:] == -7.0).all()  # neighbouring slots untouched


def test_raft_full_size_constant_flow_property(dev):
    """At the NetWarp working size (480x856 = 479x853 zero-padded, B = 2, 20 iterations): with the flow head's last
    conv zeroed and its bias set to (a, b) every iteration adds exactly (a, b), so flow_low = 20*(a, b) everywhere and,
    convex upsampling being a convex combination, flow_up = 8 * flow_low away from the zero-padded border."""
    fx = golden("raft_basic")
    m, _ = _raft(dev, fx)
    with torch.no_grad():
        m.update_block.flow_head.conv2.weight.zero_()
        m.update_block.flow_head.conv2.bias.copy_(torch.tensor([0.25, -0.125], device=dev))
    g = torch.Generator().manual_seed(9)
    a = (torch.rand(2, 3, 480, 856, generator=g) * 255).to(dev)
    b = (torch.rand(2, 3, 480, 856, generator=g) * 255).to(dev)
    low, up = m(a, b, iters=20, test_mode=True)
    assert low.shape == (2, 2, 60, 107) and up.shape == (2, 2, 480, 856)
    ref = torch.tensor([5.0, -2.5], device=dev).view(1, 2, 1, 1)
    assert (low - ref).abs().max().item() < 1e-5
    assert (up[:, :, 8:-8, 8:-8] - 8 * ref).abs().max().item() < 2e-4
