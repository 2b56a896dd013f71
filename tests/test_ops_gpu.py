"""Kernel-level parity: every HIP kernel (called through the C ABI via ops.py) against a plain PyTorch fp32 CPU
evaluation of the ATen op it replaces, forward and backward, on seeded inputs incl. ragged / odd sizes.
Tolerances: fp32 kernels with different summation order -> rtol 1e-4 / atol scaled to the reduction length."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol, what):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= tol * max(ref, 1.0), "%s: max abs err %.3e (ref max %.3e, tol %.1e)" % (what, err, ref, tol)


CONV_CASES = [
    # n, c, h, w, k, ks, stride, pad, dil, bias
    (2, 64, 20, 27, 64, 3, 1, 2, 2, False),    # dil-2 (layer3 style), ragged M
    (2, 32, 17, 19, 48, 3, 1, 4, 4, False),    # dil-4 (layer4 style), ragged N
    (2, 64, 21, 23, 128, 3, 2, 1, 1, False),   # stride-2 3x3 (layer2.0.conv2)
    (2, 64, 21, 23, 256, 1, 2, 0, 1, False),   # stride-2 1x1 downsample
    (3, 256, 9, 13, 124, 1, 1, 0, 1, True),    # class head: Cout=124, bias
    (2, 3, 33, 35, 64, 3, 2, 1, 1, False),     # stem conv1: Cin=3 scalar gather
    (1, 11, 18, 22, 16, 3, 1, 1, 1, False),    # FlowCNN 11->16
    (2, 4, 10, 12, 2, 3, 1, 1, 1, False),      # FlowCNN 4->2 (Cout=2)
    (2, 512, 12, 12, 256, 3, 1, 1, 1, True),   # bigger K: 4608 reduction
    (1, 124, 16, 16, 256, 1, 1, 0, 1, False),  # K=124 (not a multiple of 32)
    # Winograd F(2x2,3x3) path (stride 1, >= 128 channels on both sides): odd sizes, dilation sub-grids with ragged tiles
    (2, 128, 15, 15, 256, 3, 1, 4, 4, False),  # layer4 style: 4x4 sub-grids of 4x4 / 4x3 / 3x3 pixels
    (2, 256, 13, 17, 128, 3, 1, 2, 2, True),   # layer3 style, bias
    (3, 128, 9, 10, 128, 3, 1, 1, 1, False),   # undilated, one odd side
    (1, 1024, 6, 7, 512, 3, 1, 1, 1, False),   # deepsup shape class: 9216-long direct reduction
    # few-row pointwise GEMMs of the pyramid-pool branches (skinny_pointwise_kernel: a wave per output channel)
    (2, 2048, 1, 1, 512, 1, 1, 0, 1, False),   # scale-1 branch: 2 rows
    (2, 2048, 6, 6, 512, 1, 1, 0, 1, False),   # scale-6 branch: 72 rows = three row groups
    (3, 768, 5, 5, 66, 1, 1, 0, 1, True),      # 75 rows, 3 passes over K, ragged output-channel count, bias
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(dev, case):
    from cvpr2021_vspw_implement_amd import ops

    n, c, h, w, k, ks, s, p, d, bias = case
    g = torch.Generator().manual_seed(1234 + c + k)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, ks, ks, generator=g) * (2.0 / (c * ks * ks)) ** 0.5
    b = torch.randn(k, generator=g) if bias else None
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=s, padding=p, dilation=d)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd = x.to(dev).requires_grad_(True)
    wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if bias else None
    yd = ops.conv2d(xd, wd, bd, s, p, d)
    yd.backward(gy.to(dev))
    red = c * ks * ks
    _close(yd, yr, 2e-6 * red ** 0.5 + 1e-5, "conv fwd %s" % (case,))
    _close(xd.grad, xr.grad, 2e-6 * (k * ks * ks) ** 0.5 + 1e-5, "conv dgrad %s" % (case,))
    _close(wd.grad, wr.grad, 3e-6 * (n * yr.shape[2] * yr.shape[3]) ** 0.5 + 1e-5, "conv wgrad %s" % (case,))
    if bias:
        _close(bd.grad, br.grad, 1e-5, "conv bias grad %s" % (case,))


@pytest.fixture
def wino_rows_tile(request):
    """Force the row-fused Winograd GEMM (csrc/wino_rows.hip) with the given tile: the library's own dispatch rule only
    takes it for grids that fill the chip, which unit-test shapes never do.  (F(2x2) only: F(3x3) is switched off.)"""
    from cvpr2021_vspw_implement_amd import _C, ops

    _C.call("vspw_wino_rows_config", max(int(request.param), 0))
    # -3 / -4: F(3x3,3x3) / F(4x4,3x3) instead (no row-fused form)
    prev = ops.set_winograd_f3(-int(request.param) if int(request.param) < 0 else False)
    yield int(request.param)
    ops.set_winograd_f3(prev)
    _C.call("vspw_wino_rows_config", 0)


@pytest.fixture
def wino_f3(request):
    """Tile size of the Winograd path: False = F(2x2,3x3) (csrc/winograd.hip), 3 / 4 = F(3x3,3x3) / F(4x4,3x3) forced
    (csrc/winograd_f3.hip), True = the automatic choice."""
    from cvpr2021_vspw_implement_amd import ops

    prev = ops.set_winograd_f3(request.param)
    yield request.param
    ops.set_winograd_f3(prev)


WINO_CASES = [c for c in CONV_CASES if c[5] == 3 and c[6] == 1 and min(c[1], c[4]) >= 128] + [
    # exact 3x3 tilings (the bench geometry in small: 30 / 15 pixel sub-grids) and ragged ones
    (2, 128, 30, 30, 128, 3, 1, 2, 2, False),
    (1, 256, 15, 30, 128, 3, 1, 4, 4, True),
    (2, 128, 7, 11, 160, 3, 1, 1, 1, False),
    (1, 160, 20, 9, 128, 3, 1, 2, 2, False),
    (2, 128, 16, 24, 128, 3, 1, 1, 1, True),   # exact 4x4 tiling
    (2, 128, 15, 20, 128, 3, 1, 1, 1, False),  # exact 5x5 tiling
    (1, 128, 30, 25, 160, 3, 1, 2, 2, False),  # 5x5 tiles on dilation-2 sub-grids, one ragged
]


@pytest.mark.parametrize("wino_f3", [False, 3, 4, 5, True], indirect=True)
@pytest.mark.parametrize("case", WINO_CASES)
def test_conv2d_winograd_tile_sizes(dev, case, wino_f3):
    """The parity gate of test_conv2d_fwd_bwd (against F.conv2d on the CPU) for every Winograd tile size, all three
    passes (ragged tiles, dilation sub-grids); checks that the F(3x3) / F(4x4) launches happened when asked for."""
    from cvpr2021_vspw_implement_amd import ops

    keys = ("f3_launches", "f4_launches", "f5_launches")
    before = [ops._wino[q] for q in keys]
    test_conv2d_fwd_bwd(dev, case)
    got = tuple(ops._wino[q] - b for q, b in zip(keys, before))
    if wino_f3 is True:  # forward and weight gradient share V: one tile size; the data gradient takes F(5x5)
        assert sum(got) == 3 and got[2] >= 1 and sorted(got) in ([0, 0, 3], [0, 1, 2])
    else:
        assert got == {False: (0, 0, 0), 3: (3, 0, 0), 4: (0, 3, 0), 5: (0, 0, 3)}[wino_f3]


@pytest.mark.parametrize("wino_rows_tile", [12, 31, 22], indirect=True)
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[5] == 3 and c[6] == 1 and min(c[1], c[4]) >= 128])
def test_conv2d_winograd_row_fused_form(dev, case, wino_rows_tile):
    """The same parity gate as test_conv2d_fwd_bwd (against F.conv2d on the CPU) with the four GEMMs of a transform row
    fused into one workgroup: plain V operand (forward), fused operand (data gradient), every tile height, ragged tile
    counts (padded P planes), dilation sub-grids."""
    from cvpr2021_vspw_implement_amd import _C, ops

    n, c, h, w, k, ks, s, p, d, bias = case
    dsc = ops._conv_desc(torch.empty(n, c, h, w, device="meta"), k, ks, ks, s, p, d)
    assert _C.query("vspw_wino_rows_prefer", ctypes.byref(dsc), c, k, 0) == 1
    test_conv2d_fwd_bwd(dev, case)


@pytest.mark.parametrize("wino_rows_tile", [0, 12, 31, -3, -4, -5], indirect=True)
@pytest.mark.parametrize("dil,h,w", [(1, 12, 13), (2, 14, 14), (4, 15, 15)])
def test_winograd_path_equals_direct_path_in_a_fused_chain(dev, dil, h, w, wino_rows_tile):
    """1x1 conv+BN+ReLU -> 3x3 conv+BN+ReLU (fuse_input: the 3x3 data gradient carries the first node's BatchNorm-backward
    front end) -> sum of squares: outputs, batch statistics (taken from the output transform's partial sums) and every
    gradient with the Winograd path against the direct implicit GEMM.  Both are float32 evaluations of the same
    expression, so they agree to rounding; also checks that the Winograd launches actually happened."""
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(77 + dil)
    n, c0, c1, c2 = 2, 64, 128, 256
    x = torch.randn(n, c0, h, w, generator=g)
    w1 = torch.randn(c1, c0, 1, 1, generator=g) * 0.2
    w2 = torch.randn(c2, c1, 3, 3, generator=g) * 0.05
    res = []
    for wino in (False, True):
        ops.set_winograd(wino)
        before = ops._wino["launches"]
        try:
            xd = x.to(dev).requires_grad_(True)
            p1 = w1.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            p2 = w2.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            g1, b1 = torch.ones(c1, device=dev, requires_grad=True), torch.zeros(c1, device=dev, requires_grad=True)
            g2, b2 = torch.ones(c2, device=dev, requires_grad=True), torch.zeros(c2, device=dev, requires_grad=True)
            rm1, rv1, rm2, rv2 = (torch.zeros(c1, device=dev), torch.ones(c1, device=dev), torch.zeros(c2, device=dev),
                                  torch.ones(c2, device=dev))
            a = ops.conv_bn_act(xd, p1, None, g1, b1, rm1, rv1, None, None, 1, 0, 1, True, 0.1, 1e-5, True)
            z = ops.conv_bn_act(a, p2, None, g2, b2, rm2, rv2, None, None, 1, dil, dil, True, 0.1, 1e-5, True, False,
                                True)
            (z * z).sum().backward()
            torch.cuda.synchronize()
            res.append([t.detach().float().cpu() for t in (z, rm2, rv2, xd.grad, p1.grad, p2.grad, g1.grad, b1.grad,
                                                            g2.grad, b2.grad)])
        finally:
            ops.set_winograd(True)
        assert (ops._wino["launches"] - before) == (3 if wino else 0)  # forward, data gradient, weight gradient
    names = ("z", "running_mean", "running_var", "dx", "dw1", "dw2", "dgamma1", "dbeta1", "dgamma2", "dbeta2")
    for nm, a, b in zip(names, res[0], res[1]):
        err = (a - b).norm() / b.norm().clamp_min(1e-12)
        assert err < (2e-4 if nm in ("dbeta1", "dbeta2") else 2e-5), (nm, float(err))


@pytest.mark.parametrize("n,c,h,w,k,bias", [
    (4, 2048, 24, 32, 256, False),  # interior 96 / 128-row tiles, 8 chains of 256
    (2, 1024, 12, 13, 200, True),   # ragged rows and columns: guarded flush, bias added once
    (3, 512, 20, 20, 1024, False),  # two chains forward, four in the data gradient (K = output channels there)
    (1, 4096, 40, 48, 512, False),  # 16 chains on the small-problem 64x64 tile
    (4, 2048, 64, 64, 512, False),  # 16 384 rows: the 128x128 tile with TWO LDS buffers (K > 1024), flush scratch in the dead one
])
def test_pointwise_two_level_accumulation(dev, n, c, h, w, k, bias):
    """K >= 2*chunk pointwise GEMMs sum K/chunk chains of chunk terms, the finished chains parked in the output tile
    (igemm_nt_v2_body CHUNK, vspw_set_accum_chunk): same value as the single chain up to rounding, run-to-run
    bit-identical, and CLOSER to the float64 result than the single k-sequential chain (the point of it: the
    reference's k-blocked CPU GEMM has the smaller error - profiles/r05_parity_attrib.log)."""
    from cvpr2021_vspw_implement_amd import ops

    if not ops.accum_chunk_supported():
        pytest.skip("diagnostic builds only (-DVSPW_WITH_ACCUM_CHUNK, tools/diag/build_variant.py)")
    g = torch.Generator().manual_seed(5 + c + k)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, 1, 1, generator=g) * (2.0 / c) ** 0.5
    b = torch.randn(k, generator=g) if bias else None
    gy = torch.randn(n, k, h, w, generator=g)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None if b is None else b.double())
    ref.backward(gy.double())
    ref, ref_dx = ref.detach(), xr.grad
    out = {}
    prev = ops.set_accum_chunk(256)
    try:
        for chunk in (0, 256, 256, 128):
            ops.set_accum_chunk(chunk)
            xd = x.to(dev).requires_grad_(True)
            wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = ops.conv2d(xd, wd, None if b is None else b.to(dev), 1, 0, 1)
            y.backward(gy.to(dev))
            ops.join_side_streams()
            out.setdefault(chunk, []).append((y.detach().double().cpu(), xd.grad.double().cpu()))
    finally:
        ops.set_accum_chunk(prev)
    for what, idx, want, kk in (("forward", 0, ref, c), ("data gradient", 1, ref_dx, k)):
        rms = {c_: float((v[0][idx] - want).pow(2).mean().sqrt()) for c_, v in out.items()}
        mx = {c_: float((v[0][idx] - want).abs().max()) for c_, v in out.items()}
        print("%s K=%d: rms error single chain %.3e, chains of 256 %.3e, of 128 %.3e (max %.2e / %.2e / %.2e)"
              % (what, kk, rms[0], rms[256], rms[128], mx[0], mx[256], mx[128]))
        assert torch.equal(out[256][0][idx], out[256][1][idx]), what  # deterministic
        scale = max(float(want.abs().max()), 1.0)
        for c_ in mx:
            assert mx[c_] <= 2e-6 * scale * max(kk / 256, 1.0) ** 0.5, (what, c_, mx[c_])
        if kk >= 1024:
            assert rms[256] < 0.8 * rms[0], (what, rms)   # the parked chains are measurably more accurate ...
            assert rms[128] <= 1.05 * rms[256], (what, rms)  # ... and shorter ones no worse
        elif kk < 512:
            assert torch.equal(out[0][0][idx], out[256][0][idx]), what  # short reductions: one chain either way


@pytest.fixture(params=[0, 256], ids=["one-chain", "chains-of-256"])
def accum_chunk(request):
    """Run a test under both summation policies of the long pointwise reductions (vspw_set_accum_chunk): the fused
    epilogues / staged operands must compose with the parked partial sums."""
    from cvpr2021_vspw_implement_amd import ops

    if request.param and not ops.accum_chunk_supported():
        pytest.skip("diagnostic builds only (-DVSPW_WITH_ACCUM_CHUNK, tools/diag/build_variant.py)")
    prev = ops.set_accum_chunk(request.param)
    yield request.param
    ops.set_accum_chunk(prev)


@pytest.mark.parametrize("n,c,h,w,k", [(2, 256, 16, 24, 64),   # interior 128x128 / 96-row tiles: accumulators seeded
                                       (1, 64, 9, 13, 32),     # ragged: every tile takes the guarded epilogue
                                       (3, 1024, 20, 20, 256),  # forward K = 1024: four parked chains
                                       (2, 256, 16, 24, 1024),  # data gradient K = 1024: parked chains + skip + BN front
                                       (1, 128, 9, 13, 512)])   # ... on ragged tiles (guarded flush)
def test_conv_bn_act_skip_gradient_is_folded_into_dgrad(dev, n, c, h, w, k, accum_chunk):
    """Bottleneck entry (models/resnet.py:75-90): x feeds conv1 AND the skip.  With skip_out the skip gradient is added
    in conv1's data-gradient epilogue (vspw_conv2d_bwd_data_acc); the total must equal autograd's sum of both paths."""
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(77 + c)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, 1, 1, generator=g) * (2.0 / c) ** 0.5
    gamma, beta = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g) * 0.1
    gz, gs = torch.randn(n, k, h, w, generator=g), torch.randn(n, c, h, w, generator=g)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    zr = F.relu(F.batch_norm(F.conv2d(xr, wr), None, None, gamma, beta, True, 0.1, 1e-5))
    torch.autograd.backward([zr, xr * 1.0], [gz, gs])  # second output: the skip path
    xd = x.to(dev).requires_grad_(True)
    wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rm, rv = torch.zeros(k, device=dev), torch.ones(k, device=dev)
    zd, skip = ops.conv_bn_act(xd, wd, None, gamma.to(dev), beta.to(dev), rm, rv, training=True, relu=True,
                               skip_out=True)
    assert skip.shape == xd.shape and skip.grad_fn is zd.grad_fn  # the skip output belongs to conv1's node
    torch.autograd.backward([zd, skip], [gz.to(dev), gs.to(dev)])
    _close(zd, zr, 1e-4, "fwd")
    _close(xd.grad, xr.grad, 2e-4, "dgrad + skip gradient")
    _close(wd.grad, wr.grad, 3e-4, "wgrad")


@pytest.mark.parametrize("n,cm,c,h,w,k,fused", [(2, 64, 256, 16, 24, 64, True),    # interior tiles, two K-tiles
                                                 (3, 32, 1024, 20, 20, 256, True),  # layer-3 shape: 96-row tiles
                                                 (1, 32, 64, 9, 13, 32, True),      # ragged rows: clamped A rows
                                                 (2, 16, 48, 10, 10, 32, False)])   # C % 32 != 0: plain apply pass
def test_block_output_evaluated_by_the_next_conv1(dev, n, cm, c, h, w, k, fused, accum_chunk):
    """A residual block's relu(bn3(conv3(u)) + skip) left to the next block's conv1 (models/resnet.py:83-90 then :75,
    ops._fwd_apply / vspw_conv2d_fwd_apply): the block output, conv1's output and every gradient against autograd on
    the plain composition; the fused kernel is taken exactly when the geometry allows it."""
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(31 + c)
    u = torch.randn(n, cm, h, w, generator=g)
    skip = torch.randn(n, c, h, w, generator=g)
    w3 = torch.randn(c, cm, 1, 1, generator=g) * (2.0 / cm) ** 0.5
    w1 = torch.randn(k, c, 1, 1, generator=g) * (2.0 / c) ** 0.5
    g3, b3 = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    g1, b1 = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g) * 0.1
    go = torch.randn(n, k, h, w, generator=g)
    gz = torch.randn(n, c, h, w, generator=g)  # the block output's second reader (the next skip connection)
    ref = [t.clone().requires_grad_(True) for t in (u, skip, w3, w1)]
    zr = F.relu(F.batch_norm(F.conv2d(ref[0], ref[2]), None, None, g3, b3, True, 0.1, 1e-5) + ref[1])
    orr = F.relu(F.batch_norm(F.conv2d(zr, ref[3]), None, None, g1, b1, True, 0.1, 1e-5))
    torch.autograd.backward([orr, zr], [go, gz])
    dv = [u.to(dev).requires_grad_(True), skip.to(dev).requires_grad_(True),
          w3.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True),
          w1.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)]
    stats = lambda ch: (torch.zeros(ch, device=dev), torch.ones(ch, device=dev))
    before = ops._fwd_apply["nodes"]
    zd = ops.conv_bn_act(dv[0], dv[2], None, g3.to(dev), b3.to(dev), *stats(c), residual=dv[1], training=True,
                         relu=True, defer_apply=True)
    assert (getattr(zd, "_vspw_pending", None) is not None) == (c % 32 == 0)
    od, zskip = ops.conv_bn_act(zd, dv[3], None, g1.to(dev), b1.to(dev), *stats(k), training=True, relu=True,
                                skip_out=True, fuse_input=True)
    assert getattr(zd, "_vspw_pending", None) is None
    assert ops._fwd_apply["nodes"] - before == (1 if fused else 0)
    torch.autograd.backward([od, zskip], [go.to(dev), gz.to(dev)])
    ops.join_side_streams()
    _close(zd, zr, 1e-4, "block output")
    _close(od, orr, 2e-4, "conv1 output")
    for got, want, what in zip(dv, ref, ("d u", "d skip", "d w3", "d w1")):
        _close(got.grad, want.grad, 5e-4, what)


@pytest.mark.parametrize("wino_f3", [False, 3, 4, 5], indirect=True)
@pytest.mark.parametrize("dil,h,w,c1", [(2, 13, 14, 128), (1, 9, 11, 256), (4, 15, 15, 128)])
def test_conv1_apply_evaluated_by_the_winograd_input_transform(dev, dil, h, w, c1, wino_f3):
    """relu(bn1(conv1(x))) left to conv2's Winograd input transform (models/resnet.py:76-79, vspw_wino_input_apply): the
    deferred and the materialised evaluation are the same float32 expression - every output and gradient BIT-identical -
    and both match autograd on the plain composition; ragged dilation sub-grids, padding taps stay zero."""
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(1)  # (seed 9 at dil 4 holds a pre-ReLU value of 6e-9: a coin toss in any float32)
    n, c0, c2 = 2, 64, 128
    x = torch.randn(n, c0, h, w, generator=g)
    w1 = torch.randn(c1, c0, 1, 1, generator=g) * (2.0 / c0) ** 0.5
    w2 = torch.randn(c2, c1, 3, 3, generator=g) * (2.0 / (9 * c1)) ** 0.5
    g1, b1 = torch.rand(c1, generator=g) + 0.5, torch.randn(c1, generator=g) * 0.3
    g2, b2 = torch.rand(c2, generator=g) + 0.5, torch.randn(c2, generator=g) * 0.1
    go = torch.randn(n, c2, h, w, generator=g)
    ref = [t.clone().requires_grad_(True) for t in (x, w1, w2)]
    a = F.relu(F.batch_norm(F.conv2d(ref[0], ref[1]), None, None, g1, b1, True, 0.1, 1e-5))
    o = F.relu(F.batch_norm(F.conv2d(a, ref[2], padding=dil, dilation=dil), None, None, g2, b2, True, 0.1, 1e-5))
    o.backward(go)
    res = []
    for deferred in (False, True):
        ops._fwd_apply["wino"] = deferred
        try:
            dv = [x.to(dev).requires_grad_(True),
                  w1.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True),
                  w2.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)]
            stats = lambda ch: (torch.zeros(ch, device=dev), torch.ones(ch, device=dev))  # noqa: E731
            before = ops._fwd_apply["wino_nodes"]
            ad = ops.conv_bn_act(dv[0], dv[1], None, g1.to(dev), b1.to(dev), *stats(c1), training=True, relu=True,
                                 defer_apply=True)
            assert getattr(ad, "_vspw_pending", None) is not None
            od = ops.conv_bn_act(ad, dv[2], None, g2.to(dev), b2.to(dev), *stats(c2), stride=1, pad=dil, dil=dil,
                                 training=True, relu=True, fuse_input=True)
            assert getattr(ad, "_vspw_pending", None) is None
            assert ops._fwd_apply["wino_nodes"] - before == (1 if deferred else 0)
            od.backward(go.to(dev))
            ops.join_side_streams()
            torch.cuda.synchronize()
            res.append([t.detach().cpu() for t in (ad, od, dv[0].grad, dv[1].grad, dv[2].grad)])
        finally:
            ops._fwd_apply["wino"] = True
    names = ("conv1 node output", "conv2 node output", "d x", "d w1", "d w2")
    for nm, p_, q_ in zip(names, res[0], res[1]):
        assert torch.equal(p_, q_), nm
    _close(res[1][0], a, 1e-4, "conv1 node output")
    _close(res[1][1], o, 3e-4, "conv2 node output")
    for got, want, what in zip(res[1][2:], ref, ("d x", "d w1", "d w2")):
        err = float((got - want.grad).norm() / want.grad.norm())
        assert err < 2e-5, (what, err)


@pytest.mark.parametrize("shape,relu,res,train", [
    ((4, 64, 9, 11), True, False, True),
    ((2, 256, 7, 5), True, True, True),
    ((2, 6, 5, 5), False, False, True),      # C not multiple of 4
    ((3, 128, 6, 6), True, True, False),     # eval-mode statistics
    ((2, 256, 124, 1), True, False, True),   # OCR proxy shape
])
def test_batchnorm_act(dev, shape, relu, res, train):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(7)
    n, c, h, w = shape
    x = torch.randn(shape, generator=g) * 2 + 0.5
    r = torch.randn(shape, generator=g) if res else None
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g)
    rm = torch.randn(c, generator=g) * 0.1
    rv = torch.rand(c, generator=g) + 0.5
    gy = torch.randn(shape, generator=g)

    xr = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    y = F.batch_norm(xr, rm_r, rv_r, gr, br, train, 0.1, 1e-5)
    if res:
        y = y + rr
    if relu:
        y = F.relu(y)
    y.backward(gy)

    xd = x.to(dev).requires_grad_(True)
    rd = r.to(dev).requires_grad_(True) if res else None
    gd = gamma.to(dev).requires_grad_(True)
    bd = beta.to(dev).requires_grad_(True)
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    yd = ops.batch_norm_act(xd, gd, bd, rm_d, rv_d, rd, None, train, 0.1, 1e-5, relu)
    yd.backward(gy.to(dev))
    _close(yd, y, 2e-5, "bn fwd")
    _close(xd.grad, xr.grad, 5e-5, "bn dx")
    _close(gd.grad, gr.grad, 5e-5, "bn dgamma")
    _close(bd.grad, br.grad, 5e-5, "bn dbeta")
    if res:
        _close(rd.grad, rr.grad, 1e-6, "bn dres")
    _close(rm_d, rm_r, 1e-5, "running_mean")
    _close(rv_d, rv_r, 1e-5, "running_var")


@pytest.mark.parametrize("fused", [False, True])
def test_batchnorm_tiny_population_with_nearly_equal_samples(dev, fused):
    """PPM scale-1 branch: training-mode BatchNorm over B = 2 pooled vectors that nearly coincide (|a - b| ~ 1e-3 |a|;
    reference models/clip_psp.py:45-56).  The variance is (a - b)^2 / 4, 1e6 times smaller than E[x^2]: a one-pass
    E[x^2] - mean^2 over fp32 partials is off by percents there; with the two-pass fp64 statistics
    (vspw_bn_small_finalize) the result is as close to float64 torch as float32 torch is (x2)."""
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(11)
    c = 512
    a = torch.randn(1, c, 1, 1, generator=g) * 3
    x = torch.cat([a, a * (1 + 1e-3 * torch.randn(1, c, 1, 1, generator=g))], 0)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    ref = F.batch_norm(x.double(), None, None, gamma.double(), beta.double(), True, 0.1, 1e-5)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    if fused:  # through the conv + BN node: identity 1x1 convolution in front
        w = torch.eye(c).reshape(c, c, 1, 1).to(dev)
        y = ops.conv_bn_act(x.to(dev), w, None, gamma.to(dev), beta.to(dev), rm, rv, None, None, 1, 0, 1, True, 0.1,
                            1e-5, False)
    else:
        y = ops.batch_norm_act(x.to(dev), gamma.to(dev), beta.to(dev), rm, rv, None, None, True, 0.1, 1e-5, False)
    err = (y.double().cpu() - ref).norm() / ref.norm()
    # yardstick: float32 storage of the mean alone costs |mean| / std x 6e-8 (ATen's own float32 result, below)
    e32 = (F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5).double() - ref).norm() / ref.norm()
    assert err < max(2.0 * e32, 1e-5), (err, e32)
    var_u = x.double().var(0, unbiased=True).flatten()
    assert ((rv.double().cpu() - (0.9 + 0.1 * var_u)).abs() / (0.9 + 0.1 * var_u)).max() < 1e-6


def test_bn_finalize_clamped_is_the_reference_multi_device_formula(dev):
    """vspw_bn_finalize_clamped: invstd = clamp(var_biased, eps)^-1/2 (reference models/sync_batchnorm/batchnorm.py:150,
    the multi-device path), against the plain (var + eps)^-1/2 of vspw_bn_finalize, on channels with variances on both
    sides of eps; running statistics are the same in both."""
    import ctypes

    from cvpr2021_vspw_implement_amd import _C

    c, count, eps = 8, 50.0, 1e-5
    mean = torch.linspace(-1, 1, c, dtype=torch.float64)
    var = torch.tensor([0.0, 1e-7, 5e-6, 1e-5, 2e-5, 1e-3, 0.5, 3.0], dtype=torch.float64)
    sums = torch.stack([mean * count, (var + mean * mean) * count]).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {}
    for name in ("vspw_bn_finalize", "vspw_bn_finalize_clamped"):
        g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        co = torch.empty(4, c, device=dev)
        _C.call(name, sums.data_ptr(), ctypes.c_double(count), g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1,
                eps, co[0].data_ptr(), co[1].data_ptr(), co[2].data_ptr(), co[3].data_ptr(), c, st)
        outs[name] = (co.double().cpu(), rm.double().cpu(), rv.double().cpu())
    plain, clamped = outs["vspw_bn_finalize"], outs["vspw_bn_finalize_clamped"]
    assert torch.allclose(plain[0][1], (var + eps).rsqrt(), rtol=1e-5)           # sums carry ~1e-9 cancellation noise
    assert torch.allclose(clamped[0][1], var.clamp(min=eps).rsqrt(), rtol=1e-4)
    assert torch.equal(plain[1], clamped[1]) and torch.equal(plain[2], clamped[2])
    assert torch.allclose(plain[0][0], mean, atol=1e-6)


def test_conv_bn_act_fused_and_dropout(dev):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(11)
    n, c, h, w, k = 3, 64, 13, 15, 96
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, 3, 3, generator=g) * 0.05
    cb = torch.randn(k, generator=g)
    gamma = torch.rand(k, generator=g) + 0.5
    beta = torch.randn(k, generator=g)
    res = torch.randn(n, k, h, w, generator=g)
    mask = (torch.rand(n, k, generator=g) < 0.8).float() / 0.8
    gy = torch.randn(n, k, h, w, generator=g)

    leaves = [t.clone().requires_grad_(True) for t in (x, wt, cb, gamma, beta, res)]
    xr, wr, cbr, gr, br, rr = leaves
    rm, rv = torch.zeros(k), torch.ones(k)
    y = F.conv2d(xr, wr, cbr, padding=2, dilation=2)
    y = F.relu(F.batch_norm(y, rm, rv, gr, br, True, 0.1, 1e-5) + rr) * mask[:, :, None, None]
    y.backward(gy)

    dl = [t.to(dev).requires_grad_(True) for t in (x, wt.contiguous(memory_format=torch.channels_last), cb, gamma,
                                                   beta, res)]
    xd, wd, cbd, gd, bd, rd = dl
    rmd, rvd = torch.zeros(k, device=dev), torch.ones(k, device=dev)
    yd = ops.conv_bn_act(xd, wd, cbd, gd, bd, rmd, rvd, rd, mask.to(dev), 1, 2, 2, True, 0.1, 1e-5, True)
    yd.backward(gy.to(dev))
    _close(yd, y, 1e-4, "fused fwd")
    for a, b, nm in zip(dl, leaves, ("dx", "dw", "dcbias", "dgamma", "dbeta", "dres")):
        _close(a.grad, b.grad, 3e-4, "fused " + nm)
    _close(rmd, rm, 1e-5, "fused running_mean")
    _close(rvd, rv, 1e-5, "fused running_var")


@pytest.mark.parametrize("shape", [(2, 8, 33, 35), (1, 128, 240, 240), (2, 64, 16, 17)])
def test_maxpool(dev, shape):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(3)
    x = F.relu(torch.randn(shape, generator=g))  # many exact-zero ties, like post-ReLU activations
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.to(dev).requires_grad_(True)
    yd = ops.max_pool3x3s2(xd)
    yd.backward(gy.to(dev))
    _close(yd, y, 0.0, "maxpool fwd")
    _close(xd.grad, xr.grad, 1e-6, "maxpool bwd")


@pytest.mark.parametrize("hw,T", [((60, 60), 1), ((60, 107), 1), ((9, 13), 3), ((8, 12), 5)])
def test_pyramid_pool_temporal_mean(dev, hw, T):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(5)
    B, c = 2, 32
    h, w = hw
    x = torch.randn(T * B, c, h, w, generator=g)
    scales = (1, 2, 3, 6)
    xr = x.clone().requires_grad_(True)
    outs_r = []
    for s in scales:
        p = F.adaptive_avg_pool2d(xr, s)
        chunks = torch.split(p, B, dim=0)
        feats = [chunks[-1].unsqueeze(-1)] + [ch.unsqueeze(-1) for ch in chunks[:-1]]
        outs_r.append(torch.mean(torch.cat(feats, -1), -1))
    gys = [torch.randn(o.shape, generator=g) for o in outs_r]
    torch.autograd.backward(outs_r, gys)
    xd = x.to(dev).requires_grad_(True)
    outs_d = ops.pyramid_pool(xd, scales, T, None)
    torch.autograd.backward(list(outs_d), [gy.to(dev) for gy in gys])
    for a, b, s in zip(outs_d, outs_r, scales):
        _close(a, b, 1e-5, "pool scale %d" % s)
    _close(xd.grad, xr.grad, 1e-5, "pool bwd")


@pytest.mark.parametrize("ih,iw,oh,ow", [(6, 6, 60, 107), (1, 1, 9, 13), (60, 107, 480, 853), (3, 3, 8, 12), (2, 2, 60, 60)])
def test_bilinear_and_ppm_concat(dev, ih, iw, oh, ow):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(9)
    n, c = 2, 8
    if oh * ow > 100000:
        n, c = 1, 4
    x = torch.randn(n, c, ih, iw, generator=g)
    xr = x.clone().requires_grad_(True)
    y = F.interpolate(xr, (oh, ow), mode="bilinear", align_corners=False)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.to(dev).requires_grad_(True)
    yd = ops.interpolate_bilinear(xd, (oh, ow))
    yd.backward(gy.to(dev))
    _close(yd, y, 1e-5, "bilinear fwd")
    _close(xd.grad, xr.grad, 1e-4, "bilinear bwd")


def test_ppm_concat(dev):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(10)
    n, c, h, w = 2, 16, 12, 17
    x = torch.randn(n, c, h, w, generator=g)
    brs = [torch.randn(n, 8, s, s, generator=g) for s in (1, 2, 3, 6)]
    leaves = [t.clone().requires_grad_(True) for t in [x] + brs]
    y = torch.cat([leaves[0]] + [F.interpolate(b, (h, w), mode="bilinear", align_corners=False) for b in leaves[1:]], 1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dl = [t.to(dev).requires_grad_(True) for t in [x] + brs]
    yd = ops.ppm_concat(dl[0], dl[1:])
    yd.backward(gy.to(dev))
    _close(yd, y, 1e-5, "ppm concat fwd")
    for a, b in zip(dl, leaves):
        _close(a.grad, b.grad, 1e-4, "ppm concat bwd")


@pytest.mark.parametrize("hw,HW,k", [((8, 8), (65, 65), 124), ((60, 60), (479, 479), 124), ((8, 12), (64, 96), 7)])
def test_seg_nll(dev, hw, HW, k):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(21)
    n = 2
    h, w = hw
    H, W = HW
    logits = torch.randn(n, k, h, w, generator=g) * 3
    label = torch.randint(0, k, (n, 1, H, W), generator=g).float()
    label[torch.rand(n, 1, H, W, generator=g) < 0.05] = 255.0
    lr = logits.clone().requires_grad_(True)
    pred = F.interpolate(F.log_softmax(lr, dim=1), (H, W), mode="bilinear", align_corners=False)
    lab = label.squeeze(1).long()
    loss_r = F.nll_loss(pred, lab, ignore_index=255)
    preds = pred.argmax(1)
    acc_r = ((preds == lab) & (lab >= 0)).sum().float() / ((lab >= 0).sum().float() + 1e-10)
    (loss_r * 1.7).backward()
    ld = logits.to(dev).requires_grad_(True)
    loss_d, acc_d = ops.seg_nll(ld, label.to(dev), 255, True, True)
    (loss_d * 1.7).backward()
    _close(loss_d, loss_r, 1e-5, "nll loss")
    assert abs(acc_d.item() - acc_r.item()) < 2e-4, (acc_d.item(), acc_r.item())
    _close(ld.grad, lr.grad, 2e-5, "nll dlogits")
    # log-prob input variant (per-frame SegmentationModule path)
    lp = F.log_softmax(logits, dim=1)
    lpr = lp.clone().requires_grad_(True)
    F.nll_loss(F.interpolate(lpr, (H, W), mode="bilinear", align_corners=False), lab, ignore_index=255).backward()
    lpd = lp.to(dev).requires_grad_(True)
    l2, _ = ops.seg_nll(lpd, label.to(dev), 255, False, False)
    l2.backward()
    _close(l2, loss_r, 1e-5, "nll loss (logp)")
    _close(lpd.grad, lpr.grad, 2e-5, "nll dlogp")


def test_softmaxes_and_upsample_softmax(dev):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 124, 9, 13, generator=g) * 4
    for log in (True, False):
        xr = x.clone().requires_grad_(True)
        y = F.log_softmax(xr, 1) if log else F.softmax(xr, 1)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        xd = x.to(dev).requires_grad_(True)
        yd = ops.log_softmax_channels(xd) if log else ops.softmax_channels(xd)
        yd.backward(gy.to(dev))
        _close(yd, y, 1e-5, "softmax fwd log=%s" % log)
        _close(xd.grad, xr.grad, 1e-5, "softmax bwd log=%s" % log)
    a = torch.randn(2, 117, 124, generator=g)
    ar = a.clone().requires_grad_(True)
    y = F.softmax(0.0625 * ar, -1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    ad = a.to(dev).requires_grad_(True)
    yd = ops.row_softmax(ad, 0.0625)
    yd.backward(gy.to(dev))
    _close(yd, y, 1e-6, "row softmax")
    _close(ad.grad, ar.grad, 1e-6, "row softmax bwd")
    ar = a.clone().requires_grad_(True)
    y = F.softmax(ar, 1)
    y.backward(gy)
    ad = a.to(dev).requires_grad_(True)
    yd = ops.pixel_softmax(ad, 1.0)
    yd.backward(gy.to(dev))
    _close(yd, y, 1e-6, "pixel softmax")
    _close(ad.grad, ar.grad, 1e-6, "pixel softmax bwd")
    logits = torch.randn(1, 124, 8, 12, generator=g) * 3
    pr = F.softmax(F.interpolate(logits, (64, 96), mode="bilinear", align_corners=False), 1)
    pd = ops.upsample_softmax(logits.to(dev), (64, 96))
    _close(pd, pr, 1e-5, "upsample softmax")
    assert (pd.argmax(1).cpu() == pr.argmax(1)).float().mean().item() > 0.999


@pytest.mark.parametrize("B,M,N,K", [(5, 900, 124, 256),    # OCR attention scale: interior tiles, 5 batches in the grid
                                     (3, 300, 256, 124),    # reduction over the 124 classes (K % 32 != 0: generic kernel)
                                     (4, 1000, 512, 64)])
def test_batched_gemms_at_ocr_sizes(dev, B, M, N, K):
    """vspw_bmm_nt / vspw_bmm_tn (batch = grid dimension) against torch.matmul per batch, forward and both gradients."""
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(43 + N)
    a = torch.randn(B, M, K, generator=g)
    bt = torch.randn(B, N, K, generator=g)
    gy = torch.randn(B, M, N, generator=g)
    ar, br = a.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    (ar @ br.transpose(1, 2)).backward(gy)
    ad, bd = a.to(dev).requires_grad_(True), bt.to(dev).requires_grad_(True)
    yd = ops.bmm_nt(ad, bd)
    yd.backward(gy.to(dev))
    tol = 3e-5 * (K / 64.0) ** 0.5
    _close(yd, a @ bt.transpose(1, 2), tol, "bmm_nt")
    _close(ad.grad, ar.grad, 3e-5 * (N / 64.0) ** 0.5, "bmm_nt da")
    _close(bd.grad, br.grad, 3e-5 * (M / 64.0) ** 0.5, "bmm_nt db")
    # a^T b with the long dimension (pixels) as the reduction: split-R partial sums, batched reduce
    pm = torch.randn(B, M, N, generator=g)
    f = torch.randn(B, M, K, generator=g)
    gz = torch.randn(B, N, K, generator=g)
    pr, fr = pm.clone().requires_grad_(True), f.clone().requires_grad_(True)
    (pr.transpose(1, 2) @ fr).backward(gz)
    pd, fd = pm.to(dev).requires_grad_(True), f.to(dev).requires_grad_(True)
    zd = ops.bmm_tn(pd, fd)
    zd.backward(gz.to(dev))
    _close(zd, pm.transpose(1, 2) @ f, 3e-5 * (M / 64.0) ** 0.5, "bmm_tn")
    _close(pd.grad, pr.grad, 3e-5 * (K / 64.0) ** 0.5, "bmm_tn da")
    _close(fd.grad, fr.grad, 3e-5 * (N / 64.0) ** 0.5, "bmm_tn db")


def test_bmm_and_transpose(dev):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(41)
    a = torch.randn(2, 117, 64, generator=g)
    bt = torch.randn(2, 124, 64, generator=g)
    ar, br = a.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    y = ar @ br.transpose(1, 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    ad, bd = a.to(dev).requires_grad_(True), bt.to(dev).requires_grad_(True)
    yd = ops.bmm_nt(ad, bd)
    yd.backward(gy.to(dev))
    _close(yd, y, 2e-5, "bmm_nt")
    _close(ad.grad, ar.grad, 2e-5, "bmm_nt da")
    _close(bd.grad, br.grad, 2e-5, "bmm_nt db")
    p = torch.randn(2, 117, 124, generator=g)
    f = torch.randn(2, 117, 32, generator=g)
    pr, fr = p.clone().requires_grad_(True), f.clone().requires_grad_(True)
    y = pr.transpose(1, 2) @ fr
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    pd, fd = p.to(dev).requires_grad_(True), f.to(dev).requires_grad_(True)
    yd = ops.bmm_tn(pd, fd)
    yd.backward(gy.to(dev))
    _close(yd, y, 2e-5, "bmm_tn")
    _close(pd.grad, pr.grad, 2e-5, "bmm_tn da")
    _close(fd.grad, fr.grad, 2e-5, "bmm_tn db")
    _close(ops.transpose_last2(a.to(dev)), a.transpose(1, 2).contiguous(), 0.0, "transpose")


def test_flowwarp_and_blend(dev):
    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(51)
    B, C, H, W = 2, 40, 15, 19
    x = torch.randn(B, C, H, W, generator=g)
    flo = (torch.randn(B, 2, H, W, generator=g) * 1.9 - 0.7).clamp(-10, 10)
    flo[0, :, 0, 0] = torch.tensor([-30.0, 25.0])  # out-of-bounds taps

    def ref_warp(x, flo):
        xx = torch.arange(0, W).view(1, -1).repeat(H, 1).view(1, 1, H, W).repeat(B, 1, 1, 1)
        yy = torch.arange(0, H).view(-1, 1).repeat(1, W).view(1, 1, H, W).repeat(B, 1, 1, 1)
        vgrid = torch.cat((xx, yy), 1).float() + flo
        vx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0
        vy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0
        return F.grid_sample(x, torch.stack((vx, vy), -1), align_corners=False)

    xr, fr = x.clone().requires_grad_(True), flo.clone().requires_grad_(True)
    y = ref_warp(xr, fr)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, fd = x.to(dev).requires_grad_(True), flo.to(dev).requires_grad_(True)
    yd = ops.flowwarp(xd, fd)
    yd.backward(gy.to(dev))
    _close(yd, y, 1e-5, "flowwarp fwd")
    _close(xd.grad, xr.grad, 1e-5, "flowwarp dx")
    _close(fd.grad, fr.grad, 1e-4, "flowwarp dflow")
    w0, w1 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    b = torch.randn(B, C, H, W, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, b, w0, w1)]
    out = leaves[2].view(1, -1, 1, 1) * leaves[0] + leaves[3].view(1, -1, 1, 1) * leaves[1]
    out.backward(gy)
    dl = [t.to(dev).requires_grad_(True) for t in (x, b, w0, w1)]
    od = ops.chan_blend(*dl)
    od.backward(gy.to(dev))
    _close(od, out, 1e-6, "blend fwd")
    for a, r in zip(dl, leaves):
        _close(a.grad, r.grad, 2e-5, "blend bwd")


def test_sgd_matches_loop_semantics_with_duplicate_params(dev):
    """vspw SGD == torch.optim.SGD(1.3.1) applied as a Python loop over the reference's parameter-group listings,
    which contain each parameter once per enclosing module (train_clip2.py:215-236 + models/clip_psp.py:99-135)."""
    from cvpr2021_vspw_implement_amd import optim

    g = torch.Generator().manual_seed(77)
    shapes = [(64, 3, 3, 3), (64,), (70000,), (5, 7), (128, 64, 1, 1)]
    mults = [4, 2, 1, 3, 4]
    base = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(3)]
    lr, wd, mom = 0.02, 1e-4, 0.9

    # reference: loop semantics on CPU
    ref = [b.clone() for b in base]
    bufs = [None] * len(ref)
    for step in range(3):
        for i, p in enumerate(ref):
            this_wd = wd if i % 2 == 0 else 0.0
            this_lr = lr * (0.1 if i < 2 else 1.0)
            d = grads[step][i].clone()  # torch 1.3.1: d_p = p.grad.data, then d_p.add_(weight_decay, p.data) IN PLACE,
            for _ in range(mults[i]):   # so the decay term accumulates over the duplicate occurrences of a parameter
                d += this_wd * p
                bufs[i] = d.clone() if bufs[i] is None else bufs[i] * mom + d
                p -= this_lr * bufs[i]

    params = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
    params[0].data = params[0].data.contiguous(memory_format=torch.channels_last)
    groups = []
    for i, p in enumerate(params):
        groups.append({"params": [p] * mults[i], "lr": lr * (0.1 if i < 2 else 1.0),
                       "weight_decay": wd if i % 2 == 0 else 0.0})
    opt = optim.SGD(groups, lr=lr, momentum=mom, weight_decay=wd)
    for step in range(3):
        for i, p in enumerate(params):
            p.grad = grads[step][i].to(dev)
        opt.step()
    for p, r in zip(params, ref):
        _close(p, r, 2e-6, "sgd param")


@pytest.mark.parametrize("B,N,C", [(2, 200, 128), (1, 1111, 64), (3, 17, 32), (2, 4097, 128)])
def test_non_local_dot_fused(dev, B, N, C):
    """(theta phi^T / N) g without the N x N affinity, forward and the three gradients, against the two-matmul
    formulation of the reference (models/non_local.py:116-133) evaluated by torch on the CPU in float64; ragged N
    (not a multiple of the 128-query / 32-key tiles), several key chunks (N = 4097)."""
    from cvpr2021_vspw_implement_amd import ops

    g_ = torch.Generator().manual_seed(B * 1000 + N + C)
    th, ph, gg = (torch.randn(B, N, C, generator=g_) for _ in range(3))
    dy = torch.randn(B, N, C, generator=g_)
    leaves = [t.double().requires_grad_(True) for t in (th, ph, gg)]
    f = torch.matmul(leaves[0], leaves[1].transpose(1, 2)) / N
    y = torch.matmul(f, leaves[2])
    y.backward(dy.double())
    dl = [t.to(dev).requires_grad_(True) for t in (th, ph, gg)]
    yd = ops.non_local_dot(dl[0], dl[1], dl[2], 1.0 / N)
    yd.backward(dy.to(dev))
    _close(yd, y, 2e-5, "nl fwd")
    for a, r, nm in zip(dl, leaves, ("theta", "phi", "g")):
        _close(a.grad, r.grad, 2e-5, "nl d" + nm)
    # bit-reproducible: fixed summation order, no atomics
    yd2 = ops.non_local_dot(dl[0], dl[1], dl[2], 1.0 / N)
    assert torch.equal(yd, yd2)


@pytest.mark.parametrize("shape", [(2, 8, 10, 10), (3, 256, 9, 13), (1, 4, 2, 2)])
def test_avg_pool2x2_matches_aten_semantics(dev, shape):
    """F.avg_pool2d(x, (2, 2)) (models/non_local_models.py:32,136): floor output size (odd trailing row / column dropped),
    forward bit-exact against the window sum in ATen's order / 4, adjoint = dy / 4 scattered to the four taps."""
    from cvpr2021_vspw_implement_amd import ops

    n, c, h, w = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g)
    dy = torch.randn(n, c, h // 2, w // 2, generator=g)
    xd = x.to(dev).requires_grad_(True)
    y = ops.avg_pool2x2(xd)
    y.backward(dy.to(dev))
    xe = x[:, :, :h // 2 * 2, :w // 2 * 2]
    want = (((xe[:, :, 0::2, 0::2] + xe[:, :, 0::2, 1::2]) + xe[:, :, 1::2, 0::2]) + xe[:, :, 1::2, 1::2]) * 0.25
    assert torch.equal(y.detach().cpu().contiguous(), want)
    dx = torch.zeros_like(x)
    dx[:, :, :h // 2 * 2, :w // 2 * 2] = (dy * 0.25).repeat_interleave(2, 2).repeat_interleave(2, 3)
    assert torch.equal(xd.grad.cpu().contiguous(), dx)


@pytest.mark.parametrize("hw,out", [((479, 479), (60, 60)), ((480, 853), (60, 107)), ((7, 9), (15, 20)), ((60, 60), (60, 60))])
def test_flow_plumbing_gathers_match_aten(dev, hw, out):
    """The three data-movement ops around the flow network, bit-exact against the ATen calls the reference makes
    (evaluated here by torch on the CPU): F.interpolate(flow, size, mode='nearest') forward AND adjoint
    (models/netwarp.py:199,214), F.pad(mode='constant') / the unpad crop of RAFT's InputPadder (utils/utils.py:7-25),
    (img * std + mean) * 255 (models/netwarp.py:186-187)."""
    import torch.nn.functional as F

    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(11)
    h, w = hw
    flow = torch.randn(2, 2, h, w, generator=g)
    dy = torch.randn(2, 2, *out, generator=g)
    ref = flow.clone().requires_grad_(True)
    want = F.interpolate(ref, out, mode="nearest")
    want.backward(dy)
    x = flow.to(dev).requires_grad_(True)
    got = ops.nearest_resize(x, out)
    got.backward(dy.to(dev))
    assert torch.equal(got.detach().cpu(), want.detach())
    assert torch.allclose(x.grad.cpu(), ref.grad, rtol=0, atol=1e-6)  # (sums of <= 4 terms when up-sampling)
    img = torch.randn(2, 3, h, w, generator=g)
    ph, pw = (((h // 8) + 1) * 8 - h) % 8, (((w // 8) + 1) * 8 - w) % 8
    pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    padded = ops.plane_shift(img.to(dev), (h + ph, w + pw), pad[2], pad[0])
    assert torch.equal(padded.cpu(), F.pad(img, pad, mode="constant"))
    back = ops.plane_shift(padded, (h, w), -pad[2], -pad[0])
    assert torch.equal(back.cpu(), img)
    from cvpr2021_vspw_implement_amd.RAFT_core.utils.utils import InputPadder

    for mode in ("sintel", "kitti"):  # the import-surface mirror of the reference's class, on the same gather
        ip = InputPadder(img.shape, mode)
        ref_pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2] if mode == "sintel" else [pw // 2, pw - pw // 2, 0, ph]
        assert ip._pad == ref_pad
        pp = ip.pad(img.to(dev))
        assert torch.equal(pp.cpu(), F.pad(img, ref_pad, mode="constant")) and torch.equal(ip.unpad(pp).cpu(), img)
    mean, std = torch.FloatTensor([0.485, 0.456, 0.406]), torch.FloatTensor([0.229, 0.224, 0.225])
    un = ops.unnormalize_rgb(img.to(dev), std.tolist(), mean.tolist(), 255.0)
    assert torch.equal(un.cpu(), (img * std.view(1, 3, 1, 1) + mean.view(1, 3, 1, 1)) * 255.0)


@pytest.mark.parametrize("tiles,c,training", [(375, 1024, 1), (94, 256, 1), (7, 40, 1), (33, 512, 0)])
def test_bn_backward_partial_reduction_with_coefficients_is_bit_identical(dev, tiles, c, training):
    """vspw_bn_bwd_reduce_partials_coeffs_f32 = vspw_bn_bwd_reduce_partials_f32 followed by vspw_bn_bwd_affine_coeffs, in
    one launch: sums (fp64), dgamma / dbeta and the three coefficient rows must be the SAME bits."""
    from cvpr2021_vspw_implement_amd import _C
    from cvpr2021_vspw_implement_amd.ops import _p, _stream

    g = torch.Generator().manual_seed(tiles + c)
    part = torch.randn(tiles, 2, c, generator=g).to(dev)
    gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
    mean, invstd = torch.randn(c, generator=g).to(dev), (torch.rand(c, generator=g) + 0.5).to(dev)
    count = 36000.0
    st = _stream()
    s0 = torch.empty(2, c, device=dev, dtype=torch.float64)
    dg0, db0, k0 = torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(3, c, device=dev)
    _C.call("vspw_bn_bwd_reduce_partials_f32", _p(part), tiles, c, _p(s0), _p(dg0), _p(db0), st)
    _C.call("vspw_bn_bwd_affine_coeffs", _p(s0), ctypes.c_double(count), _p(gamma), _p(mean), _p(invstd), _p(k0), c, training, st)
    s1 = torch.empty(2, c, device=dev, dtype=torch.float64)
    dg1, db1, k1 = torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(3, c, device=dev)
    _C.call("vspw_bn_bwd_reduce_partials_coeffs_f32", _p(part), tiles, c, ctypes.c_double(count), _p(gamma), _p(mean),
            _p(invstd), training, _p(s1), _p(dg1), _p(db1), _p(k1), st)
    torch.cuda.synchronize()
    for a, b, what in ((s0, s1, "sums"), (dg0, dg1, "dgamma"), (db0, db1, "dbeta"), (k0, k1, "coef")):
        assert torch.equal(a, b), what
    ref = part.double().sum(0)
    assert torch.allclose(s1, ref, rtol=1e-12, atol=1e-12)


def test_inference_fold_cache_is_not_served_to_a_new_model_in_recycled_storage(dev):
    """Regression (round 5, seen once in the full GPU suite): the folded conv+BN weights of inference were cached under
    (id, address, version) of the tensors they derive from - identities that a NEW model built after the old one was
    dropped inherits together with its storage.  Six generations of same-shape modules, each dropped before the next is
    built: the folded path must follow the unfolded one every time."""
    import gc

    from cvpr2021_vspw_implement_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 12, 13, generator=g).to(dev)
    for gen in range(6):
        w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
        gamma, beta = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
        rm, rv = torch.randn(64, generator=g).to(dev), (torch.rand(64, generator=g) + 0.5).to(dev)
        out = {}
        for folded in (True, False):
            ops.set_inference_folding(folded)
            try:
                with torch.no_grad():
                    out[folded] = ops.conv_bn_act(x, w, None, gamma, beta, rm, rv, pad=1, training=False, relu=True).cpu()
            finally:
                ops.set_inference_folding(True)
        err = float((out[True] - out[False]).abs().max())
        assert err < 1e-4 * float(out[False].abs().max() + 1.0), (gen, err)
        del w, gamma, beta, rm, rv, out
        gc.collect()
        torch.cuda.empty_cache() if gen % 2 else None


@pytest.mark.parametrize("n,oh,ow,h,w,c,stride", [(2, 11, 12, 21, 23, 8, 2), (1, 4, 5, 10, 13, 64, 3), (3, 7, 7, 7, 7, 4, 1)])
def test_strided_scatter_is_the_adjoint_of_strided_sampling(dev, n, oh, ow, h, w, c, stride):
    """vspw_strided_scatter_nhwc (second half of the strided pointwise data gradient): dst[:, s*oy, s*ox] = src, zero
    elsewhere - bit-exact against indexing, for odd extents and a trailing rim that no output pixel reaches."""
    from cvpr2021_vspw_implement_amd import _C
    from cvpr2021_vspw_implement_amd.ops import _p, _stream

    g = torch.Generator().manual_seed(h * w + c)
    src = torch.randn(n, oh, ow, c, generator=g)
    want = torch.zeros(n, h, w, c)
    want[:, ::stride, ::stride][:, :oh, :ow] = src
    dst = torch.full((n, h, w, c), -3.0, device=dev)
    _C.call("vspw_strided_scatter_nhwc", _p(src.to(dev)), _p(dst), n, oh, ow, h, w, c, stride, _stream())
    assert torch.equal(dst.cpu(), want)


def test_winograd_tile_choice_on_the_bench_geometry(dev):
    """ops._wino_f3: which Winograd tile a stride-1 3x3 of the bench workload (10 frames, 60x60) takes - the measured rule
    of profiles/r06_wino345_probe.log and r06_f5_eval.log: forward + weight gradient (they share V) F(5x5) from 512 channels
    on (heads, layer 4: their error is not amplified by many later blocks), F(4x4) on 256 channels with an exact 4-tiling,
    F(3x3) otherwise, nothing under 128 channels; the data gradient F(5x5) everywhere; the forced settings."""
    from cvpr2021_vspw_implement_amd import ops
    from cvpr2021_vspw_implement_amd._C import ConvDesc

    def m(c, k, dil, n=10, h=60, w=60, dgrad=False):
        d = ConvDesc(n, h, w, c, h, w, k, 3, 3, 1, dil, dil, dil)
        return ops._wino_f3(d, dgrad) if ops._wino_ok(d) else -1

    prev = ops.set_winograd_f3(True)
    try:
        assert m(256, 256, 2) == 3 and m(256, 256, 1) == 4 and m(128, 128, 1) == 3
        assert m(512, 512, 4) == 5 and m(512, 512, 2) == 5 and m(1024, 512, 1) == 5 and m(4096, 512, 1, n=2) == 5
        assert m(256, 256, 2, dgrad=True) == 5 and m(128, 128, 1, dgrad=True) == 5 and m(512, 512, 4, dgrad=True) == 5
        assert m(64, 64, 1) == -1                       # below VSPW_WINO_MINC: direct kernel
        assert m(2048, 512, 1, n=1, h=60, w=107) == 5   # the 480x853 inference frame
        ops.set_winograd_tile(3)
        assert m(512, 512, 4) == 3 and m(1024, 512, 1) == 3 and m(256, 256, 2, dgrad=True) == 3
        ops.set_winograd_tile(2)
        assert m(512, 512, 4) == 0 and m(256, 256, 2) == 0 and m(256, 256, 2, dgrad=True) == 0
    finally:
        ops.set_winograd_f3(prev)


def test_winograd_tile_hint_is_followed_by_forward_and_weight_gradient(dev):
    """ops.winograd_tile_hint (the model's "late block" request, models/models.py ResnetDilated): a 256-channel dilated 3x3
    - F(3x3) by the automatic rule - takes F(5x5) in the forward pass issued under the hint, its weight gradient follows
    through the kept V, the data gradient is F(5x5) anyway; results against F.conv2d on the CPU as in test_conv2d_fwd_bwd."""
    from cvpr2021_vspw_implement_amd import ops

    case = (2, 256, 30, 30, 256, 3, 1, 2, 2, False)  # 15-pixel sub-grids: F(3x3) by the automatic rule
    prev = ops.set_winograd_f3(True)
    try:
        keys = ("f3_launches", "f4_launches", "f5_launches")
        before = [ops._wino[q] for q in keys]
        test_conv2d_fwd_bwd(dev, case)
        assert tuple(ops._wino[q] - b for q, b in zip(keys, before)) == (2, 0, 1)
        before = [ops._wino[q] for q in keys]
        orig = ops.conv2d

        def hinted(*a, **k):
            with ops.winograd_tile_hint(5):
                return orig(*a, **k)

        ops.conv2d = hinted
        try:
            test_conv2d_fwd_bwd(dev, case)
        finally:
            ops.conv2d = orig
        assert tuple(ops._wino[q] - b for q, b in zip(keys, before)) == (0, 0, 3)
    finally:
        ops.set_winograd_f3(prev)
