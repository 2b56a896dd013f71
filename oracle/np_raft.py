"""ORACLE (test infrastructure, NOT the product): numpy restatement of the frozen RAFT flow network's forward pass,
RAFT_core/raft.py:75-127 as NetWarp calls it (models/netwarp.py:170-176: eval mode, iters=20, test_mode=True).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  Pinned by
tests/golden/raft_basic.npz (tests/golden/make_golden.py imports the reference's RAFT with oracle/det_init.py weights).
Layout: the reference's NCHW; `sd` is a {state_dict key: ndarray} mapping with the reference's key names.
"""
import numpy as np

F = np.float32


def conv2d(x, w, b=None, stride=1, pad=(0, 0)):
    """F.conv2d with per-axis zero padding (update.py:36-42 uses (1,5)/(5,1) kernels with (0,2)/(2,0) padding)."""
    k, c, kh, kw = w.shape
    n, _, h, wd = x.shape
    ph, pw = pad
    oh = (h + 2 * ph - kh) // stride + 1
    ow = (wd + 2 * pw - kw) // stride + 1
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    cols = np.empty((n, c, kh, kw, oh, ow), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            cols[:, :, ky, kx] = xp[:, :, ky:ky + stride * (oh - 1) + 1:stride, kx:kx + stride * (ow - 1) + 1:stride]
    a = cols.transpose(0, 4, 5, 1, 2, 3).reshape(n * oh * ow, c * kh * kw)
    y = a @ w.reshape(k, -1).T
    if b is not None:
        y = y + b
    return np.ascontiguousarray(y.reshape(n, oh, ow, k).transpose(0, 3, 1, 2))


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d defaults (no affine, batch statistics always; extractor.py:27-31,131)."""
    m = x.mean(axis=(2, 3), keepdims=True, dtype=np.float64)
    v = ((x - m) ** 2).mean(axis=(2, 3), keepdims=True, dtype=np.float64)
    return ((x - m) / np.sqrt(v + eps)).astype(x.dtype)


def batch_norm_eval(x, sd, prefix, eps=1e-5):
    g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    s = (g / np.sqrt(rv.astype(np.float64) + eps)).astype(x.dtype)
    return x * s[None, :, None, None] + (b - rm * s).astype(x.dtype)[None, :, None, None]


def relu(x):
    return np.maximum(x, 0)


def sigmoid(x):
    with np.errstate(over="ignore"):  # exp overflow -> inf -> 1/(1+inf) = 0, the correct limit
        return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def _norm(x, sd, prefix, kind):
    return instance_norm(x) if kind == "instance" else batch_norm_eval(x, sd, prefix)


def residual_block(x, sd, p, kind, stride):
    """extractor.py:44-56"""
    y = relu(_norm(conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride, (1, 1)), sd, p + ".norm1", kind))
    y = relu(_norm(conv2d(y, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], 1, (1, 1)), sd, p + ".norm2", kind))
    if stride != 1:
        # norm3 and downsample.1 are one module under two state_dict names (extractor.py:25,40-41); load_state_dict
        # copies both into the same storage, the later key (downsample.1) wins
        x = _norm(conv2d(x, sd[p + ".downsample.0.weight"], sd[p + ".downsample.0.bias"], stride, (0, 0)), sd,
                  p + ".downsample.1", kind)
    return relu(x + y)


def basic_encoder(x, sd, p, kind):
    """extractor.py:168-190 (eval mode: no dropout)"""
    x = relu(_norm(conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], 2, (3, 3)), sd, p + ".norm1", kind))
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = residual_block(x, sd, "%s.%s.0" % (p, name), kind, stride)
        x = residual_block(x, sd, "%s.%s.1" % (p, name), kind, 1)
    return conv2d(x, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], 1, (0, 0))


def grid_sample_ac(img, gx, gy):
    """F.grid_sample(bilinear, zeros, align_corners=True) on [P,1,H,W] planes at normalised coords [P,a,b]."""
    P, _, H, W = img.shape
    ix = ((gx + 1) / 2) * (W - 1)
    iy = ((gy + 1) / 2) * (H - 1)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    out = np.zeros(gx.shape, dtype=img.dtype)
    pidx = np.arange(P)[:, None, None]
    for dy in (0, 1):
        for dx in (0, 1):
            xx = x0 + dx
            yy = y0 + dy
            wx = (x0 + 1 - ix) if dx == 0 else (ix - x0)
            wy = (y0 + 1 - iy) if dy == 0 else (iy - y0)
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            xi = np.clip(xx, 0, W - 1).astype(np.int64)
            yi = np.clip(yy, 0, H - 1).astype(np.int64)
            out += np.where(ok, img[pidx, 0, yi, xi] * (wx * wy).astype(img.dtype), 0).astype(img.dtype)
    return out


class CorrBlock:
    """corr.py:12-62"""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        b, d, h, w = fmap1.shape
        f1 = fmap1.reshape(b, d, h * w)
        f2 = fmap2.reshape(b, d, h * w)
        corr = np.matmul(f1.transpose(0, 2, 1), f2) / np.sqrt(np.array(d, dtype=F)).astype(fmap1.dtype)
        corr = corr.reshape(b * h * w, 1, h, w)
        self.pyr = [corr]
        self.radius = radius
        for _ in range(num_levels - 1):
            hh, ww = corr.shape[2] // 2, corr.shape[3] // 2
            c = corr[:, :, :2 * hh, :2 * ww]
            corr = (c[:, :, 0::2, 0::2] + c[:, :, 0::2, 1::2] + c[:, :, 1::2, 0::2] + c[:, :, 1::2, 1::2]) * F(0.25)
            self.pyr.append(corr.astype(fmap1.dtype))

    def __call__(self, coords):
        r = self.radius
        b, _, h1, w1 = coords.shape
        c = coords.transpose(0, 2, 3, 1).reshape(b * h1 * w1, 1, 1, 2)
        d = np.linspace(-r, r, 2 * r + 1).astype(coords.dtype)
        # delta = stack(meshgrid(dy, dx), -1): component 0 varies along the FIRST window axis and is added to x
        delta = np.stack(np.meshgrid(d, d, indexing="ij"), axis=-1)[None]
        out = []
        for i, corr in enumerate(self.pyr):
            cl = c / (2 ** i) + delta
            H, W = corr.shape[-2:]
            gx = 2 * cl[..., 0] / (W - 1) - 1
            gy = 2 * cl[..., 1] / (H - 1) - 1
            s = grid_sample_ac(corr, gx.astype(coords.dtype), gy.astype(coords.dtype))
            out.append(s.reshape(b, h1, w1, -1))
        return np.ascontiguousarray(np.concatenate(out, axis=-1).transpose(0, 3, 1, 2))


def update_block(net, inp, corr, flow, sd, p="update_block"):
    """update.py:114-136 (BasicUpdateBlock) with BasicMotionEncoder (:79-96), SepConvGRU (:33-60), FlowHead (:6-14)."""
    e = p + ".encoder"
    cor = relu(conv2d(corr, sd[e + ".convc1.weight"], sd[e + ".convc1.bias"], 1, (0, 0)))
    cor = relu(conv2d(cor, sd[e + ".convc2.weight"], sd[e + ".convc2.bias"], 1, (1, 1)))
    flo = relu(conv2d(flow, sd[e + ".convf1.weight"], sd[e + ".convf1.bias"], 1, (3, 3)))
    flo = relu(conv2d(flo, sd[e + ".convf2.weight"], sd[e + ".convf2.bias"], 1, (1, 1)))
    out = relu(conv2d(np.concatenate([cor, flo], 1), sd[e + ".conv.weight"], sd[e + ".conv.bias"], 1, (1, 1)))
    x = np.concatenate([inp, out, flow], 1)
    g = p + ".gru"
    h = net
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = np.concatenate([h, x], 1)
        z = sigmoid(conv2d(hx, sd[g + ".convz" + sfx + ".weight"], sd[g + ".convz" + sfx + ".bias"], 1, pad))
        r = sigmoid(conv2d(hx, sd[g + ".convr" + sfx + ".weight"], sd[g + ".convr" + sfx + ".bias"], 1, pad))
        q = np.tanh(conv2d(np.concatenate([r * h, x], 1), sd[g + ".convq" + sfx + ".weight"],
                           sd[g + ".convq" + sfx + ".bias"], 1, pad))
        h = (1 - z) * h + z * q
    f = p + ".flow_head"
    d = conv2d(relu(conv2d(h, sd[f + ".conv1.weight"], sd[f + ".conv1.bias"], 1, (1, 1))), sd[f + ".conv2.weight"],
               sd[f + ".conv2.bias"], 1, (1, 1))
    m = relu(conv2d(h, sd[p + ".mask.0.weight"], sd[p + ".mask.0.bias"], 1, (1, 1)))
    m = conv2d(m, sd[p + ".mask.2.weight"], sd[p + ".mask.2.bias"], 1, (0, 0)) * F(0.25)
    return h, m.astype(h.dtype), d


def upsample_flow(flow, mask):
    """raft.py:57-68"""
    n, _, h, w = flow.shape
    m = mask.reshape(n, 1, 9, 8, 8, h, w)
    m = np.exp(m - m.max(axis=2, keepdims=True))
    m = m / m.sum(axis=2, keepdims=True)
    fp = np.pad(8 * flow, ((0, 0), (0, 0), (1, 1), (1, 1)))
    nb = np.stack([fp[:, :, ky:ky + h, kx:kx + w] for ky in range(3) for kx in range(3)], axis=2)  # [n,2,9,h,w]
    up = (m * nb[:, :, :, None, None]).sum(axis=2)  # [n,2,8,8,h,w]
    return np.ascontiguousarray(up.transpose(0, 1, 4, 2, 5, 3).reshape(n, 2, 8 * h, 8 * w)).astype(flow.dtype)


def coords_grid(n, h, w, dtype):
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    return np.broadcast_to(np.stack([xs, ys], 0).astype(dtype)[None], (n, 2, h, w)).copy()


def raft_forward(sd, image1, image2, iters=20, trace=None):
    """RAFT.forward(..., test_mode=True) -> (flow at 1/8 resolution, convex-upsampled flow); images in [0, 255]."""
    dt = image1.dtype
    im1 = 2 * (image1 / dt.type(255.0)) - 1
    im2 = 2 * (image2 / dt.type(255.0)) - 1
    f = basic_encoder(np.concatenate([im1, im2], 0), sd, "fnet", "instance")
    n = image1.shape[0]
    fmap1, fmap2 = f[:n], f[n:]
    corr_fn = CorrBlock(fmap1, fmap2)
    c = basic_encoder(im1, sd, "cnet", "batch")
    net, inp = np.tanh(c[:, :128]), relu(c[:, 128:])
    h, w = image1.shape[2] // 8, image1.shape[3] // 8
    coords0 = coords_grid(n, h, w, dt)
    coords1 = coords0.copy()
    if trace is not None:
        trace["fmap1"] = fmap1
        trace["cnet"] = c
        trace["corr0"] = corr_fn(coords1)
    mask = None
    for _ in range(iters):
        corr = corr_fn(coords1)
        net, mask, delta = update_block(net, inp, corr, coords1 - coords0, sd)
        coords1 = coords1 + delta
        if trace is not None:
            trace.setdefault("flows", []).append(coords1 - coords0)
    flow = coords1 - coords0
    return flow, upsample_flow(flow, mask)
