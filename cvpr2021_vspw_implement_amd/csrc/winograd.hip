// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the path (forward and data gradient).
//
// 59 % of the FLOPs of a TCB-PSP step are stride-1 3x3 convolutions (reference models/resnet.py:63-64 after the
// dilation rewrite of models/models.py:737-750, heads models/clip_psp.py:29-35,74-79) and the direct implicit GEMM
// already runs them at 0.8-0.9 of the fp32 MFMA peak: only fewer multiplications help.  F(2x2, 3x3) computes a 2x2
// output tile from a 4x4 input patch with 16 multiplications per (cin, cout) pair instead of 36:
//     Y = A^T [ (G g G^T) o (B^T d B) ] A,
// i.e. per transform position xi = (a, b) one plain GEMM  M[xi] = V[xi] (tiles x Cin) . U[xi]^T (Cout x Cin):
// 16 batched GEMMs with 4/9 of the direct FLOPs, run by the SAME MFMA kernel as the pointwise convolutions
// (vspw_bmm_nt).  The transforms only add / subtract (B, A: entries 0, +-1) or halve (G): in fp32 the result is
// *closer* to exact than the direct convolution, whose accumulation chain is 9x longer (measured with the oracle,
// decisions pinned: 4.4e-4 against 8.2e-4 median gradient error).
//
// Dilation d: a stride-1 convolution with dilation d is d*d independent UNdilated convolutions on the sub-grids
// {(y, x): y = sy (mod d), x = sx (mod d)}; tiles are laid out per (image, sub-grid).  Ragged edges (odd sub-grid
// sizes; 15x15 sub-grids of the dilation-4 stage) are zero-padded tiles whose surplus outputs are dropped.
//
// Layouts: activations NHWC; V [16][T][Cin], U [16][Cout][Cin], M [16][T][Cout], T = n * d*d * th * tw tiles.
// The data gradient is the same convolution applied to dY with the filter rotated by 180 degrees and its channel axes
// swapped; only the weight transform differs.
#include "common.h"

struct WinoGeom {
    int n, h, w, d;       // images, height, width, dilation
    int th, tw, tpi, T;   // tiles per sub-grid column / row, tiles per image, total
};

static bool wino_geom(const vspw_conv_desc* dsc, WinoGeom& g) {
    if (!dsc || dsc->kh != 3 || dsc->kw != 3 || dsc->stride != 1 || dsc->dil < 1 || dsc->pad != dsc->dil ||
        dsc->pad_w != dsc->dil || dsc->oh != dsc->h || dsc->ow != dsc->w || dsc->n < 1)
        return false;
    g.n = dsc->n; g.h = dsc->h; g.w = dsc->w; g.d = dsc->dil;
    const int hs = (g.h + g.d - 1) / g.d, ws = (g.w + g.d - 1) / g.d;
    g.th = (hs + 1) / 2;
    g.tw = (ws + 1) / 2;
    const long long tpi = (long long)g.d * g.d * g.th * g.tw;
    if (tpi * g.n > 0x3fffffffLL) return false;
    g.tpi = (int)tpi;
    g.T = (int)(tpi * g.n);
    return true;
}

__device__ __forceinline__ void wino_tile(const WinoGeom& g, int t, int& img, int& sy, int& sx, int& ty, int& tx) {
    img = t / g.tpi;
    int r = t - img * g.tpi;
    const int per = g.th * g.tw;
    const int sg = r / per;
    r -= sg * per;
    sy = sg / g.d;
    sx = sg - sy * g.d;
    ty = r / g.tw;
    tx = r - ty * g.tw;
}

// ------------------------------------------------------------------------------------------------ weights
// U = G g G^T with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].  w: [K][3][3][C] (channels_last OIHW).
// mode 0 (forward):        U[xi][k][c] from g = w[k, :, :, c]
// mode 1 (data gradient):  U[xi][c][k] from g = w[k, 2-ky, 2-kx, c]   (rows = Cin, reduction over Cout)
__device__ __forceinline__ void wino_g(const float g[9], float u[16]) {
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        t[0][j] = g0;
        t[1][j] = 0.5f * (g0 + g1 + g2);
        t[2][j] = 0.5f * (g0 - g1 + g2);
        t[3][j] = g2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u[i * 4 + 0] = t[i][0];
        u[i * 4 + 1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        u[i * 4 + 2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        u[i * 4 + 3] = t[i][2];
    }
}

// modes: bit 0 = forward transform into u, bit 1 = data-gradient transform into u2
__device__ __forceinline__ void wino_weight_tile(const float* __restrict__ w, float* __restrict__ u,
                                                 float* __restrict__ u2, int K, int C, int modes, int c0, int k0) {
    __shared__ float gs[9][32][33];  // [tap][k][c]
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int kk = ty; kk < 32; kk += 8) {
        const int k = k0 + kk, c = c0 + tx;
#pragma unroll
        for (int t = 0; t < 9; ++t) gs[t][kk][tx] = (k < K && c < C) ? w[((size_t)k * 9 + t) * C + c] : 0.f;
    }
    __syncthreads();
    const size_t plane = (size_t)K * C;
    for (int rr = ty; rr < 32; rr += 8) {
        float g[9], uu[16];
        if (modes & 1) {  // row = k (rr), column = c (tx): coalesced along c
#pragma unroll
            for (int t = 0; t < 9; ++t) g[t] = gs[t][rr][tx];
            wino_g(g, uu);
            const int k = k0 + rr, c = c0 + tx;
            if (k < K && c < C)
#pragma unroll
                for (int x = 0; x < 16; ++x) u[x * plane + (size_t)k * C + c] = uu[x];
        }
        if (modes & 2) {  // row = c (rr), column = k (tx): coalesced along k; filter rotated by 180 degrees
#pragma unroll
            for (int t = 0; t < 9; ++t) g[t] = gs[8 - t][tx][rr];
            wino_g(g, uu);
            const int c = c0 + rr, k = k0 + tx;
            if (k < K && c < C)
#pragma unroll
                for (int x = 0; x < 16; ++x) u2[x * plane + (size_t)c * K + k] = uu[x];
        }
    }
}

__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int K,
                                                          int C, int mode) {
    wino_weight_tile(w, u, u, K, C, mode == 0 ? 1 : 2, blockIdx.x * 32, blockIdx.y * 32);
}

// Both transforms of MANY weight tensors in one launch (they all go stale together, at the optimizer step): entries[]
// (device, sorted by tile0 = number of 32x32 (k, c) tiles of all preceding tensors); entry.wT -> [2][16][K*C]
// (forward transform, then data-gradient transform).
__global__ __launch_bounds__(256) void wino_weight_multi_kernel(const vspw_wt_entry* __restrict__ entries, int n_entries) {
    const long long b = blockIdx.x;
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (entries[mid].tile0 <= b)
            lo = mid;
        else
            hi = mid - 1;
    }
    const vspw_wt_entry e = entries[lo];
    const int local = (int)(b - e.tile0);
    const int tc = (e.c + 31) / 32;
    wino_weight_tile(e.w, e.wT, e.wT + (size_t)16 * e.k * e.c, e.k, e.c, 3, (local % tc) * 32, (local / tc) * 32);
}

// ------------------------------------------------------------------------------------------------ input
// V[xi][t][c] = (B^T d B)[xi],  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].  One thread: one tile, 4 channels.
// APPLY: x has not been written yet - it is relu(scale * y + shift) of the conv+BN+ReLU node that produces this
// convolution's input (its BatchNorm apply deferred into its one reader, ops._fwd_apply): the patch is evaluated from y
// (same expression tree as bn_apply_kernel: bit-identical values; padding stays zero) and the tile's own 2x2 pixels -
// every pixel belongs to exactly one tile - are written to zout for the backward pass.
template <bool APPLY>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ v, WinoGeom g,
                                                         int C, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ zout) {
    const int c4n = C >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)g.T * c4n) return;
    const int t = (int)(gid / c4n);
    const int c = (int)(gid - (long long)t * c4n) * 4;
    int img, sy, sx, ty, tx;
    wino_tile(g, t, img, sy, sx, ty, tx);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 sc = zero, sh = zero;
    if (APPLY) {
        sc = *reinterpret_cast<const f32x4*>(scale + c);
        sh = *reinterpret_cast<const f32x4*>(shift + c);
    }
    f32x4 dd[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gy = 2 * ty - 1 + i;
        const int py = gy * g.d + sy;
        const bool oky = (gy >= 0) & (py < g.h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = 2 * tx - 1 + j;
            const int px = gx * g.d + sx;
            const bool ok = oky & (gx >= 0) & (px < g.w);
            const size_t e = (((size_t)img * g.h + py) * g.w + px) * C + c;
            f32x4 val = ok ? *reinterpret_cast<const f32x4*>(x + e) : zero;
            if (APPLY) {
                val = val * sc + sh;
#pragma unroll
                for (int q = 0; q < 4; ++q) val[q] = val[q] > 0.f ? val[q] : 0.f;
                if (!ok) val = zero;
                if (ok && (i == 1 || i == 2) && (j == 1 || j == 2)) *reinterpret_cast<f32x4*>(zout + e) = val;
            }
            dd[i][j] = val;
        }
    }
    f32x4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // rows: B^T d
        r[0][j] = dd[0][j] - dd[2][j];
        r[1][j] = dd[1][j] + dd[2][j];
        r[2][j] = dd[2][j] - dd[1][j];
        r[3][j] = dd[1][j] - dd[3][j];
    }
    const size_t plane = (size_t)g.T * C;
    float* out = v + (size_t)t * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // columns: (.) B
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 4 + 0) * plane) = r[i][0] - r[i][2];
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 4 + 1) * plane) = r[i][1] + r[i][2];
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 4 + 2) * plane) = r[i][2] - r[i][1];
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 4 + 3) * plane) = r[i][1] - r[i][3];
    }
}

// ------------------------------------------------------------------------------------------------ output
// Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]; + bias; optional fused epilogues of the direct kernels:
//   FRONT: the BatchNorm-backward front end of the node that produced this convolution's input (data gradient only):
//          o = relu_src > 0 ? o : 0, partial sums of o and o * (bn_y - mean) * invstd  (see IgemmNT in conv_igemm.hip)
//   stats (stat_part != nullptr, forward): partial sums of o and o*o for the training-mode BatchNorm that follows.
// A workgroup owns WINO_TB (= wino_tb()) tiles x CL4 channel quads (grid.y walks the channel quads) and leaves one [2][K] partial
// row per blockIdx.x, summed over its tiles in a fixed order.
// (measured, 256 -> 256 layer3 shape, T = 9 000 tiles: 32 tiles per workgroup give 282 workgroups - barely one per CU,
// 4 waves to hide 16 plane-strided loads each behind; 8 tiles per workgroup give 1 125)
static int wino_tb() {
    static const int v = getenv("VSPW_WINO_TB") ? atoi(getenv("VSPW_WINO_TB")) : 8;
    return (v == 8 || v == 16 || v == 32 || v == 64) ? v : 8;
}

// ROWS: m holds the 8 half-transformed planes P [4][2][m_rows][K] of wino_rows.hip instead of the 16 planes of M
// ([16][m_rows][K], m_rows = T): Y[i][j] = sum_a A^T[i][a] P[a][j].
template <bool FRONT, bool ROWS>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ m, const float* __restrict__ bias,
                                                          float* __restrict__ y, const float* __restrict__ relu_src,
                                                          const float* __restrict__ bn_y,
                                                          const float* __restrict__ bn_mean,
                                                          const float* __restrict__ bn_invstd,
                                                          float* __restrict__ stat_part,
                                                          const float* __restrict__ addend, int act, WinoGeom g, int K,
                                                          int cl4, int WINO_TB, int ldy, int m_rows) {
    __shared__ f32x4 red[2][256];
    const int tid = threadIdx.x;
    const int lane_c = tid % cl4, lane_t = tid / cl4;
    const int tpi_iter = 256 / cl4;  // tiles per iteration
    const int k = (blockIdx.y * cl4 + lane_c) * 4;
    const int t0 = blockIdx.x * WINO_TB;
    const size_t plane = (size_t)m_rows * K;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + k) : zero;
    f32x4 mu = zero, is = zero;
    if (FRONT) {
        mu = *reinterpret_cast<const f32x4*>(bn_mean + k);
        is = *reinterpret_cast<const f32x4*>(bn_invstd + k);
    }
    f32x4 s4 = zero, q4 = zero;
    for (int tt = lane_t; tt < WINO_TB; tt += tpi_iter) {
        const int t = t0 + tt;
        if (t >= g.T) break;
        int img, sy, sx, ty, tx;
        wino_tile(g, t, img, sy, sx, ty, tx);
        const float* in = m + (size_t)t * K + k;
        f32x4 r[2][4];   // 16 planes: rows reduced, r[a][.]
        f32x4 pp[4][2];  // ROWS: P[a][j]
        if constexpr (ROWS) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) pp[i][j] = *reinterpret_cast<const f32x4*>(in + (size_t)(i * 2 + j) * plane);
        } else {
            f32x4 mm[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mm[i][j] = *reinterpret_cast<const f32x4*>(in + (size_t)(i * 4 + j) * plane);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[0][j] = mm[0][j] + mm[1][j] + mm[2][j];
                r[1][j] = mm[1][j] - mm[2][j] - mm[3][j];
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = (2 * ty + a) * g.d + sy;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ox = (2 * tx + b) * g.d + sx;
                if (oy >= g.h || ox >= g.w) continue;
                f32x4 o;
                if constexpr (ROWS)
                    o = (a == 0 ? (pp[0][b] + pp[1][b] + pp[2][b]) : (pp[1][b] - pp[2][b] - pp[3][b])) + bv;
                else
                    o = (b == 0 ? (r[a][0] + r[a][1] + r[a][2]) : (r[a][1] - r[a][2] - r[a][3])) + bv;
                const size_t e = (((size_t)img * g.h + oy) * g.w + ox) * ldy + k;  // (ldy = K except vspw_wino_output_ex)
                if (FRONT) {
                    const f32x4 z = *reinterpret_cast<const f32x4*>(relu_src + e);
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(bn_y + e);
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = z[c] > 0.f ? o[c] : 0.f;
                    q4 += o * ((yv - mu) * is);
                    s4 += o;
                } else {
                    if (addend != nullptr) o += *reinterpret_cast<const f32x4*>(addend + e);  // inference: residual
                    if (act == 1) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], 0.f);
                    }
                    if (stat_part != nullptr) {
                        q4 += o * o;
                        s4 += o;
                    }
                }
                *reinterpret_cast<f32x4*>(y + e) = o;
            }
        }
    }
    if (stat_part == nullptr) return;
    red[0][tid] = s4;
    red[1][tid] = q4;
    __syncthreads();
    if (lane_t == 0) {
        for (int j = 1; j < tpi_iter; ++j) {
            s4 += red[0][j * cl4 + lane_c];
            q4 += red[1][j * cl4 + lane_c];
        }
        float* out = stat_part + (size_t)blockIdx.x * 2 * K;
        *reinterpret_cast<f32x4*>(out + k) = s4;
        *reinterpret_cast<f32x4*>(out + K + k) = q4;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// Y = A^T M A, M = U o V  =>  dM = A dY A^T,  dU[xi] = dM[xi]^T V[xi] (a TN GEMM over the tiles, vspw_bmm_tn, batch 16),
// dg = G^T dU G.  A = [[1,0],[1,1],[1,-1],[0,-1]].  Output pixels outside the image (ragged tiles) carry no gradient.
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, float* __restrict__ dm, WinoGeom g,
                                                      int K) {
    const int k4n = K >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)g.T * k4n) return;
    const int t = (int)(gid / k4n);
    const int k = (int)(gid - (long long)t * k4n) * 4;
    int img, sy, sx, ty, tx;
    wino_tile(g, t, img, sy, sx, ty, tx);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = (2 * ty + a) * g.d + sy;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ox = (2 * tx + b) * g.d + sx;
            d[a][b] = (oy < g.h && ox < g.w)
                          ? *reinterpret_cast<const f32x4*>(dy + (((size_t)img * g.h + oy) * g.w + ox) * K + k)
                          : zero;
        }
    }
    f32x4 r[4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        r[0][b] = d[0][b];
        r[1][b] = d[0][b] + d[1][b];
        r[2][b] = d[0][b] - d[1][b];
        r[3][b] = -d[1][b];
    }
    const size_t plane = (size_t)g.T * K;
    float* out = dm + (size_t)t * K + k;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 4 + 0) * plane) = r[i][0];
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 4 + 1) * plane) = r[i][0] + r[i][1];
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 4 + 2) * plane) = r[i][0] - r[i][1];
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 4 + 3) * plane) = -r[i][1];
    }
}

// dW[k][ky][kx][c] = (G^T dU G)[ky][kx]; dU [16][K][C]; one thread per (k, 4 channels)
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ du, float* __restrict__ dw, int K, int C) {
    const int c4n = C >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)K * c4n) return;
    const int k = (int)(gid / c4n);
    const int c = (int)(gid - (long long)k * c4n) * 4;
    const size_t plane = (size_t)K * C;
    const float* in = du + (size_t)k * C + c;
    f32x4 t[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(in + (size_t)(0 * 4 + b) * plane);
        const f32x4 u1 = *reinterpret_cast<const f32x4*>(in + (size_t)(1 * 4 + b) * plane);
        const f32x4 u2 = *reinterpret_cast<const f32x4*>(in + (size_t)(2 * 4 + b) * plane);
        const f32x4 u3 = *reinterpret_cast<const f32x4*>(in + (size_t)(3 * 4 + b) * plane);
        t[0][b] = u0 + 0.5f * (u1 + u2);
        t[1][b] = 0.5f * (u1 - u2);
        t[2][b] = 0.5f * (u1 + u2) + u3;
    }
    float* out = dw + (size_t)k * 9 * C + c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 3 + 0) * C) = t[i][0] + 0.5f * (t[i][1] + t[i][2]);
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 3 + 1) * C) = 0.5f * (t[i][1] - t[i][2]);
        *reinterpret_cast<f32x4*>(out + (size_t)(i * 3 + 2) * C) = 0.5f * (t[i][1] + t[i][2]) + t[i][3];
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
static int wino_cl4(int K) {
    const int k4 = K / 4;
    if (K % 4) return 0;
    for (int cl = 256; cl >= 8; cl >>= 1)
        if (k4 % cl == 0) return cl;
    return 0;
}

extern "C" size_t vspw_wino_supported(const vspw_conv_desc* d) {
    WinoGeom g;
    if (!wino_geom(d, g)) return 0;
    if (d->c % 32 != 0 || d->k % 32 != 0) return 0;  // vector gathers + the v2 GEMM kernel on both sides
    return (wino_cl4(d->k) && wino_cl4(d->c)) ? 1 : 0;
}

extern "C" long long vspw_wino_tiles(const vspw_conv_desc* d) {
    WinoGeom g;
    return wino_geom(d, g) ? g.T : 0;
}

extern "C" size_t vspw_wino_stat_partials(const vspw_conv_desc* d) {
    WinoGeom g;
    return wino_geom(d, g) ? (size_t)vspw_cdiv(g.T, wino_tb()) : 0;
}

extern "C" int vspw_wino_weights(const float* w, float* u, int k, int c, int data_gradient, void* stream) {
    if (!w || !u || k <= 0 || c <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(wino_weight_kernel, dim3(vspw_cdiv(c, 32), vspw_cdiv(k, 32)), dim3(256), 0, vspw_stream(stream), w,
                       u, k, c, data_gradient ? 1 : 0);
    return vspw_launch_status();
}

extern "C" int vspw_wino_input(const vspw_conv_desc* d, const float* x, int channels, float* v, void* stream) {
    WinoGeom g;
    if (!wino_geom(d, g) || !x || !v || channels <= 0 || channels % 4) return VSPW_EINVAL;
    const long long items = (long long)g.T * (channels / 4);
    hipLaunchKernelGGL(wino_input_kernel<false>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, vspw_stream(stream),
                       x, v, g, channels, nullptr, nullptr, nullptr);
    return vspw_launch_status();
}

// vspw_wino_input of z = relu(scale * y + shift), which has NOT been materialised (the producing conv+BN+ReLU node
// deferred its apply, cf. vspw_conv2d_fwd_apply): V from y, and z written to z_out for the node's other readers (the
// backward pass).  scale_shift [2][channels].
extern "C" int vspw_wino_input_apply(const vspw_conv_desc* d, const float* y, const float* scale_shift, float* z_out,
                                     int channels, float* v, void* stream) {
    WinoGeom g;
    if (!wino_geom(d, g) || !y || !scale_shift || !z_out || !v || channels <= 0 || channels % 4) return VSPW_EINVAL;
    const long long items = (long long)g.T * (channels / 4);
    hipLaunchKernelGGL(wino_input_kernel<true>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, vspw_stream(stream),
                       y, v, g, channels, scale_shift, scale_shift + channels, z_out);
    return vspw_launch_status();
}

extern "C" int vspw_wino_output(const vspw_conv_desc* d, const float* m, int channels, const float* bias, float* y,
                                const float* relu_src, const float* bn_y, const float* bn_mean, const float* bn_invstd,
                                float* stat_part, const float* addend, int act, void* stream) {
    WinoGeom g;
    const int cl4 = wino_cl4(channels);
    if (act != 0 && act != 1) return VSPW_EINVAL;
    if (relu_src != nullptr && (addend != nullptr || act != 0)) return VSPW_EINVAL;
    if (!wino_geom(d, g) || !m || !y || cl4 == 0) return VSPW_EINVAL;
    const bool front = relu_src != nullptr;
    if (front && (!bn_y || !bn_mean || !bn_invstd || !stat_part)) return VSPW_EINVAL;
    const int tb = wino_tb();
    const dim3 grid(vspw_cdiv(g.T, tb), channels / 4 / cl4);
    if (front)
        hipLaunchKernelGGL((wino_output_kernel<true, false>), grid, dim3(256), 0, vspw_stream(stream), m, bias, y, relu_src,
                           bn_y, bn_mean, bn_invstd, stat_part, nullptr, 0, g, channels, cl4, tb, channels, g.T);
    else
        hipLaunchKernelGGL((wino_output_kernel<false, false>), grid, dim3(256), 0, vspw_stream(stream), m, bias, y, nullptr,
                           nullptr, nullptr, nullptr, stat_part, addend, act, g, channels, cl4, tb, channels, g.T);
    return vspw_launch_status();
}

// vspw_wino_output for the half-transformed planes of vspw_wino_gemm_rows / vspw_wino_gemm_fused_rows:
// tp [4][2][tpad][channels], tpad = vspw_wino_rows_tpad(d, reduce channels, channels).
extern "C" int vspw_wino_output_rows(const vspw_conv_desc* d, const float* tp, long long tpad, int channels,
                                     const float* bias, float* y, const float* relu_src, const float* bn_y,
                                     const float* bn_mean, const float* bn_invstd, float* stat_part, const float* addend,
                                     int act, void* stream) {
    WinoGeom g;
    const int cl4 = wino_cl4(channels);
    if (act != 0 && act != 1) return VSPW_EINVAL;
    if (relu_src != nullptr && (addend != nullptr || act != 0)) return VSPW_EINVAL;
    if (!wino_geom(d, g) || !tp || !y || cl4 == 0 || tpad < g.T || tpad > 0x3fffffffLL) return VSPW_EINVAL;
    const bool front = relu_src != nullptr;
    if (front && (!bn_y || !bn_mean || !bn_invstd || !stat_part)) return VSPW_EINVAL;
    const int tb = wino_tb();
    const dim3 grid(vspw_cdiv(g.T, tb), channels / 4 / cl4);
    if (front)
        hipLaunchKernelGGL((wino_output_kernel<true, true>), grid, dim3(256), 0, vspw_stream(stream), tp, bias, y, relu_src,
                           bn_y, bn_mean, bn_invstd, stat_part, nullptr, 0, g, channels, cl4, tb, channels, (int)tpad);
    else
        hipLaunchKernelGGL((wino_output_kernel<false, true>), grid, dim3(256), 0, vspw_stream(stream), tp, bias, y, nullptr,
                           nullptr, nullptr, nullptr, stat_part, addend, act, g, channels, cl4, tb, channels, (int)tpad);
    return vspw_launch_status();
}

// y = act(A^T M A + bias) written with pixel stride ldy >= channels (a channel slot of a wider NHWC buffer): the frozen
// flow network's 3x3 convolutions (RAFT_core/update.py:16-17,82-87), whose outputs land inside concatenation buffers.
extern "C" int vspw_wino_output_ex(const vspw_conv_desc* d, const float* m, int channels, const float* bias, float* y,
                                   long long ldy, int act, void* stream) {
    WinoGeom g;
    const int cl4 = wino_cl4(channels);
    if ((act != 0 && act != 1) || !wino_geom(d, g) || !m || !y || cl4 == 0 || ldy < channels || (ldy & 3) ||
        ldy > 0x7fffffff)
        return VSPW_EINVAL;
    const int tb = wino_tb();
    const dim3 grid(vspw_cdiv(g.T, tb), channels / 4 / cl4);
    hipLaunchKernelGGL((wino_output_kernel<false, false>), grid, dim3(256), 0, vspw_stream(stream), m, bias, y, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, act, g, channels, cl4, tb, (int)ldy, g.T);
    return vspw_launch_status();
}

// vspw_wino_output_ex for the half-transformed planes (see vspw_wino_output_rows).
extern "C" int vspw_wino_output_rows_ex(const vspw_conv_desc* d, const float* tp, long long tpad, int channels,
                                        const float* bias, float* y, long long ldy, int act, void* stream) {
    WinoGeom g;
    const int cl4 = wino_cl4(channels);
    if ((act != 0 && act != 1) || !wino_geom(d, g) || !tp || !y || cl4 == 0 || ldy < channels || (ldy & 3) ||
        ldy > 0x7fffffff || tpad < g.T || tpad > 0x3fffffffLL)
        return VSPW_EINVAL;
    const int tb = wino_tb();
    const dim3 grid(vspw_cdiv(g.T, tb), channels / 4 / cl4);
    hipLaunchKernelGGL((wino_output_kernel<false, true>), grid, dim3(256), 0, vspw_stream(stream), tp, bias, y, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, act, g, channels, cl4, tb, (int)ldy, (int)tpad);
    return vspw_launch_status();
}

extern "C" int vspw_wino_dy(const vspw_conv_desc* d, const float* dy, int channels, float* dm, void* stream) {
    WinoGeom g;
    if (!wino_geom(d, g) || !dy || !dm || channels <= 0 || channels % 4) return VSPW_EINVAL;
    const long long items = (long long)g.T * (channels / 4);
    hipLaunchKernelGGL(wino_dy_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, vspw_stream(stream), dy, dm, g,
                       channels);
    return vspw_launch_status();
}

extern "C" int vspw_wino_dw(const float* du, float* dw, int k, int c, void* stream) {
    if (!du || !dw || k <= 0 || c <= 0 || c % 4) return VSPW_EINVAL;
    const long long items = (long long)k * (c / 4);
    hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, vspw_stream(stream), du, dw, k,
                       c);
    return vspw_launch_status();
}

extern "C" long long vspw_wino_weight_tiles(int k, int c) { return (long long)vspw_cdiv(k, 32) * vspw_cdiv(c, 32); }

extern "C" int vspw_wino_weights_multi(const vspw_wt_entry* entries, int n_entries, long long total_tiles, void* stream) {
    if (!entries || n_entries <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffLL) return VSPW_EINVAL;
    hipLaunchKernelGGL(wino_weight_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, vspw_stream(stream), entries,
                       n_entries);
    return vspw_launch_status();
}
