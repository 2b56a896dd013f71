// Winograd F(3x3, 3x3) and F(4x4, 3x3) for the stride-1 3x3 convolutions of the path (forward, data gradient, weight
// gradient).  One template over the output tile edge M; the text below describes M = 3, the F(4x4) notes follow it.
//
// Round 6.  F(2x2, 3x3) (winograd.hip) executes 16 multiplications per 4 outputs (4.0 per output pixel, 4/9 of the direct
// count) and pads the 15 x 15 dilation sub-grids of the dilation-4 stage to 16 x 16 (13.8 % surplus).  F(3x3, 3x3) computes
// a 3x3 output tile from a 5x5 input patch with 25 multiplications per (cin, cout) pair: 2.78 per output pixel (25/81 of
// the direct count), and 3 divides 60, 30 and 15 - the sub-grid sizes of every dilation of the 60 x 60 stage - exactly.
//     Y = A^T [ (G g G^T) o (B^T d B) ] A      over the interpolation points {0, 1, -1, 2, inf}
// i.e. per transform position xi = (a, b) one plain GEMM  M[xi] = V[xi] (tiles x Cin) . U[xi]^T (Cout x Cin): 25 batched
// GEMMs run by the pointwise MFMA kernel (vspw_bmm_nt).  The transforms of the activations stay in fp32 (B^T and A^T have
// integer entries of magnitude <= 4); the weight-side transforms (G has 1/2, 1/6, 1/3, 2/3) are evaluated in fp64 and
// rounded once - they touch 9 / 25 values per (cout, cin) pair and cost nothing.
// Numerics (tools/diag/wino_f33_probe.py, oracle arithmetic, where the product uses it = min(Cin, Cout) >= 128): one
// layer-3 convolution is 2.4x (forward) farther from float64 than the direct fp32 convolution, but at the network level the
// early layers' error dominates: the pinned-decision gradient error is 8.5e-4 against 8.2e-4 (direct) / 7.9e-4 (F(2x2)),
// raw-weight forward activations 1.03x the direct convolution's own distance from float64.
//
// Dilation d: d*d independent undilated convolutions on the sub-grids {(y, x): y = sy (mod d), x = sx (mod d)}, tiles laid
// out per (image, sub-grid) as in winograd.hip.  Ragged edges are zero-padded tiles whose surplus outputs are dropped.
// Layouts: activations NHWC; V [25][T][Cin], U [25][Cout][Cin], M [25][T][Cout], T = n * d*d * th * tw tiles.
// F(4x4, 3x3) (M = 4): a 4x4 output tile from a 6x6 patch over the points {0, 1, -1, 1/2, -2, inf} - 36 multiplications
// per 16 outputs = 2.25 per output pixel.  4 divides 60 exactly (the undilated shapes: heads, layer3.0, layer2); the 30 / 15
// pixel sub-grids of dilation 2 / 4 pad to 32 / 16 (13.8 % surplus: 2.56 per output, still 8 % below F(3x3)).  The point set
// was chosen by the conv-level probe (tools/diag/wino_f33_probe.py): {0, +-1, 1/2, -2} gives 1.65e-6 forward error on a
// layer-3 convolution - F(3x3): 1.45e-6 - where the textbook {0, +-1, +-2} gives 2.5e-6; B^T and A^T entries stay exact
// binary fractions (1/2, 3/2, 5/2, 1/4, 1/8, 8), G has 1/3, 1/15, 16/15 (fp64, rounded once).  ops._wino_fm picks, per
// geometry, the tile size that executes fewer multiplications.
// Reference call sites: the 3x3 convolutions of models/resnet.py:72-92 after models/models.py:737-750, the heads'
// models/clip_psp.py:29-35,74-79, models/clip_ocr.py:41-52 (F.conv2d, and its two gradients under loss.backward(),
// train_clip2.py:99).
#include "common.h"

// B^T [N][N], A^T [M][N] (fp32: exact binary fractions), G [N][3] (fp64); N = M + 2.  Generated from the Cook-Toom
// construction of oracle/np_wino.py (points {0,1,-1,2,inf}, {0,1,-1,1/2,-2,inf} and {0,1,-1,1/2,-1/2,2,inf}).
__device__ constexpr float kBT3[5][5] = {{2.f, -1.f, -2.f, 1.f, 0.f}, {0.f, -2.f, -1.f, 1.f, 0.f}, {0.f, 2.f, -3.f, 1.f, 0.f}, {0.f, -1.f, 0.f, 1.f, 0.f}, {0.f, 2.f, -1.f, -2.f, 1.f}};
__device__ constexpr float kAT3[3][5] = {{1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, 0.f}, {0.f, 1.f, 1.f, 4.f, 1.f}};
__device__ constexpr double kG3[5][3] = {{1. / 2., 0., 0.}, {-1. / 2., -1. / 2., -1. / 2.}, {-1. / 6., 1. / 6., -1. / 6.}, {1. / 6., 1. / 3., 2. / 3.}, {0., 0., 1.}};
__device__ constexpr float kBT4[6][6] = {{1.f, -3.f / 2.f, -2.f, 3.f / 2.f, 1.f, 0.f}, {0.f, -1.f, 1.f / 2.f, 5.f / 2.f, 1.f, 0.f}, {0.f, 1.f, -5.f / 2.f, 1.f / 2.f, 1.f, 0.f}, {0.f, -2.f, -1.f, 2.f, 1.f, 0.f}, {0.f, 1.f / 2.f, -1.f, -1.f / 2.f, 1.f, 0.f}, {0.f, 1.f, -3.f / 2.f, -2.f, 3.f / 2.f, 1.f}};
__device__ constexpr float kAT4[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 1.f / 2.f, -2.f, 0.f}, {0.f, 1.f, 1.f, 1.f / 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 1.f / 8.f, -8.f, 1.f}};
__device__ constexpr double kG4[6][3] = {{1., 0., 0.}, {1. / 3., 1. / 3., 1. / 3.}, {-1. / 3., 1. / 3., -1. / 3.}, {-16. / 15., -8. / 15., -4. / 15.}, {1. / 15., -2. / 15., 4. / 15.}, {0., 0., 1.}};
__device__ constexpr float kBT5[7][7] = {{-1.f / 2.f, 1.f / 4.f, 5.f / 2.f, -5.f / 4.f, -2.f, 1.f, 0.f}, {0.f, 1.f / 2.f, 1.f / 4.f, -9.f / 4.f, -1.f, 1.f, 0.f}, {0.f, -1.f / 2.f, 3.f / 4.f, 7.f / 4.f, -3.f, 1.f, 0.f}, {0.f, 1.f, 3.f / 2.f, -2.f, -3.f / 2.f, 1.f, 0.f}, {0.f, -1.f, 5.f / 2.f, 0.f, -5.f / 2.f, 1.f, 0.f}, {0.f, 1.f / 4.f, 0.f, -5.f / 4.f, 0.f, 1.f, 0.f}, {0.f, -1.f / 2.f, 1.f / 4.f, 5.f / 2.f, -5.f / 4.f, -2.f, 1.f}};
__device__ constexpr float kAT5[5][7] = {{1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 1.f / 2.f, -1.f / 2.f, 2.f, 0.f}, {0.f, 1.f, 1.f, 1.f / 4.f, 1.f / 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 1.f / 8.f, -1.f / 8.f, 8.f, 0.f}, {0.f, 1.f, 1.f, 1.f / 16.f, 1.f / 16.f, 16.f, 1.f}};
__device__ constexpr double kG5[7][3] = {{-2., 0., 0.}, {-2. / 3., -2. / 3., -2. / 3.}, {-2. / 9., 2. / 9., -2. / 9.}, {16. / 9., 8. / 9., 4. / 9.}, {16. / 15., -8. / 15., 4. / 15.}, {2. / 45., 4. / 45., 8. / 45.}, {0., 0., 1.}};
template <int M> __device__ __forceinline__ constexpr float cBT(int a, int i) { if constexpr (M == 3) return kBT3[a][i]; else if constexpr (M == 4) return kBT4[a][i]; else return kBT5[a][i]; }
template <int M> __device__ __forceinline__ constexpr float cAT(int a, int i) { if constexpr (M == 3) return kAT3[a][i]; else if constexpr (M == 4) return kAT4[a][i]; else return kAT5[a][i]; }
template <int M> __device__ __forceinline__ constexpr double cG(int a, int i) { if constexpr (M == 3) return kG3[a][i]; else if constexpr (M == 4) return kG4[a][i]; else return kG5[a][i]; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int V> struct VecT { typedef f32x4 type; };
template <> struct VecT<2> { typedef f32x2 type; };
// channels per thread of the input / dY transforms: F(5x5)'s 49 accumulators x 4 would not fit the register file
template <int WM> struct WinoVec { static constexpr int value = WM >= 5 ? 2 : 4; };

struct Geom {
    int n, h, w, d;      // images, height, width, dilation
    int th, tw, tpi, T;  // tiles per sub-grid column / row, tiles per image, total
};

template <int WM>
static bool geom(const vspw_conv_desc* dsc, Geom& g) {
    if (!dsc || dsc->kh != 3 || dsc->kw != 3 || dsc->stride != 1 || dsc->dil < 1 || dsc->pad != dsc->dil ||
        dsc->pad_w != dsc->dil || dsc->oh != dsc->h || dsc->ow != dsc->w || dsc->n < 1)
        return false;
    g.n = dsc->n; g.h = dsc->h; g.w = dsc->w; g.d = dsc->dil;
    const int hs = (g.h + g.d - 1) / g.d, ws = (g.w + g.d - 1) / g.d;
    g.th = (hs + WM - 1) / WM;
    g.tw = (ws + WM - 1) / WM;
    const long long tpi = (long long)g.d * g.d * g.th * g.tw;
    if (tpi * g.n > 0x3fffffffLL) return false;
    g.tpi = (int)tpi;
    g.T = (int)(tpi * g.n);
    return true;
}

__device__ __forceinline__ void tile_of(const Geom& g, int t, int& img, int& sy, int& sx, int& ty, int& tx) {
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
// U = G g G^T.  w: [K][3][3][C] (channels_last OIHW).
// mode bit 0 (forward):        U[xi][k][c] from g = w[k, :, :, c]
// mode bit 1 (data gradient):  U[xi][c][k] from g = w[k, 2-ky, 2-kx, c]   (rows = Cin, reduction over Cout)
// (explicit fma: the one-tensor and the multi-tensor kernel must round identically - left to the compiler's contraction
// choices they differed by one fp32 ulp in a few elements per tensor, and which of the two refreshed a transform depends
// on the cache's history)
template <int WM>
__device__ __forceinline__ void weight_g(const float g[9], float* u) {
    constexpr int WN = WM + 2;
    double t[WN][3];
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (cG<WM>(a, i) != 0.) s = __builtin_fma(cG<WM>(a, i), (double)g[i * 3 + j], s);
            t[a][j] = s;
        }
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            double s = 0.;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (cG<WM>(b, j) != 0.) s = __builtin_fma(t[a][j], cG<WM>(b, j), s);
            u[a * WN + b] = (float)s;
        }
}

template <int WM>
__device__ __forceinline__ void weight_tile(const float* __restrict__ w, float* __restrict__ u, float* __restrict__ u2,
                                            int K, int C, int modes, int c0, int k0) {
    constexpr int WN = WM + 2, WP = WN * WN;
    (void)WN;
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
        float g[9], uu[WP];
        if (modes & 1) {  // row = k (rr), column = c (tx): coalesced along c
#pragma unroll
            for (int t = 0; t < 9; ++t) g[t] = gs[t][rr][tx];
            weight_g<WM>(g, uu);
            const int k = k0 + rr, c = c0 + tx;
            if (k < K && c < C)
#pragma unroll
                for (int x = 0; x < WP; ++x) u[x * plane + (size_t)k * C + c] = uu[x];
        }
        if (modes & 2) {  // row = c (rr), column = k (tx): coalesced along k; filter rotated by 180 degrees
#pragma unroll
            for (int t = 0; t < 9; ++t) g[t] = gs[8 - t][tx][rr];
            weight_g<WM>(g, uu);
            const int c = c0 + rr, k = k0 + tx;
            if (k < K && c < C)
#pragma unroll
                for (int x = 0; x < WP; ++x) u2[x * plane + (size_t)c * K + k] = uu[x];
        }
    }
}

template <int WM>
__global__ __launch_bounds__(256) void wino3_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int K, int C,
                                                     int mode) {
    weight_tile<WM>(w, u, u, K, C, mode == 0 ? 1 : 2, blockIdx.x * 32, blockIdx.y * 32);
}

// Both transforms of MANY weight tensors in one launch (cf. wino_weight_multi_kernel): entry.wT -> [2][25][K*C].
template <int WM>
__global__ __launch_bounds__(256) void wino3_weight_multi_kernel(const vspw_wt_entry* __restrict__ entries, int n_entries) {
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
    weight_tile<WM>(e.w, e.wT, e.wT + (size_t)(WM + 2) * (WM + 2) * e.k * e.c, e.k, e.c, 3, (local % tc) * 32, (local / tc) * 32);
}

// ------------------------------------------------------------------------------------------------ input
// V[xi][t][c] = (B^T d B)[xi].  One thread: one tile, 4 channels; patch rows are consumed as they arrive
// (row i contributes BT[a][i] * (d[i][.] B)[b] to every V[a][b]).
// APPLY: x has not been written yet - it is relu(scale * y + shift) of the conv+BN+ReLU node that produces this convolution's
// input (its BatchNorm apply deferred into its one reader, ops._fwd_apply; cf. wino_input_kernel<true> in winograd.hip): the
// patch is evaluated from y (same expression tree as bn_apply_kernel: bit-identical values; padding stays zero) and the tile's
// own M x M pixels - every pixel belongs to exactly one tile - are written to zout for the backward pass.
template <int WM, bool APPLY>
__global__ __launch_bounds__(256) void wino3_input_kernel(const float* __restrict__ x, float* __restrict__ v, Geom g, int C,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ zout) {
    constexpr int WN = WM + 2;
    constexpr int VEC = WinoVec<WM>::value;
    typedef typename VecT<VEC>::type vec;
    const int c4n = C / VEC;
    // XCD-aware order: neighbouring tiles share two of their five patch columns / rows - keep them in one L2 (measured
    // before: FETCH_SIZE 83 MB per launch for a 37 MB input, workgroups of neighbouring tiles landing on eight XCDs)
    const long long gid = (long long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (gid >= (long long)g.T * c4n) return;
    const int t = (int)(gid / c4n);
    const int c = (int)(gid - (long long)t * c4n) * VEC;
    int img, sy, sx, ty, tx;
    tile_of(g, t, img, sy, sx, ty, tx);
    const vec zero = (vec)(0.f);
    vec sc = zero, sh = zero;
    if (APPLY) {
        sc = *reinterpret_cast<const vec*>(scale + c);
        sh = *reinterpret_cast<const vec*>(shift + c);
    }
    vec acc[WN][WN];
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = zero;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int gy = WM * ty - 1 + i;
        const int py = gy * g.d + sy;
        const bool oky = (gy >= 0) & (py < g.h);
        vec dd[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int gx = WM * tx - 1 + j;
            const int px = gx * g.d + sx;
            const bool ok = oky & (gx >= 0) & (px < g.w);
            const size_t e = (((size_t)img * g.h + py) * g.w + px) * C + c;
            vec val = ok ? *reinterpret_cast<const vec*>(x + e) : zero;
            if (APPLY) {
                val = val * sc + sh;
#pragma unroll
                for (int q = 0; q < VEC; ++q) val[q] = val[q] > 0.f ? val[q] : 0.f;
                if (!ok) val = zero;
                if (ok && i >= 1 && i <= WM && j >= 1 && j <= WM) *reinterpret_cast<vec*>(zout + e) = val;
            }
            dd[j] = val;
        }
        vec r[WN];
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            vec s = zero;
#pragma unroll
            for (int j = 0; j < WN; ++j)
                if (cBT<WM>(b, j) != 0.f) s += cBT<WM>(b, j) * dd[j];
            r[b] = s;
        }
#pragma unroll
        for (int a = 0; a < WN; ++a)
            if (cBT<WM>(a, i) != 0.f) {
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] += cBT<WM>(a, i) * r[b];
            }
    }
    const size_t plane = (size_t)g.T * C;
    float* out = v + (size_t)t * C + c;
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) *reinterpret_cast<vec*>(out + (size_t)(a * WN + b) * plane) = acc[a][b];
}

// ------------------------------------------------------------------------------------------------ output
// Y = A^T M A + bias, with the fused epilogues of wino_output_kernel (winograd.hip):
//   FRONT: BatchNorm-backward front end of the node that produced this convolution's input (data gradient only):
//          o = relu_src > 0 ? o : 0, partial sums of o and o * (bn_y - mean) * invstd
//   stats (stat_part != nullptr, forward): partial sums of o and o*o for the training-mode BatchNorm that follows.
// A workgroup owns TB tiles x cl4 channel quads and leaves one [2][K] partial row per blockIdx.x.
constexpr int TB = 8;

template <int WM, bool FRONT>
__global__ __launch_bounds__(256, WM == 3 ? 3 : (WM == 4 ? 2 : 1)) void wino3_output_kernel(const float* __restrict__ m, const float* __restrict__ bias,
                                                     float* __restrict__ y, const float* __restrict__ relu_src,
                                                     const float* __restrict__ bn_y, const float* __restrict__ bn_mean,
                                                     const float* __restrict__ bn_invstd, float* __restrict__ stat_part,
                                                     const float* __restrict__ addend, int act, Geom g, int K, int cl4) {
    constexpr int WN = WM + 2;
    __shared__ f32x4 red[2][256];
    const int tid = threadIdx.x;
    const int lane_c = tid % cl4, lane_t = tid / cl4;
    const int tpi_iter = 256 / cl4;  // tiles per iteration
    const int k = (blockIdx.y * cl4 + lane_c) * 4;
    const int t0 = blockIdx.x * TB;
    const size_t plane = (size_t)g.T * K;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + k) : zero;
    f32x4 mu = zero, is = zero;
    if (FRONT) {
        mu = *reinterpret_cast<const f32x4*>(bn_mean + k);
        is = *reinterpret_cast<const f32x4*>(bn_invstd + k);
    }
    // (registers: F(3x3) 161-170 = three waves per SIMD, F(4x4) 229-239 = two; with per-plane 64-bit addresses they were 216 /
    // 300 = two / one)
    f32x4 s4 = zero, q4 = zero;
    for (int tt = lane_t; tt < TB; tt += tpi_iter) {
        const int t = t0 + tt;
        if (t >= g.T) break;
        int img, sy, sx, ty, tx;
        tile_of(g, t, img, sy, sx, ty, tx);
        // plane bases are workgroup-uniform (scalar registers), the per-thread part of every plane address is ONE 32-bit
        // element offset (36 / 25 64-bit addresses would cost 72 / 50 registers)
        const unsigned in = (unsigned)t * (unsigned)K + (unsigned)k;
        f32x4 yy[WM][WM];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WM; ++j) yy[i][j] = zero;
#pragma unroll
        for (int a = 0; a < WN; ++a) {
            f32x4 mm[WN];
#pragma unroll
            for (int b = 0; b < WN; ++b) mm[b] = *reinterpret_cast<const f32x4*>(m + (size_t)(a * WN + b) * plane + in);
            f32x4 p[WM];
#pragma unroll
            for (int j = 0; j < WM; ++j) {
                f32x4 s = zero;
#pragma unroll
                for (int b = 0; b < WN; ++b)
                    if (cAT<WM>(j, b) != 0.f) s += cAT<WM>(j, b) * mm[b];
                p[j] = s;
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
                if (cAT<WM>(i, a) != 0.f) {
#pragma unroll
                    for (int j = 0; j < WM; ++j) yy[i][j] += cAT<WM>(i, a) * p[j];
                }
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int oy = (WM * ty + i) * g.d + sy;
#pragma unroll
            for (int j = 0; j < WM; ++j) {
                const int ox = (WM * tx + j) * g.d + sx;
                if (oy >= g.h || ox >= g.w) continue;
                f32x4 o = yy[i][j] + bv;
                const size_t e = (((size_t)img * g.h + oy) * g.w + ox) * K + k;
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
// dM = A dY A^T (A = (A^T)^T, 5x3), dU[xi] = dM[xi]^T V[xi] (vspw_bmm_tn, batch 25), dg = G^T dU G.
// Output pixels outside the image (ragged tiles) carry no gradient.
template <int WM>
__global__ __launch_bounds__(256) void wino3_dy_kernel(const float* __restrict__ dy, float* __restrict__ dm, Geom g, int K) {
    constexpr int WN = WM + 2;
    constexpr int VEC = WinoVec<WM>::value;
    typedef typename VecT<VEC>::type vec;
    const int k4n = K / VEC;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)g.T * k4n) return;
    const int t = (int)(gid / k4n);
    const int k = (int)(gid - (long long)t * k4n) * VEC;
    int img, sy, sx, ty, tx;
    tile_of(g, t, img, sy, sx, ty, tx);
    const vec zero = (vec)(0.f);
    vec acc[WN][WN];
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = zero;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int oy = (WM * ty + i) * g.d + sy;
        vec d[WM];
#pragma unroll
        for (int j = 0; j < WM; ++j) {
            const int ox = (WM * tx + j) * g.d + sx;
            d[j] = (oy < g.h && ox < g.w) ? *reinterpret_cast<const vec*>(dy + (((size_t)img * g.h + oy) * g.w + ox) * K + k)
                                          : zero;
        }
        vec r[WN];
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            vec s = zero;
#pragma unroll
            for (int j = 0; j < WM; ++j)
                if (cAT<WM>(j, b) != 0.f) s += cAT<WM>(j, b) * d[j];
            r[b] = s;
        }
#pragma unroll
        for (int a = 0; a < WN; ++a)
            if (cAT<WM>(i, a) != 0.f) {
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] += cAT<WM>(i, a) * r[b];
            }
    }
    const size_t plane = (size_t)g.T * K;
    float* out = dm + (size_t)t * K + k;
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) *reinterpret_cast<vec*>(out + (size_t)(a * WN + b) * plane) = acc[a][b];
}

// dW[k][ky][kx][c] = (G^T dU G)[ky][kx]; dU [25][K][C]; one thread per (k, c), fp64 arithmetic
template <int WM>
__global__ __launch_bounds__(256) void wino3_dw_kernel(const float* __restrict__ du, float* __restrict__ dw, int K, int C) {
    constexpr int WN = WM + 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)K * C) return;
    const int k = (int)(gid / C);
    const int c = (int)(gid - (long long)k * C);
    const size_t plane = (size_t)K * C;
    const float* in = du + (size_t)k * C + c;
    double t[3][WN];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int b = 0; b < WN; ++b) t[i][b] = 0.;
#pragma unroll
    for (int a = 0; a < WN; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            const double u = (double)in[(size_t)(a * WN + b) * plane];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (cG<WM>(a, i) != 0.) t[i][b] = __builtin_fma(cG<WM>(a, i), u, t[i][b]);
        }
    float* out = dw + (size_t)k * 9 * C + c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.;
#pragma unroll
            for (int b = 0; b < WN; ++b)
                if (cG<WM>(b, j) != 0.) s = __builtin_fma(t[i][b], cG<WM>(b, j), s);
            out[(size_t)(i * 3 + j) * C] = (float)s;
        }
}

static int cl4_of(int K) {
    const int k4 = K / 4;
    if (K % 4) return 0;
    for (int cl = 256; cl >= 8; cl >>= 1)
        if (k4 % cl == 0) return cl;
    return 0;
}

// ------------------------------------------------------------------------------------------------ C ABI
// vspw_wino3_* = F(3x3,3x3), vspw_wino4_* = F(4x4,3x3), vspw_wino5_* = F(5x5,3x3): the same nine calls (25 / 36 / 49 planes).
template <int WM>
static size_t t_supported(const vspw_conv_desc* d) {
    Geom g;
    if (!geom<WM>(d, g)) return 0;
    if (d->c % 32 != 0 || d->k % 32 != 0) return 0;  // vector gathers + the v2 GEMM kernel on both sides
    const long long cmax = d->c > d->k ? d->c : d->k;
    if ((long long)g.T * cmax >= (1LL << 31)) return 0;  // 32-bit element offsets inside a plane (transform kernels)
    return (cl4_of(d->k) && cl4_of(d->c)) ? 1 : 0;
}
template <int WM>
static long long t_tiles(const vspw_conv_desc* d) {
    Geom g;
    return geom<WM>(d, g) ? g.T : 0;
}
template <int WM>
static int t_weights(const float* w, float* u, int k, int c, int data_gradient, void* stream) {
    if (!w || !u || k <= 0 || c <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(wino3_weight_kernel<WM>, dim3(vspw_cdiv(c, 32), vspw_cdiv(k, 32)), dim3(256), 0, vspw_stream(stream),
                       w, u, k, c, data_gradient ? 1 : 0);
    return vspw_launch_status();
}
template <int WM>
static int t_weights_multi(const vspw_wt_entry* entries, int n_entries, long long total_tiles, void* stream) {
    if (!entries || n_entries <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffLL) return VSPW_EINVAL;
    hipLaunchKernelGGL(wino3_weight_multi_kernel<WM>, dim3((unsigned)total_tiles), dim3(256), 0, vspw_stream(stream), entries,
                       n_entries);
    return vspw_launch_status();
}
template <int WM>
static int t_input(const vspw_conv_desc* d, const float* x, int channels, float* v, void* stream,
                   const float* scale_shift = nullptr, float* z_out = nullptr) {
    Geom g;
    if (!geom<WM>(d, g) || !x || !v || channels <= 0 || channels % 4 || (scale_shift != nullptr) != (z_out != nullptr))
        return VSPW_EINVAL;
    const long long items = (long long)g.T * (channels / WinoVec<WM>::value);
    const dim3 grid((unsigned)((items + 255) / 256));
    if (scale_shift)
        hipLaunchKernelGGL((wino3_input_kernel<WM, true>), grid, dim3(256), 0, vspw_stream(stream), x, v, g, channels,
                           scale_shift, scale_shift + channels, z_out);
    else
        hipLaunchKernelGGL((wino3_input_kernel<WM, false>), grid, dim3(256), 0, vspw_stream(stream), x, v, g, channels,
                           nullptr, nullptr, nullptr);
    return vspw_launch_status();
}
template <int WM>
static int t_output(const vspw_conv_desc* d, const float* m, int channels, const float* bias, float* y,
                    const float* relu_src, const float* bn_y, const float* bn_mean, const float* bn_invstd,
                    float* stat_part, const float* addend, int act, void* stream) {
    Geom g;
    const int cl4 = cl4_of(channels);
    if (act != 0 && act != 1) return VSPW_EINVAL;
    if (relu_src != nullptr && (addend != nullptr || act != 0)) return VSPW_EINVAL;
    if (!geom<WM>(d, g) || !m || !y || cl4 == 0) return VSPW_EINVAL;
    const bool front = relu_src != nullptr;
    if (front && (!bn_y || !bn_mean || !bn_invstd || !stat_part)) return VSPW_EINVAL;
    const dim3 grid(vspw_cdiv(g.T, TB), channels / 4 / cl4);
    if (front)
        hipLaunchKernelGGL((wino3_output_kernel<WM, true>), grid, dim3(256), 0, vspw_stream(stream), m, bias, y, relu_src,
                           bn_y, bn_mean, bn_invstd, stat_part, nullptr, 0, g, channels, cl4);
    else
        hipLaunchKernelGGL((wino3_output_kernel<WM, false>), grid, dim3(256), 0, vspw_stream(stream), m, bias, y, nullptr,
                           nullptr, nullptr, nullptr, stat_part, addend, act, g, channels, cl4);
    return vspw_launch_status();
}
template <int WM>
static int t_dy(const vspw_conv_desc* d, const float* dy, int channels, float* dm, void* stream) {
    Geom g;
    if (!geom<WM>(d, g) || !dy || !dm || channels <= 0 || channels % 4) return VSPW_EINVAL;
    const long long items = (long long)g.T * (channels / WinoVec<WM>::value);
    hipLaunchKernelGGL(wino3_dy_kernel<WM>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, vspw_stream(stream), dy, dm,
                       g, channels);
    return vspw_launch_status();
}
template <int WM>
static int t_dw(const float* du, float* dw, int k, int c, void* stream) {
    if (!du || !dw || k <= 0 || c <= 0) return VSPW_EINVAL;
    const long long items = (long long)k * c;
    hipLaunchKernelGGL(wino3_dw_kernel<WM>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, vspw_stream(stream), du, dw,
                       k, c);
    return vspw_launch_status();
}

#define VSPW_WINO_M(NAME, WM)                                                                                              \
    extern "C" size_t vspw_##NAME##_supported(const vspw_conv_desc* d) { return t_supported<WM>(d); }                      \
    extern "C" long long vspw_##NAME##_tiles(const vspw_conv_desc* d) { return t_tiles<WM>(d); }                            \
    extern "C" size_t vspw_##NAME##_stat_partials(const vspw_conv_desc* d) {                                                \
        const long long t = t_tiles<WM>(d);                                                                                 \
        return t ? (size_t)vspw_cdiv(t, TB) : 0;                                                                            \
    }                                                                                                                       \
    extern "C" int vspw_##NAME##_weights(const float* w, float* u, int k, int c, int data_gradient, void* stream) {         \
        return t_weights<WM>(w, u, k, c, data_gradient, stream);                                                            \
    }                                                                                                                       \
    extern "C" int vspw_##NAME##_weights_multi(const vspw_wt_entry* e, int n, long long tiles, void* stream) {              \
        return t_weights_multi<WM>(e, n, tiles, stream);                                                                    \
    }                                                                                                                       \
    extern "C" int vspw_##NAME##_input(const vspw_conv_desc* d, const float* x, int channels, float* v, void* stream) {     \
        return t_input<WM>(d, x, channels, v, stream);                                                                      \
    }                                                                                                                       \
    extern "C" int vspw_##NAME##_input_apply(const vspw_conv_desc* d, const float* y, const float* scale_shift,             \
                                             float* z_out, int channels, float* v, void* stream) {                          \
        return t_input<WM>(d, y, channels, v, stream, scale_shift, z_out);                                                  \
    }                                                                                                                       \
    extern "C" int vspw_##NAME##_output(const vspw_conv_desc* d, const float* m, int channels, const float* bias, float* y, \
                                        const float* relu_src, const float* bn_y, const float* bn_mean,                     \
                                        const float* bn_invstd, float* stat_part, const float* addend, int act,             \
                                        void* stream) {                                                                     \
        return t_output<WM>(d, m, channels, bias, y, relu_src, bn_y, bn_mean, bn_invstd, stat_part, addend, act, stream);   \
    }                                                                                                                       \
    extern "C" int vspw_##NAME##_dy(const vspw_conv_desc* d, const float* dy, int channels, float* dm, void* stream) {      \
        return t_dy<WM>(d, dy, channels, dm, stream);                                                                       \
    }                                                                                                                       \
    extern "C" int vspw_##NAME##_dw(const float* du, float* dw, int k, int c, void* stream) {                               \
        return t_dw<WM>(du, dw, k, c, stream);                                                                              \
    }
VSPW_WINO_M(wino3, 3)
VSPW_WINO_M(wino4, 4)
VSPW_WINO_M(wino5, 5)
