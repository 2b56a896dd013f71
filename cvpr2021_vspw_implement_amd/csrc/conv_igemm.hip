// Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces, for the VSPW hot path, every ATen conv2d call site of the reference:
//   models/resnet.py:61-66,100-106,130 (stem / BasicBlock / Bottleneck convs),
//   models/models.py:737-750 (dilated / de-strided variants made by _nostride_dilate),
//   models/clip_psp.py:29,35,40,74,79; models/clip_ocr.py:43,56,58,62 (head convs),
//   models/ocr_modules/spatial_ocr_block.py:208-244,351 (1x1 convs),
// and their autograd backward (data gradient and weight gradient).
//
// Layout: activations NHWC fp32 ([pixel rows][channel cols]); weights [Cout][KH][KW][Cin]
// (torch channels_last of an OIHW tensor), so both GEMM operands of the forward pass are
// K-contiguous ("NT" GEMM):   Y[P][Cout] = im2col(X)[P][KH*KW*Cin] * W[Cout][KH*KW*Cin]^T
// The data gradient is the same kernel with the gather run "backwards" (mode 1) over dY and a
// [Cin][KH][KW][Cout] copy of the weights; the weight gradient is a "TN" GEMM whose reduction
// runs over pixels (split-K across workgroups, deterministic two-pass reduction).
//
// Tile: 128x128 outputs per 256-thread workgroup (4 waves, each 64x64 = 2x2 MFMA 32x32 tiles),
// BK = 32, operands staged global -> registers -> LDS (one LDS buffer, next tile's global loads
// are in flight during the MFMA phase). fp32 MFMA issues one instruction per 64 cycles per SIMD,
// so LDS and global bandwidth are far from binding; the kernel is MFMA-issue bound.
#include "common.h"
#include <cstdlib>
#include <type_traits>

#define BM 128
#define BN 128
#define BK 32
#define LDA 36  // padded LDS row stride (floats): 16 consecutive rows hit 16 distinct 16-B slots

#ifdef VSPW_NT_TIMING
// DIAGNOSTIC build (-DVSPW_NT_TIMING): per-workgroup s_memtime stamps [start, after prologue, after K loop, end] + CU id
// (-DVSPW_NT_TIMING=2: slot 1 is overwritten with "every store of the tile issued"); read by tools/diag/nt_phase.py
// (-DVSPW_NT_TIMING=3: the stamps are the 100 MHz wall clock instead - the same on every XCD, so a launch's first start and
// last end can be compared with the kernel's duration as rocprofv3 / HIP events see it; tools/diag/nt_balance.py)
__device__ unsigned long long vspw_nt_stamps[8192 * 5];
#if VSPW_NT_TIMING == 3
#define NT_CLOCK(i) wall_clock64()
#elif VSPW_NT_TIMING == 4  // calibration: slots 0 / 3 = s_memtime, slots 1 / 2 = wall clock at (nearly) the same moments
#define NT_CLOCK(i) (((i) == 1 || (i) == 2) ? wall_clock64() : __builtin_readcyclecounter())
#else
#define NT_CLOCK(i) __builtin_readcyclecounter()
#endif
#define NT_STAMP(i) if (threadIdx.x == 0 && stamp_slot < 8192) vspw_nt_stamps[stamp_slot * 5 + (i)] = NT_CLOCK(i)
#define TN_STAMP(i)                                                                       \
    if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 8192)                   \
    vspw_nt_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 5 + (i)] = NT_CLOCK(i)
#else
#define NT_STAMP(i)
#define TN_STAMP(i)
#endif

struct IgemmNT {
    const float* src;   // gathered tensor [nb][h][w][c]
    const float* wt;    // [nout][kdim]
    const float* bias;  // [nout] or nullptr
    float* dst;         // [m][ldd]
    float* stat_part;   // optional [tiles_m][2][nout] per-tile column sum / sumsq (fused BN stats)
    const float* addend;  // optional [m][ldd], added in the epilogue (skip-connection gradient folded into the dgrad)
    // Fused BatchNorm-backward front end (data gradient only): the tensor this launch produces is dL/dz of the
    // conv+BN+ReLU node that produced this conv's input z.  With relu_src = z and bn_y = that node's pre-BN conv output:
    //   g = (z > 0) ? dx : 0  is what gets stored, and stat_part receives the per-tile column sums of g and of
    //   g * (bn_y - bn_mean) * bn_invstd  - the two reductions of batch_norm backward - so that node needs no separate
    // reduction pass and no ReLU-mask read.  All three are [m][ldd] / [nout]; all or none are set.
    const float* relu_src;
    const float* bn_y;
    const float* bn_mean;
    const float* bn_invstd;
    int nb, h, w, c;    // src dims
    int oh, ow;         // pixel grid of the GEMM M dimension
    int kh, kw, stride, pad, padw, dil;  // pad: rows (H), padw: columns (W)
    int mode;           // 0: forward gather, 1: data-gradient gather
    int nout, ldd;
    int m, kdim;
    int vec;            // 1: float4 loads are legal (c % 4 == 0 and lds % 4 == 0)
    int lds;            // pixel stride of src in floats (c, or wider when src is a channel slice of a concat buffer)
    int act;            // epilogue activation: 0 none, 1 relu, 2 sigmoid, 3 tanh (inference-only entry point)
    // Affine A operand (pointwise data gradient only): the GEMM's A element is coef[0][k]*src + coef[1][k]*src2 + coef[2][k]
    // - BatchNorm's backward "apply" (dy = a*(g - mean(g) - xhat*mean(g*xhat))) evaluated while the operand is staged,
    // so that dy (the gradient w.r.t. the conv output) is never written to / re-read from memory.  src = g, src2 = the
    // node's pre-BN activations y; coef [3][kdim] from vspw_bn_bwd_affine_coeffs.
    const float* src2;
    const float* coef;
    // Forward "apply" fused into the NEXT pointwise convolution (FAP): this conv's input z is the previous node's
    // relu(scale*y + shift + residual), which has not been materialised: src = y, src2 = residual, coef = [2][kdim]
    // (scale, shift); the A operand is evaluated while it is staged and - by the workgroups of the first column tile -
    // also written to zout ([m][lds], the tensor every later reader of z uses).
    float* zout;
    // batched plain GEMM (vspw_bmm_nt): blockIdx.y = batch index, element strides of src / wt / dst between batches
    int batch;
    long long bs_src, bs_wt, bs_dst;
    // Winograd input operand (AFF 4, winograd.hip): the A matrix of batch xi = (a, b) is (B^T d B)[a][b] of the 4x4 input
    // patch of tile m - four +-1-weighted pixels of src - evaluated while the operand is staged, so the transformed
    // input V [16][T][c] is never written.  src = the NHWC image tensor (nb, h, w, lds), m = tiles, oh = tiles per image,
    // ow = 1; wino_d = dilation (0: off), wino_th x wino_tw = tiles per dilation sub-grid.  The grid is x-only:
    // 16 batches fastest, so that the 16 transforms of a tile's patch are read while it is L2-resident.
    int wino_d, wino_th, wino_tw;
    // Two-level accumulation (CHUNK variants of the v2 pointwise kernel): every chunk_tiles K-tiles the running sums are
    // folded into the output tile and the accumulators restart from zero; the epilogue adds the parked partial sums.
    // 0 = one k-sequential chain.  See vspw_set_accum_chunk.
    int chunk_tiles;
#ifdef VSPW_NT_DBG
    int dbg;  // diagnostic builds only (tools/diag/nt_exposed.py): 1 = no epilogue memory traffic, 2 = no K loop
#endif
};

__device__ __forceinline__ float nt_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + expf(-v));
    if (act == 3) return tanhf(v);
    return v;
}

__device__ __forceinline__ bool nt_src_coord(const IgemmNT& p, int by, int bx, int ky, int kx, int& sy, int& sx) {
    if (p.mode == 0) {
        sy = by + ky * p.dil;
        sx = bx + kx * p.dil;
        return sy >= 0 && sy < p.h && sx >= 0 && sx < p.w;
    } else {
        int ty = by - ky * p.dil;
        int tx = bx - kx * p.dil;
        if (ty < 0 || tx < 0) return false;
        if (p.stride != 1) {
            if ((ty % p.stride) != 0 || (tx % p.stride) != 0) return false;
            ty /= p.stride;
            tx /= p.stride;
        }
        sy = ty;
        sx = tx;
        return sy < p.h && sx < p.w;
    }
}

// (workgroup -> tile order: xcd_remap, common.h)

// WM x WN = MFMA 32x32 tiles per wave; 2x2 waves -> workgroup tile (64*WM) x (64*WN).
template <int WM, int WN>
__global__ __launch_bounds__(256) void igemm_nt_kernel(IgemmNT pin) {
    IgemmNT p = pin;
    if (pin.batch > 1) {
        p.src += (size_t)blockIdx.y * pin.bs_src;
        p.wt += (size_t)blockIdx.y * pin.bs_wt;
        p.dst += (size_t)blockIdx.y * pin.bs_dst;
    }
    constexpr int TM = 64 * WM, TN = 64 * WN;
    constexpr int RA = TM / 32, RB = TN / 32;  // rows staged per thread
    __shared__ __attribute__((aligned(16))) float As[TM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[TN * LDA];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tiles_n = (p.nout + TN - 1) / TN;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = vb % tiles_n;
    const int tile_m = vb / tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;

    const int lrow = tid >> 3;       // 0..31
    const int lcol = (tid & 7) * 4;  // 0,4,..,28

    // Row (pixel) decode for the A rows this thread stages; independent of k.
    int a_img[RA], a_by[RA], a_bx[RA];
    bool a_ok[RA];
    const int ohw = p.oh * p.ow;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + lrow + 32 * i;
        a_ok[i] = m < p.m;
        int mm = a_ok[i] ? m : 0;
        int n = mm / ohw;
        int r = mm - n * ohw;
        int oy = r / p.ow;
        int ox = r - oy * p.ow;
        a_img[i] = n;
        if (p.mode == 0) {
            a_by[i] = oy * p.stride - p.pad;
            a_bx[i] = ox * p.stride - p.padw;
        } else {
            a_by[i] = oy + p.pad;
            a_bx[i] = ox + p.padw;
        }
    }
    bool b_ok[RB];
    size_t b_off[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        int n = n0 + lrow + 32 * i;
        b_ok[i] = n < p.nout;
        b_off[i] = (size_t)(b_ok[i] ? n : 0) * (size_t)p.kdim;
    }

    f32x4 ra[RA], rb[RB];

    auto load_tile = [&](int k0) {
        const int k4 = k0 + lcol;
        if (p.vec) {
            const bool kin = k4 < p.kdim;  // kdim % 4 == 0 in the vector path
            int tap = 0, ci = 0, ky = 0, kx = 0;
            if (kin) {
                tap = k4 / p.c;
                ci = k4 - tap * p.c;
                ky = tap / p.kw;
                kx = tap - ky * p.kw;
            }
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                int sy, sx;
                if (kin && a_ok[i] && nt_src_coord(p, a_by[i], a_bx[i], ky, kx, sy, sx)) {
                    size_t off = (((size_t)a_img[i] * p.h + sy) * p.w + sx) * (size_t)p.lds + ci;
                    v = *reinterpret_cast<const f32x4*>(p.src + off);
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                f32x4 wv = {0.f, 0.f, 0.f, 0.f};
                if (kin && b_ok[i]) wv = *reinterpret_cast<const f32x4*>(p.wt + b_off[i] + k4);
                rb[i] = wv;
            }
        } else {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int k = k4 + e;
                    if (k < p.kdim) {
                        int tap = k / p.c;
                        int ci = k - tap * p.c;
                        int ky = tap / p.kw;
                        int kx = tap - ky * p.kw;
                        int sy, sx;
                        if (a_ok[i] && nt_src_coord(p, a_by[i], a_bx[i], ky, kx, sy, sx)) {
                            size_t off = (((size_t)a_img[i] * p.h + sy) * p.w + sx) * (size_t)p.lds + ci;
                            v[e] = p.src[off];
                        }
                    }
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                f32x4 wv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k4 + e < p.kdim && b_ok[i]) wv[e] = p.wt[b_off[i] + k4 + e];
                rb[i] = wv;
            }
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.kdim + BK - 1) / BK;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&As[(lrow + 32 * i) * LDA + lcol]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * i) * LDA + lcol]) = rb[i];
        __syncthreads();
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        // Each lane half (lh) owns 4 consecutive k of every 8-k chunk; MFMA step s pairs
        // k = 8*kc + s (lanes 0-31) with k = 8*kc + 4 + s (lanes 32-63) for both operands.
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            f32x4 a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(&As[(wm * 32 * WM + i * 32 + l31) * LDA + kc * 8 + 4 * lh]);
#pragma unroll
            for (int j = 0; j < WN; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn * 32 * WN + j * 32 + l31) * LDA + kc * 8 + 4 * lh]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // Epilogue. C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    float csum[WN], csq[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        csum[j] = 0.f;
        csq[j] = 0.f;
        const int col = n0 + wn * 32 * WN + j * 32 + l31;
        const bool cok = col < p.nout;
        const float bv = (p.bias != nullptr && cok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < p.m) {
                    float t = acc[i][j][r] + bv;
                    if (p.addend != nullptr) t += p.addend[(size_t)row * p.ldd + col];
                    const float v = nt_act(t, p.act);  // act(conv + bias + addend): residual add, then ReLU
                    p.dst[(size_t)row * p.ldd + col] = v;
                    csum[j] += v;
                    csq[j] += v * v;
                }
            }
        }
    }
    if (p.stat_part != nullptr) {
        // Per-tile column partial sums: lanes l and l^32 share a column; so do waves wm=0/1.
        float* red = As;  // reuse LDS: [2 stats][2 wm][TN cols]  (TM*LDA >= 4*TN always)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float s = csum[j] + __shfl_xor(csum[j], 32, 64);
            float q = csq[j] + __shfl_xor(csq[j], 32, 64);
            if (lh == 0) {
                int c = wn * 32 * WN + j * 32 + l31;
                red[(0 * 2 + wm) * TN + c] = s;
                red[(1 * 2 + wm) * TN + c] = q;
            }
        }
        __syncthreads();
        if (tid < TN) {
            int col = n0 + tid;
            if (col < p.nout) {
                float* out = p.stat_part + (size_t)tile_m * 2 * p.nout;
                out[col] = red[0 * TN + tid] + red[1 * TN + tid];
                out[p.nout + col] = red[2 * TN + tid] + red[3 * TN + tid];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// v2 of the NT kernel for Cin % 32 == 0 (every conv of the path except the 3-channel stem conv), forward or stride-1
// data gradient.  A K-tile (BK = 32 channels) never straddles a filter tap, so the tap (ky, kx) and the channel base
// are WORKGROUP-UNIFORM: they live in scalar registers, and each thread caches, per tap, the byte offset of the
// (clamped) source pixel of each of its rows plus an in-image bit.  The K loop then contains no per-load address
// arithmetic at all - a load is `scalar base + cached 32-bit offset` - and the loop body is straight-line apart from
// the uniform (scalar-branch) "tap changed" refresh every Cin/32 iterations.  Measured on the GEMM probe
// (tools/probe/mfma_ablate.hip): VALU work in the loop is not hidden behind the fp32 MFMAs, it is added to them.
//   * NBUF 2: two LDS buffers, ONE barrier per K-tile; NBUF 1: one buffer, two barriers, half the LDS (residency).
//   * operands arrive through raw buffer loads: an out-of-image tap is an out-of-range offset that the hardware bounds
//     check turns into zeros (no select on the data); rows >= M and columns >= Cout read clamped (valid) addresses and
//     are simply never stored.  The K loops contain NO vector-ALU instruction: fp32 MFMA executes on the SIMD's f32
//     vector lanes, so VALU work in the loop is not hidden behind the MFMAs, it is subtracted from them.
// MODE 0: forward gather (any stride);  MODE 1: data-gradient gather (any stride);  MODE 2: pointwise.
// WGM = waves along M (2: 2x2 waves, 1: 1x4 waves); workgroup tile = (32*WM*WGM) x (32*WN*(4/WGM)).
// Register budget: tiles with <= 48 accumulator registers per lane are compiled for 4 waves per SIMD (<= 128 unified
// registers, accumulators in VGPRs, no spills - checked with -Rpass-analysis=kernel-resource-usage); the 128x128 tile
// (64 accumulators) would spill under that cap and stays at 3.  Measured: +1 % on the 3x3 shapes; the pointwise
// shapes lose 4-5 % with the fourth wave (more L2 pressure per CU), so MODE 2 stays at 3 as well.
// TAPS = 9: K order "channel slab outer, filter tap inner" for 3x3 filters.  In the default order (tap outer) the tiles
// resident on one XCD sweep their whole input window once per tap - ~4.7 MB for the layer3 shapes, just over the 4 MB
// L2, so every tap pass re-fetched it from the Infinity Cache (measured 261 MB fetched for 39 MB of operands).  With
// the tap inside, one 32-channel slab of the window (1/8 of it) is reused by all nine taps while it sits in L2.  The
// nine per-tap offsets of each staged row are computed ONCE (9*RA registers, hence 3 waves per SIMD), the K loop is
// unrolled over the taps, and nothing is left of the per-tap refresh: its K loop has no VALU instruction at all.
// Wave priority outside the K loop.  fp32 MFMA occupies the SIMD's vector lanes for 64 cycles an instruction, so a
// wave in its prologue (address arithmetic) or epilogue (a few hundred dependent VALU / LDS / memory instructions)
// that shares a SIMD with two waves issuing MFMAs back to back gets one instruction in per MFMA: measured 8-16 k cycles
// of prologue and 20-60 k cycles of epilogue per tile, during which the workgroup's registers and LDS are held without
// feeding the matrix pipe.  Raised priority lets those phases issue back to back (the MFMA pipe loses the same issue
// slots either way) and frees the slot for the next tile sooner.
#ifndef NT_PRIO_EDGE
#define NT_PRIO_EDGE 2
#endif
#define NT_PRIO(x) __builtin_amdgcn_s_setprio(x)
// Progress-dependent wave priority inside the K loop (experiment, -DVSPW_PROGRESS_PRIO): the SIMD arbitrates equal
// priorities oldest-first, so the three co-resident workgroups of a single-round launch finish one after the other
// (stamps, tools/diag/nt_balance.py: lives of 150 k / 230 k / 325 k cycles on one CU) and the last one runs alone, one
// wave per SIMD, for the final third of the launch.  Priority falling with progress lets the laggards catch up.
#ifdef VSPW_PROGRESS_PRIO
#define NT_PROGRESS_PRIO(kt, nk)                                              \
    do {                                                                      \
        const int q_ = (nk) >> 2;                                             \
        if ((kt) == 0) NT_PRIO(3);                                            \
        else if ((kt) == q_) NT_PRIO(2);                                      \
        else if ((kt) == 2 * q_) NT_PRIO(1);                                  \
        else if ((kt) == 3 * q_) NT_PRIO(0);                                  \
    } while (0)
#else
#define NT_PROGRESS_PRIO(kt, nk)
#endif

// (the tap-outer gather variants - MODE 0 / 1, TAPS 0 - used to ask for 4 workgroups per CU: 128 registers, 3-10 of them
// spilled to scratch; at 3 they fit)
#define NT_V2_BOUNDS(WM, WN, MODE, NBUF, TAPS, AFF) ((NBUF == 1 && !(AFF && WM * WN > 3)) ? 3 : 2)

// One output tile.  vb_in = tile index of this workgroup (already XCD-remapped), batch_idx = batch of a batched GEMM,
// stamp_slot = slot of the diagnostic stamps.  Called once per workgroup by igemm_nt_v2_kernel and in a loop by the
// kernel below.
template <int WGM, int WM, int WN, int MODE, int NBUF, int TAPS, int AFF, int CHUNK = 0, int FOLD = 0>
__device__ __forceinline__ void igemm_nt_v2_body(const IgemmNT& pin, int vb_in, int batch_idx, int stamp_slot) {
    IgemmNT p = pin;
    if (AFF != 4 && pin.batch > 1) {
        p.src += (size_t)batch_idx * pin.bs_src;
        p.wt += (size_t)batch_idx * pin.bs_wt;
        p.dst += (size_t)batch_idx * pin.bs_dst;
    }
    (void)stamp_slot;
    static_assert(TAPS == 0 || (NBUF == 1 && MODE != 2), "tap-inner order: single LDS buffer, non-pointwise");
    static_assert(!AFF || (MODE == 2 && NBUF == 1), "transformed A operand: pointwise, single LDS buffer");
    static_assert(AFF != 4 || WM * WN <= 3, "Winograd operand: 4 staged float4 per row - the 96- / 64-row tiles only");
    static_assert(!CHUNK || (MODE == 2 && TAPS == 0 && AFF != 4), "two-level accumulation: plain pointwise K loop only");
    static_assert(!CHUNK || NBUF == 1 || 32 * WM * WGM >= 128, "flush scratch of 4 waves must fit ONE operand buffer");
    static_assert(FOLD == 0 || (TAPS > 0 && TAPS % FOLD == 0 && WM * WN == 1), "folded chains: tap-inner loop, the 64x64 tile");
    // AFF 1: A = coef0*src + coef1*src2 + coef2 (BatchNorm-backward apply); AFF 2: A = relu(coef0*src + coef1 + src2);
    // AFF 3: A = relu(coef0*src + coef1) (no residual)
    // (BatchNorm-forward apply + residual + ReLU of the producing node), also stored to zout
    NT_STAMP(0);
    NT_PRIO(NT_PRIO_EDGE);
    constexpr int WGN = 4 / WGM;
    constexpr int TM = 32 * WM * WGM, TN = 32 * WN * WGN;
    constexpr int RA = TM / 32, RB = TN / 32;
    // one LDS block: the operand tiles of the K loop, reused by the epilogue as per-wave transposition scratch
    __shared__ __attribute__((aligned(16))) float smem[NBUF * (TM + TN) * LDA];
    static_assert(NBUF * (TM + TN) * LDA >= 4 * 32 * LDA, "epilogue scratch: 32 x LDA floats per wave");
    float (*As)[TM * LDA] = reinterpret_cast<float (*)[TM * LDA]>(smem);
    float (*Bs)[TN * LDA] = reinterpret_cast<float (*)[TN * LDA]>(smem + NBUF * TM * LDA);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (WGM == 2) ? (wave >> 1) : 0, wn = (WGM == 2) ? (wave & 1) : wave;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tiles_n = (p.nout + TN - 1) / TN;
    int vb = vb_in;
    int xi = 0;
    if constexpr (AFF == 4) {  // batch index = the 4 low bits of the (remapped) workgroup id
        xi = vb & 15;
        vb >>= 4;
        p.wt += (size_t)xi * pin.bs_wt;
        p.dst += (size_t)xi * pin.bs_dst;
    }
    const int tile_n = vb % tiles_n;
    const int tile_m = vb / tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;

    const int lrow = tid >> 3;
    const int lcol = (tid & 7) * 4;

    // offsets are relative to the first image this tile touches (uniform 64-bit base), so 32-bit byte offsets suffice
    const int ohw = p.oh * p.ow;
    const int img0 = m0 / ohw;
    const char* src0 = reinterpret_cast<const char*>(p.src + (size_t)img0 * p.h * p.w * p.lds);
    const char* wt0 = reinterpret_cast<const char*>(p.wt + (size_t)n0 * p.kdim);
    // Operand loads are RAW BUFFER loads (base in a scalar descriptor, 32-bit byte offset per lane, channel / k base as
    // the scalar offset): the hardware bounds check returns 0 for an offset >= num_records, so a padding tap is just
    // an out-of-range offset (NT_OOR) - no select on the loaded data.  That matters here more than anywhere: fp32 MFMA
    // runs on the SIMD's f32 vector lanes, so every VALU instruction in the K loop is taken out of MFMA issue time
    // (measured: the 12 v_cndmask + 6 v_and/v_cmp per K-tile of the select form cost 4-6 % on every 3x3 shape).
    constexpr unsigned NT_OOR = 0x80000000u;
    const long long a_rem = (long long)(p.nb - img0) * p.h * p.w * p.lds * 4;
    const long long b_rem = (long long)(p.nout - n0) * p.kdim * 4;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(src0), 0, (int)(unsigned)(a_rem < (long long)NT_OOR ? a_rem : (long long)NT_OOR), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wt0), 0, (int)(unsigned)(b_rem < (long long)NT_OOR ? b_rem : (long long)NT_OOR), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>((AFF == 1 || AFF == 2) ? p.src2 + (size_t)img0 * p.h * p.w * p.lds : p.src)), 0,
        (int)(unsigned)(a_rem < (long long)NT_OOR ? a_rem : (long long)NT_OOR), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>((AFF && AFF != 4) ? p.coef : p.wt), 0, 3 * p.kdim * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>((AFF == 2 || AFF == 3) ? p.zout + (size_t)img0 * p.h * p.w * p.lds : p.dst), 0,
        (int)(unsigned)(a_rem < (long long)NT_OOR ? a_rem : (long long)NT_OOR), 0x00020000);
    // MODE 2 = pointwise at compile time (1x1, stride 1, no padding: source pixel == output pixel, every tap in the
    // image): no tap state, no in-image bits, no selects - the K loop is a plain GEMM loop
    constexpr bool PW = MODE == 2;
    const bool pointwise = PW || ((p.kh * p.kw == 1) & (p.stride == 1) & (p.pad == 0) & (p.padw == 0));

    int a_base[RA], a_by[RA], a_bx[RA];
    unsigned a_voff[RA];    // byte offset (from src0) of this row's source pixel for the current tap, + lcol; NT_OOR = pad
    unsigned a_w4[AFF == 4 ? 4 : 1][RA];  // AFF 4: the four patch pixels (2 rows x 2 columns of B^T d B's [a][b] entry)
    float w4c[4] = {0.f, 0.f, 0.f, 0.f};  //        and their +-1 coefficients (workgroup-uniform)
    if constexpr (AFF == 4) {
        // B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]: row a has two non-zeros, at patch rows I0[a], I1[a]
        const int a = xi >> 2, b = xi & 3;
        const int ia0 = a == 0 ? 0 : 1, ia1 = a == 0 ? 2 : (a == 3 ? 3 : 2);
        const int jb0 = b == 0 ? 0 : 1, jb1 = b == 0 ? 2 : (b == 3 ? 3 : 2);
        const float sa0 = a == 2 ? -1.f : 1.f, sa1 = (a == 0 || a == 3) ? -1.f : 1.f;
        const float sb0 = b == 2 ? -1.f : 1.f, sb1 = (b == 0 || b == 3) ? -1.f : 1.f;
        w4c[0] = sa0 * sb0; w4c[1] = sa0 * sb1; w4c[2] = sa1 * sb0; w4c[3] = sa1 * sb1;
        const int per = p.wino_th * p.wino_tw;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int t = min(m0 + lrow + 32 * i, p.m - 1);
            const int img = t / ohw;
            int r = t - img * ohw;
            const int sg = r / per;
            r -= sg * per;
            const int sy = sg / p.wino_d, sx = sg - sy * p.wino_d;
            const int ty = r / p.wino_tw, tx = r - ty * p.wino_tw;
            const int gy0 = 2 * ty - 1 + ia0, gy1 = 2 * ty - 1 + ia1, gx0 = 2 * tx - 1 + jb0, gx1 = 2 * tx - 1 + jb1;
            const int py0 = gy0 * p.wino_d + sy, py1 = gy1 * p.wino_d + sy;
            const int px0 = gx0 * p.wino_d + sx, px1 = gx1 * p.wino_d + sx;
            const bool oy0 = (gy0 >= 0) & (py0 < p.h), oy1 = (gy1 >= 0) & (py1 < p.h);
            const bool ox0 = (gx0 >= 0) & (px0 < p.w), ox1 = (gx1 >= 0) & (px1 < p.w);
            const int rb0 = ((img - img0) * p.h + py0) * p.w, rb1 = ((img - img0) * p.h + py1) * p.w;
            a_w4[0][i] = (oy0 & ox0) ? (unsigned)((rb0 + px0) * p.lds + lcol) * 4u : NT_OOR;
            a_w4[1][i] = (oy0 & ox1) ? (unsigned)((rb0 + px1) * p.lds + lcol) * 4u : NT_OOR;
            a_w4[2][i] = (oy1 & ox0) ? (unsigned)((rb1 + px0) * p.lds + lcol) * 4u : NT_OOR;
            a_w4[3][i] = (oy1 & ox1) ? (unsigned)((rb1 + px1) * p.lds + lcol) * 4u : NT_OOR;
            a_voff[i] = 0;
            a_base[i] = a_by[i] = a_bx[i] = 0;
        }
    } else if (pointwise) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = min(m0 + lrow + 32 * i, p.m - 1) - img0 * ohw;
            a_voff[i] = (unsigned)(m * p.lds + lcol) * 4u;
            a_base[i] = a_by[i] = a_bx[i] = 0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int mm = min(m0 + lrow + 32 * i, p.m - 1);
            const int n = mm / ohw;
            const int r = mm - n * ohw;
            const int oy = r / p.ow;
            const int ox = r - oy * p.ow;
            a_base[i] = (n - img0) * p.h;
            if (MODE == 0) {
                a_by[i] = oy * p.stride - p.pad;
                a_bx[i] = ox * p.stride - p.padw;
            } else {
                a_by[i] = oy + p.pad;
                a_bx[i] = ox + p.padw;
            }
        }
    }
    const int sgn = (MODE == 0) ? p.dil : -p.dil;
    auto tap_offsets = [&](int ky, int kx, unsigned* out) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int sy = a_by[i] + sgn * ky;
            int sx = a_bx[i] + sgn * kx;
            bool ok = true;
            if (MODE == 1 && p.stride != 1) {
                // strided data gradient: tap (ky, kx) reaches this input pixel only from output pixel
                // ((iy + pad - ky*dil) / stride, ...) when the division is exact (uniform branch, taken per tap)
                ok = (sy >= 0) & (sx >= 0) & (sy % p.stride == 0) & (sx % p.stride == 0);
                sy = sy >= 0 ? sy / p.stride : -1;
                sx = sx >= 0 ? sx / p.stride : -1;
            }
            ok = ok & ((unsigned)sy < (unsigned)p.h) & ((unsigned)sx < (unsigned)p.w);
            const unsigned off = (unsigned)(((a_base[i] + sy) * p.w + sx) * p.lds + lcol) * 4u;
            out[i] = ok ? off : NT_OOR;
        }
    };
    auto set_tap = [&](int ky, int kx) { tap_offsets(ky, kx, a_voff); };
    if (!pointwise) set_tap(0, 0);
    unsigned b_voff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) b_voff[i] = (unsigned)(min(lrow + 32 * i, p.nout - 1 - n0) * p.kdim + lcol) * 4u;

    // uniform K state: kb = k offset of the tile the next load fetches, cb = its channel base inside tap (ky, kx)
    int kb = 0, cb = 0, ky = 0, kx = 0;
    bool more = true;
    auto advance = [&]() {
        kb += BK;
        cb += BK;
        more = kb < p.kdim;
        if (!PW && more && cb == p.c) {
            cb = 0;
            if (++kx == p.kw) {
                kx = 0;
                ++ky;
            }
            set_tap(ky, kx);
        }
    };
    f32x4 ra[RA], rb[RB];
    f32x4 ra2[(AFF == 1 || AFF == 2) ? RA : 1], cf[3];
    f32x4 rw[AFF == 4 ? 3 : 1][RA];  // AFF 4: patch pixels 1..3 (pixel 0 sits in ra)
    int kb_regs = 0;  // k base of the tile currently held in the staging registers (AFF 2: where its z goes)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto load_tile = [&]() {
        if (more) {
            if constexpr (AFF == 4) {
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_w4[0][i], cb * 4, 0));
#pragma unroll
                    for (int q = 1; q < 4; ++q)
                        rw[q - 1][i] =
                            __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_w4[q][i], cb * 4, 0));
                }
            } else {
#pragma unroll
            for (int i = 0; i < RA; ++i)
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[i], cb * 4, 0));
            }
            if (AFF && AFF != 4) {
                if (AFF != 3) {
#pragma unroll
                    for (int i = 0; i < RA; ++i)
                        ra2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a2, a_voff[i], cb * 4, 0));
                }
#pragma unroll
                for (int e = 0; e < (AFF >= 2 ? 2 : 3); ++e)
                    cf[e] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_c, (unsigned)(e * p.kdim + lcol) * 4u, kb * 4, 0));
                kb_regs = kb;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i)
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff[i], kb * 4, 0));
        }
    };
    auto store_tile = [&](float* Ad, float* Bd) {
        if (AFF == 1) {  // the only vector-ALU work of this loop: 8 FMAs per staged float4
#pragma unroll
            for (int i = 0; i < RA; ++i) ra[i] = cf[0] * ra[i] + (cf[1] * ra2[i] + cf[2]);
        }
        if constexpr (AFF == 4) {  // (B^T d B)[a][b]: four pixels with +-1 coefficients
#pragma unroll
            for (int i = 0; i < RA; ++i)
                ra[i] = (w4c[0] * ra[i] + w4c[1] * rw[0][i]) + (w4c[2] * rw[1][i] + w4c[3] * rw[2][i]);
        }
        if (AFF == 2 || AFF == 3) {  // same expression tree as bn_apply_kernel: (y*scale + shift) [+ residual], then ReLU
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                f32x4 v = ra[i] * cf[0] + cf[1];
                if (AFF == 2) v += ra2[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                ra[i] = v;
                if (tile_n == 0)  // one column tile materialises z for the node's other readers (uniform branch)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_z, a_voff[i], kb_regs * 4, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&Ad[(lrow + 32 * i) * LDA + lcol]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(&Bd[(lrow + 32 * i) * LDA + lcol]) = rb[i];
    };

    f32x16 acc[WM][WN];
    // interior tile (uniform): unguarded, 16-byte-per-lane epilogue (below)
    const bool interior = (m0 + TM <= p.m) & (n0 + TN <= p.nout) & (p.act <= 1) & ((p.ldd & 3) == 0) &
                          (WM * WN < 4 || p.relu_src == nullptr) &
                          ((((size_t)p.dst | (size_t)p.addend | (size_t)p.relu_src | (size_t)p.bn_y | (size_t)p.bias |
                             (size_t)p.bn_mean | (size_t)p.bn_invstd) & 15) == 0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- two-level accumulation (CHUNK) ----
    // A k-sequential fp32 chain of K terms carries a rounding error ~ eps*sqrt(K/2) of the result; summed as K/b chains
    // of b terms it is ~ eps*sqrt(b/2 + K/(2b)) (the shape of a k-blocked CPU GEMM, which is what the reference runs).
    // Measured on the raw-weight full-size fixtures (profiles/r05_parity_attrib.log): the k-sequential chains of the
    // K >= 1024 pointwise convolutions were the whole 1.3-1.5x excess of |hip - ref64| over the reference's own
    // |ref32 - ref64|; Winograd on / off made no difference.  A second accumulator set costs a resident workgroup per CU,
    // so the partial sums are parked where they belong anyway: every chunk_tiles K-tiles the block is added into the
    // OUTPUT tile (first flush: = chain [+ the caller's addend]) through the epilogue's 16-byte LDS-transposed path and
    // the accumulators restart from zero; the final epilogue then takes dst as its addend.  Same lanes, same addresses,
    // one workgroup per tile: plain loads / stores, deterministic.
    const int nk_all = p.kdim / BK;
    const bool chunked = CHUNK && p.chunk_tiles > 0 && nk_all > p.chunk_tiles;
    auto flush_partial = [&](const float* prev, float* scr_base) {
        // (opaque copies: keeps the compiler from hoisting the flush's address arithmetic out of the K loop, where it
        // would be spilled for the whole loop's duration)
        int m0 = tile_m * TM, n0 = tile_n * TN;
        asm volatile("" : "+s"(m0), "+s"(n0));
        if (interior) {
            float* scr = scr_base + wave * (32 * LDA);
            const int erow = lane >> 3, ec4 = (lane & 7) * 4;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = n0 + wn * 32 * WN + j * 32 + ec4;
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const size_t e0 = (size_t)(m0 + wm * 32 * WM + i * 32 + erow) * p.ldd + col;
                    const size_t estep = (size_t)8 * p.ldd;
                    f32x4 ad[4];
                    if (prev != nullptr) {
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) ad[rq] = *reinterpret_cast<const f32x4*>(prev + e0 + rq * estep);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDA + l31] = acc[i][j][r];
                        acc[i][j][r] = 0.f;
                    }
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        f32x4 o = *reinterpret_cast<const f32x4*>(&scr[(rq * 8 + erow) * LDA + ec4]);
                        if (prev != nullptr) o += ad[rq];
                        *reinterpret_cast<f32x4*>(p.dst + e0 + rq * estep) = o;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = n0 + wn * 32 * WN + j * 32 + l31;
#pragma unroll
                for (int i = 0; i < WM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (col < p.nout && row < p.m) {
                            float v = acc[i][j][r];
                            if (prev != nullptr) v += prev[(size_t)row * p.ldd + col];
                            p.dst[(size_t)row * p.ldd + col] = v;
                        }
                        acc[i][j][r] = 0.f;
                    }
                }
            }
        }
        __syncthreads();  // the scratch is operand space: nobody stages the next tile into it before every wave is done
    };
    (void)flush_partial;
    int chunk_left = chunked ? p.chunk_tiles : 0x7fffffff;
    bool flushed = false;
    // ---- folded chains (FOLD, the direct 3x3 kernels) ----
    // The 3x3 convolutions that do not go through Winograd - the stem, layer1, the strided ones: 64 / 128 channels, a
    // 576- / 1152-term reduction at the START of the network, where a rounding error is amplified by every BatchNorm'd
    // residual block that follows - carried the whole parity excess of the raw-weight fixtures
    // (profiles/r05_parity_attrib_c_direct3x3.log: |hip - ref64| / |ref32 - ref64| 1.30-1.63 -> 0.80-1.02 with chains of
    // 96 terms).  They run on the 64x64 tile (16 accumulators, see nt_folds), so here the second level is a second
    // accumulator set in registers: every FOLD K-tiles (3 = one filter row of a 32-channel slab) acc2 += acc and the
    // chain restarts.
    f32x16 acc2[FOLD > 0 ? WM : 1][FOLD > 0 ? WN : 1];
    if constexpr (FOLD > 0) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    }
    (void)acc2;
    NT_STAMP(1);
    NT_PRIO(0);
    if constexpr (TAPS > 0) {
        // ---- channel-slab-outer / tap-inner K loop (see the template comment) ----
        unsigned a_toff[TAPS][RA];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) tap_offsets(t / 3, t % 3, a_toff[t]);
#ifdef VSPW_NT_DBG
        const int nslab = (p.dbg & 2) ? 0 : p.c / BK;
#else
        const int nslab = p.c / BK;
#endif
        auto load_kt = [&](int t, int cs) {  // t is a compile-time constant at every call site
#pragma unroll
            for (int i = 0; i < RA; ++i)
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_toff[t][i], cs * (BK * 4), 0));
#pragma unroll
            for (int i = 0; i < RB; ++i)
                rb[i] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff[i], (t * p.c + cs * BK) * 4, 0));
        };
        load_kt(0, 0);
        for (int cs = 0; cs < nslab; ++cs) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                store_tile(As[0], Bs[0]);
                __syncthreads();
                const float* Ac = As[0];
                const float* Bc = Bs[0];
#pragma unroll
                for (int kc = 0; kc < BK / 8; ++kc) {
                    f32x4 fa[WM], fb[WN];
#pragma unroll
                    for (int i = 0; i < WM; ++i)
                        fa[i] = *reinterpret_cast<const f32x4*>(&Ac[(wm * 32 * WM + i * 32 + l31) * LDA + kc * 8 + 4 * lh]);
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        fb[j] = *reinterpret_cast<const f32x4*>(&Bc[(wn * 32 * WN + j * 32 + l31) * LDA + kc * 8 + 4 * lh]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < WM; ++i)
#pragma unroll
                            for (int j = 0; j < WN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
                    if (kc == 0) {  // next K-tile: next tap of this slab, or tap 0 of the next slab
                        if (t + 1 < TAPS)
                            load_kt(t + 1 < TAPS ? t + 1 : 0, cs);
                        else if (cs + 1 < nslab)
                            load_kt(0, cs + 1);
                    }
                }
                __syncthreads();
                if constexpr (FOLD > 0) {
                    if ((t + 1) % FOLD == 0) {  // (compile-time: t is an unrolled index, TAPS % FOLD == 0)
#pragma unroll
                        for (int i = 0; i < WM; ++i)
#pragma unroll
                            for (int j = 0; j < WN; ++j) {
                                acc2[i][j] += acc[i][j];
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                            }
                    }
                }
            }
        }
    } else {
#ifdef VSPW_NT_DBG
    const int nk = (p.dbg & 2) ? 0 : p.kdim / BK;
#else
    const int nk = p.kdim / BK;
#endif
    // prologue: tile 0 -> LDS[0]; tile 1 -> registers
    load_tile();
    if (NBUF == 2) {
        store_tile(As[0], Bs[0]);
        advance();  // state now describes tile 1 (tiles past kdim are not fetched; stale registers are never consumed)
        load_tile();
        __syncthreads();
    }

    for (int kt = 0; kt < nk; ++kt) {
        NT_PROGRESS_PRIO(kt, nk);
        const int cur = (NBUF == 2) ? (kt & 1) : 0;
        if (NBUF == 1) {
            store_tile(As[0], Bs[0]);  // tile kt (loaded during iteration kt-1)
            __syncthreads();
            advance();
        }
        const float* Ac = As[cur];
        const float* Bc = Bs[cur];
        // fragments of k-group kc+1 are fetched while group kc's MFMAs issue (VSPW_FRAG_PREFETCH): the wave never sits
        // in an LDS-latency wait between groups
        f32x4 fa[2][WM], fb[2][WN];
        auto load_frag = [&](int kc, int slot) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
                fa[slot][i] = *reinterpret_cast<const f32x4*>(&Ac[(wm * 32 * WM + i * 32 + l31) * LDA + kc * 8 + 4 * lh]);
#pragma unroll
            for (int j = 0; j < WN; ++j)
                fb[slot][j] = *reinterpret_cast<const f32x4*>(&Bc[(wn * 32 * WN + j * 32 + l31) * LDA + kc * 8 + 4 * lh]);
        };
#ifdef VSPW_FRAG_PREFETCH
        load_frag(0, 0);
#endif
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
#ifdef VSPW_FRAG_PREFETCH
            if (kc + 1 < BK / 8) load_frag(kc + 1, (kc + 1) & 1);
            const int sl = kc & 1;
#else
            load_frag(kc, 0);
            const int sl = 0;
#endif
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sl][i][s], fb[sl][j][s], acc[i][j], 0, 0, 0);
            if (kc == 0) {
                if (NBUF == 2) {
                    // tile kt+1 has been in flight since the middle of the previous iteration: registers -> the
                    // other LDS buffer (its last readers passed the barrier that ended iteration kt-1), then reuse
                    // the registers for tile kt+2, which gets a whole iteration to land.
                    store_tile(As[(NBUF - 1) & (cur ^ 1)], Bs[(NBUF - 1) & (cur ^ 1)]);
                    advance();
                }
                load_tile();  // NBUF 1: tile kt+1, consumed at the top of the next iteration
            }
        }
        __syncthreads();
        if constexpr (CHUNK) {
            if (--chunk_left == 0 && kt + 1 < nk) {  // (uniform) tile kt's operand buffer is dead: the flush scratch
                flush_partial(flushed ? p.dst : p.addend, As[cur]);
                flushed = true;
                chunk_left = p.chunk_tiles;
            }
        }
    }
    }  // tap-outer order
    if constexpr (CHUNK) {
        if (flushed) p.addend = p.dst;  // the epilogue adds the parked partial sums (they already hold the caller's addend)
    }
    if constexpr (FOLD > 0) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] += acc2[i][j];
    }
    NT_STAMP(2);
    NT_PRIO(NT_PRIO_EDGE);
#ifdef VSPW_NT_DBG
    if (p.dbg & 1) {  // no epilogue traffic: keep the accumulators alive through a never-true store
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 123.456f) p.dst[tid] = t;
        return;
    }
#endif

    float csum[WN], csq[WN];
    f32x4 cs4[WN], cq4[WN];
    if (interior) {
        // The accumulator layout (lane = column, registers = rows) would make every global access of the epilogue a
        // 4-byte-per-lane instruction - 16 per 32x32 block and stream, and the vector-memory pipe issues those no
        // faster than 16-byte ones: with three workgroups per CU the stores (and the skip / BatchNorm-front loads) of
        // one workgroup held up the operand loads of the other two (measured: 163 us with, 135 us without the store
        // phase on the 1024->256 data gradient, and no overlap between the two).  So each 32x32 block goes through a
        // wave-private LDS scratch (the operand tiles are dead: every wave is past the K loop's last barrier) and comes
        // back with a lane owning 4 consecutive channels of rows (lane/8 + 8s): 4 x 16-byte accesses per block and
        // stream, all element-wise work and the per-channel partial sums in that layout.
        float* scr = smem + wave * (32 * LDA);
        const int erow = lane >> 3, ec4 = (lane & 7) * 4;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // the three switches are compile-time inside the body (straight-line code per block; as run-time tests the
        // compiler re-branched on them for every 16-byte access)
        auto epilogue = [&](auto ACT, auto ADD, auto BNF, auto STATS) {
            constexpr bool act1 = decltype(ACT)::value, has_add = decltype(ADD)::value, bnf = decltype(BNF)::value;
            constexpr bool stats = decltype(STATS)::value;  // per-channel partial sums wanted (p.stat_part)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                cs4[j] = zero4;
                cq4[j] = zero4;
                const int col = n0 + wn * 32 * WN + j * 32 + ec4;
                const f32x4 bv4 = (p.bias != nullptr) ? *reinterpret_cast<const f32x4*>(p.bias + col) : zero4;
                f32x4 mu4 = zero4, is4 = zero4;
                if constexpr (bnf) {
                    mu4 = *reinterpret_cast<const f32x4*>(p.bn_mean + col);
                    is4 = *reinterpret_cast<const f32x4*>(p.bn_invstd + col);
                }
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const size_t e0 = (size_t)(m0 + wm * 32 * WM + i * 32 + erow) * p.ldd + col;
                    const size_t estep = (size_t)8 * p.ldd;
                    // two half-blocks (rows erow + 0/8, then + 16/24): the half's loads first - they overlap the LDS
                    // round trip / the previous half's arithmetic - then the arithmetic and the stores
                    f32x4 ad[2][2], zz[2][2], yy[2][2];
                    auto fetch = [&](int h) {  // h is a compile-time constant at every call site
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const size_t e = e0 + (2 * h + q) * estep;
                            if constexpr (has_add) ad[h][q] = *reinterpret_cast<const f32x4*>(p.addend + e);
                            if constexpr (bnf) {
                                zz[h][q] = *reinterpret_cast<const f32x4*>(p.relu_src + e);
                                yy[h][q] = *reinterpret_cast<const f32x4*>(p.bn_y + e);
                            }
                        }
                    };
                    fetch(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDA + l31] = acc[i][j][r];
                    fetch(1);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int rq = 2 * h + q;
                            f32x4 o = *reinterpret_cast<const f32x4*>(&scr[(rq * 8 + erow) * LDA + ec4]) + bv4;
                            if constexpr (has_add) o += ad[h][q];  // act(conv + bias + addend)
                            if constexpr (act1) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                            }
                            if constexpr (bnf) {  // fused BN-backward front end (see IgemmNT)
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] = (zz[h][q][e] > 0.f) ? o[e] : 0.f;
                                cq4[j] += o * ((yy[h][q] - mu4) * is4);
                            } else if constexpr (stats) {
                                cq4[j] += o * o;
                            }
                            if constexpr (stats) cs4[j] += o;
                            *reinterpret_cast<f32x4*>(p.dst + e0 + rq * estep) = o;
                        }
                    }
                    // one block at a time: without the fence the scheduler hoists every block's loads to the top of
                    // the epilogue (3 streams x 16 registers x WM*WN blocks) and spills the accumulators
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        using T = std::true_type;
        using F = std::false_type;
        const bool has_add = p.addend != nullptr;
        if (p.relu_src != nullptr) {
            // (never launched on the 2x2-block-per-wave tile, see nt_decide: its 64 accumulators + three operand
            // streams do not fit the register budget of 3 workgroups / CU)
            if constexpr (WM * WN < 4) {
                if (has_add) epilogue(F{}, T{}, T{}, T{}); else epilogue(F{}, F{}, T{}, T{});
            }
        } else if (p.act == 1) {  // (statistics are taken of pre-activation outputs: never together with ReLU)
            if (has_add) epilogue(T{}, T{}, F{}, F{}); else epilogue(T{}, F{}, F{}, F{});
        } else if (p.stat_part != nullptr) {
            if (has_add) epilogue(F{}, T{}, F{}, T{}); else epilogue(F{}, F{}, F{}, T{});
        } else {
            if (has_add) epilogue(F{}, T{}, F{}, F{}); else epilogue(F{}, F{}, F{}, F{});
        }
#if defined(VSPW_NT_TIMING) && VSPW_NT_TIMING == 2
        NT_STAMP(1);  // experiment: slot 1 = every store of the tile issued (not drained)
#endif
    } else {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            csum[j] = 0.f;
            csq[j] = 0.f;
            const int col = n0 + wn * 32 * WN + j * 32 + l31;
            const bool cok = col < p.nout;
            const float bv = (p.bias != nullptr && cok) ? p.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (cok && row < p.m) {
                        float v = acc[i][j][r] + bv;
                        if (p.addend != nullptr) v += p.addend[(size_t)row * p.ldd + col];
                        v = nt_act(v, p.act);  // act(conv + bias + addend)
                        float w2 = v;
                        if (p.relu_src != nullptr) {  // fused BN-backward front end (see IgemmNT)
                            if (!(p.relu_src[(size_t)row * p.ldd + col] > 0.f)) v = 0.f;
                            w2 = (p.bn_y[(size_t)row * p.ldd + col] - p.bn_mean[col]) * p.bn_invstd[col];
                        }
                        p.dst[(size_t)row * p.ldd + col] = v;
                        csum[j] += v;
                        csq[j] += v * w2;
                    }
                }
            }
        }
    }
    if (p.stat_part != nullptr) {
        float* red = As[0];  // [2 stats][WGM][TN]  (TM*LDA >= 2*WGM*TN for every instantiated tile)
        __syncthreads();
        if (interior) {
            // a lane holds 4 channels' sums over the rows lane/8 + 8s of its wave's blocks: fold the 8 row groups
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                f32x4 s4 = cs4[j], q4 = cq4[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = s4[e], b = q4[e];
                    a += __shfl_xor(a, 8, 64);
                    b += __shfl_xor(b, 8, 64);
                    a += __shfl_xor(a, 16, 64);
                    b += __shfl_xor(b, 16, 64);
                    a += __shfl_xor(a, 32, 64);
                    b += __shfl_xor(b, 32, 64);
                    s4[e] = a;
                    q4[e] = b;
                }
                if (lane < 8) {
                    const int c = wn * 32 * WN + j * 32 + lane * 4;
                    *reinterpret_cast<f32x4*>(&red[(0 * WGM + wm) * TN + c]) = s4;
                    *reinterpret_cast<f32x4*>(&red[(1 * WGM + wm) * TN + c]) = q4;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                float s = csum[j] + __shfl_xor(csum[j], 32, 64);
                float q = csq[j] + __shfl_xor(csq[j], 32, 64);
                if (lh == 0) {
                    int c = wn * 32 * WN + j * 32 + l31;
                    red[(0 * WGM + wm) * TN + c] = s;
                    red[(1 * WGM + wm) * TN + c] = q;
                }
            }
        }
        __syncthreads();
        if (tid < TN) {
            int col = n0 + tid;
            if (col < p.nout) {
                float* out = p.stat_part + (size_t)tile_m * 2 * p.nout;
                float s = red[tid], q = red[WGM * TN + tid];
                if (WGM == 2) {
                    s += red[TN + tid];
                    q += red[3 * TN + tid];
                }
                out[col] = s;
                out[p.nout + col] = q;
            }
        }
    }
#ifdef VSPW_NT_TIMING
    __builtin_amdgcn_s_waitcnt(0);  // the diagnostic stamp counts the stores as drained; production waves just end
    NT_STAMP(3);
    if (threadIdx.x == 0 && stamp_slot < 8192) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        vspw_nt_stamps[stamp_slot * 5 + 4] = ((unsigned long long)xcc << 32) | id;
    }
#endif
}

// Sustained shader clock under the real load (bench.py's `roofline.sustained_clock_ghz`): workgroup 0 of every v2 NT launch
// reads s_memtime (shader cycles) and the 100 MHz wall clock when it starts and when it ends and adds both differences to
// two device counters - four scalar reads and two atomics per LAUNCH.  Why it is worth having: a launch costs a constant
// number of CYCLES, and the clock it gets is the power management's decision - measured on the 36 000 x 256 x 1024 GEMM
// (tools/diag/nt_clock{,_idle}.py, profiles/r05_nt_clock.log): 329-354 k cycles every time, at 1.87 ... 2.40 GHz depending on
// operand values (zero-filled: 2.34 GHz, random: 1.9) and on what ran in the milliseconds before = 109 ... 131 TFLOP/s from
// the same code.  The 157.3 TFLOP/s peak is quoted at 2.4 GHz.
// OFF unless armed (vspw_debug_nt_clock_enable): an unarmed launch pays one scalar load in one wave.  One start stamp per
// device: arm it only while the NT launches of ONE stream run one after the other (bench.py's timed region does; launches
// that overlap on several streams would mix their stamps).
__device__ unsigned long long vspw_nt_clock_probe[4];  // [0] ticks start, [1] wall start (scratch); [2] sum ticks, [3] sum wall
__device__ int vspw_nt_clock_armed = 0;
template <int WGM, int WM, int WN, int MODE, int NBUF, int TAPS = 0, int AFF = 0, int CHUNK = 0, int FOLD = 0>
__global__ __launch_bounds__(256, NT_V2_BOUNDS(WM, WN, MODE, NBUF, TAPS, AFF)) void igemm_nt_v2_kernel(IgemmNT pin) {
    const bool probe = (blockIdx.x | blockIdx.y | threadIdx.x) == 0 && vspw_nt_clock_armed != 0;
    if (probe) {
        vspw_nt_clock_probe[0] = __builtin_readcyclecounter();
        vspw_nt_clock_probe[1] = wall_clock64();
    }
    igemm_nt_v2_body<WGM, WM, WN, MODE, NBUF, TAPS, AFF, CHUNK, FOLD>(pin, xcd_remap(blockIdx.x, gridDim.x), blockIdx.y,
                                                                      blockIdx.y * gridDim.x + blockIdx.x);
    if (probe) {
        const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
        atomicAdd(&vspw_nt_clock_probe[2], t1 - vspw_nt_clock_probe[0]);
        atomicAdd(&vspw_nt_clock_probe[3], w1 - vspw_nt_clock_probe[1]);
    }
}

// out[0] = shader cycles, out[1] = 100 MHz wall ticks accumulated by the probe above since the last reset (synchronises the
// device: diagnostics / bench.py only); reset != 0 zeroes the sums afterwards.
extern "C" int vspw_debug_nt_clock(unsigned long long* out, int reset) {
    unsigned long long h[4];
    if (hipDeviceSynchronize() != hipSuccess) return VSPW_ELAUNCH;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(vspw_nt_clock_probe), sizeof(h)) != hipSuccess) return VSPW_ELAUNCH;
    if (out) { out[0] = h[2]; out[1] = h[3]; }
    if (reset) {
        h[2] = h[3] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(vspw_nt_clock_probe), h, sizeof(h)) != hipSuccess) return VSPW_ELAUNCH;
    }
    return VSPW_OK;
}

// Arm (on != 0) / disarm the probe above.  Synchronises the device.
extern "C" int vspw_debug_nt_clock_enable(int on) {
    const int v = on ? 1 : 0;
    if (hipDeviceSynchronize() != hipSuccess) return VSPW_ELAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vspw_nt_clock_armed), &v, sizeof(v)) != hipSuccess) return VSPW_ELAUNCH;
    return VSPW_OK;
}

#ifdef VSPW_NT_TIMING
extern "C" int vspw_debug_nt_stamps(unsigned long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vspw_nt_stamps), sizeof(unsigned long long) * n);
}
#endif

// Tile choice.  fp32 MFMA work is uniform per output element, so apart from a small per-tile efficiency difference
// (bigger tiles amortise barriers and fragment loads better) the scheduling loss is the tail: with workgroups handed
// out greedily, the busiest of the 256 CUs processes ceil(workgroups / 256) tiles.  Pick the tile height that
// minimises  ceil(wg/256) * rows / efficiency  among 128 (2x2 waves of 2x2 MFMA tiles), 96 (1x4 waves of 3x1) and
// 64 (2x2 waves of 1x2).  Codes: 22 / 31 / 12; narrow outputs (Cout <= 64, the stem) use 128x64 (21) or 64x64 (11).
static int nt_pick_tile(long long m, int nout, int batch = 1) {
    if (batch < 1) batch = 1;
    if (nout <= 64) return (((m + 127) / 128) * batch >= 1024) ? 21 : 11;
    const long long tn = (long long)((nout + 127) / 128) * batch;  // batched GEMMs: the batch multiplies the grid
    // small problems (the RAFT update block: 12 840 pixels): with 64x128 tiles there would be at most two workgroups
    // per CU - one wave per SIMD, nothing to hide LDS / barrier latency behind.  64x64 tiles quadruple the number of
    // workgroups; the lost operand reuse does not matter at sizes that live in L2.
    // measured on the RAFT forward (B = 2, 480x856): 27.3 -> 24.3 ms with the threshold at two workgroups per CU
    static const int small_wg = getenv("VSPW_SMALL_WG") ? atoi(getenv("VSPW_SMALL_WG")) : 520;
    if (((m + 63) / 64) * tn <= small_wg) return 11;
    const int rows[3] = {128, 96, 64};
    const int code[3] = {22, 31, 12};
    const double eff[3] = {1.0, 0.97, 0.92};
    int best = 0;
    double best_cost = 1e300;
    for (int i = 0; i < 3; ++i) {
        const long long wg = ((m + rows[i] - 1) / rows[i]) * tn;
        const double cost = (double)((wg + 255) / 256) * rows[i] / eff[i];
        if (cost < best_cost * 0.999) {
            best_cost = cost;
            best = i;
        }
    }
    return code[best];
}

static int nt_tile_rows(int cfg) { return (cfg == 22 || cfg == 21) ? 128 : (cfg == 31 ? 96 : 64); }

static int nt_tap_inner() {
    static const int v = getenv("VSPW_TAP_INNER") ? atoi(getenv("VSPW_TAP_INNER")) : 1;
    return v;
}
// Folded chains of the direct 3x3 kernels (igemm_nt_v2_body, FOLD): on unless VSPW_DIRECT_FOLD=0 (A/B runs,
// tools/diag/parity_attrib.sh)
static int nt_direct_fold() {
    static const int v = getenv("VSPW_DIRECT_FOLD") ? atoi(getenv("VSPW_DIRECT_FOLD")) : 1;
    return v;
}
// A direct (non-Winograd) 3x3 launch of the v2 kernel that folds its chains: the small-channel convolutions of the stem /
// layer1 / the strided ones (c <= 128 - wider 3x3s reach the direct kernel only with Winograd switched off).  They run
// on the 64x64 tile: 16 + 16 accumulators at four workgroups per CU, no spills - measured (tools/diag/direct3x3_time.py)
// as fast as the unfolded 128x64 / 128x128 tiles on the stem (402 + 361 vs 388 + 358 us, 666 + 648 vs 625 + 635 us) and
// FASTER on layer1 / layer2.0 (88 vs 97 us, 99 vs 113 us), where the larger tiles left CUs idle.
static bool nt_folds(const IgemmNT& p) {
    return p.kh == 3 && p.kw == 3 && p.c <= 128 && nt_tap_inner() && nt_direct_fold();
}

// Process-wide summation policy of the K >= 2*chunk pointwise GEMMs (see igemm_nt_v2_body, "two-level accumulation"):
// chunk length in k (multiple of 32); 0 = one k-sequential chain = the default (measured, profiles/r05_parity_attrib_*:
// chains of 256 halve the rounding error of a K >= 1024 GEMM but move the end-to-end parity figures by < 0.1x while
// costing 2.8 ms per step; the excess sat in the direct 3x3 kernels, see FOLD).  VSPW_ACCUM_CHUNK / vspw_set_accum_chunk.
static int g_accum_chunk = -1;
// The variants are compiled only with -DVSPW_WITH_ACCUM_CHUNK (tools/diag/build_variant.py; 29-45 SGPR spills each, +2.8 ms
// per step when on): the shipped library accepts 0 only.
static int accum_chunk() {
#ifdef VSPW_WITH_ACCUM_CHUNK
    if (g_accum_chunk < 0) g_accum_chunk = getenv("VSPW_ACCUM_CHUNK") ? atoi(getenv("VSPW_ACCUM_CHUNK")) / BK * BK : 0;
    return g_accum_chunk;
#else
    return 0;
#endif
}
extern "C" int vspw_set_accum_chunk(int k) {
    if (k < 0 || k % BK != 0) return VSPW_EINVAL;
#ifndef VSPW_WITH_ACCUM_CHUNK
    if (k != 0) return VSPW_EINVAL;  // not compiled in
#endif
    g_accum_chunk = k;
    return VSPW_OK;
}
extern "C" int vspw_get_accum_chunk(void) { return accum_chunk(); }
extern "C" int vspw_accum_chunk_compiled(void) {
#ifdef VSPW_WITH_ACCUM_CHUNK
    return 1;
#else
    return 0;
#endif
}

// pointwise (MODE 2) launch of one A-operand flavour (AFF 0-3), with / without two-level accumulation
template <int A, int CH>
static void launch_nt_pw(const IgemmNT& p, int cfg, hipStream_t st) {
    if (cfg == 22) {
        int tiles = vspw_cdiv(p.m, 128) * vspw_cdiv(p.nout, 128);
        // measured: short reductions (K <= 1024, i.e. the 1x1 convs) gain ~10 % from the higher residency of the
        // single-buffer variant (3 workgroups/CU); long ones gain 2-5 % from the second buffer (one barrier per tile)
        static const int nbuf1_max_k = getenv("VSPW_NBUF1_MAXK") ? atoi(getenv("VSPW_NBUF1_MAXK")) : 1024;
        if constexpr (A == 0) {
            if (p.kdim > nbuf1_max_k) {
                hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 2, 2, 2, 2, 0, 0, CH>), dim3(tiles, p.batch), dim3(256), 0, st, p);
                return;
            }
        }
        hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 2, 2, 2, 1, 0, A, CH>), dim3(tiles, p.batch), dim3(256), 0, st, p);
    } else if (cfg == 31) {
        int tiles = vspw_cdiv(p.m, 96) * vspw_cdiv(p.nout, 128);
        hipLaunchKernelGGL((igemm_nt_v2_kernel<1, 3, 1, 2, 1, 0, A, CH>), dim3(tiles, p.batch), dim3(256), 0, st, p);
    } else if (cfg == 12) {
        int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 128);
        hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 1, 2, 2, 1, 0, A, CH>), dim3(tiles, p.batch), dim3(256), 0, st, p);
    } else if (cfg == 21) {
        int tiles = vspw_cdiv(p.m, 128) * vspw_cdiv(p.nout, 64);
        hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 2, 1, 2, 1, 0, A, CH>), dim3(tiles, p.batch), dim3(256), 0, st, p);
    } else {
        int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 64);
        hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 1, 1, 2, 1, 0, A, CH>), dim3(tiles, p.batch), dim3(256), 0, st, p);
    }
}

template <int MODE>
static void launch_nt_v2(const IgemmNT& p, int cfg, hipStream_t st) {
    if constexpr (MODE == 2) {
        if (p.wino_d > 0) {  // Winograd input operand: 16 batches folded into grid.x (fastest), 96- or 64-row tiles
            if (cfg == 12) {
                int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 128);
                hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 1, 2, 2, 1, 0, 4>), dim3(tiles * 16), dim3(256), 0, st, p);
            } else {
                int tiles = vspw_cdiv(p.m, 96) * vspw_cdiv(p.nout, 128);
                hipLaunchKernelGGL((igemm_nt_v2_kernel<1, 3, 1, 2, 1, 0, 4>), dim3(tiles * 16), dim3(256), 0, st, p);
            }
            return;
        }
        // A operand: 2 = fused forward apply of the producing node (A operand + z), 3 = ... of a node without a residual
        // branch, 1 = affine (fused BatchNorm-backward apply), 0 = plain
        const int aff = (p.src2 != nullptr && p.zout != nullptr) ? 2 : (p.zout != nullptr ? 3 : (p.src2 != nullptr ? 1 : 0));
#ifdef VSPW_WITH_ACCUM_CHUNK  // two-level accumulation variants: diagnostic builds only (tools/diag/build_variant.py)
        if (p.chunk_tiles > 0) {
            switch (aff) {
                case 0: launch_nt_pw<0, 1>(p, cfg, st); break;
                case 1: launch_nt_pw<1, 1>(p, cfg, st); break;
                case 2: launch_nt_pw<2, 1>(p, cfg, st); break;
                default: launch_nt_pw<3, 1>(p, cfg, st); break;
            }
            return;
        }
#endif
        switch (aff) {
            case 0: launch_nt_pw<0, 0>(p, cfg, st); break;
            case 1: launch_nt_pw<1, 0>(p, cfg, st); break;
            case 2: launch_nt_pw<2, 0>(p, cfg, st); break;
            default: launch_nt_pw<3, 0>(p, cfg, st); break;
        }
        return;
    }
    if constexpr (MODE != 2) {
        if (nt_tap_inner() && p.kh == 3 && p.kw == 3) {  // 3x3: channel-slab-outer / tap-inner K order
            if (nt_folds(p)) {  // folded chains on the 64x64 tile (nt_decide has set cfg = 11)
                int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 64);
                hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 1, 1, MODE, 1, 9, 0, 0, 3>), dim3(tiles, p.batch), dim3(256), 0, st, p);
                return;
            }
            // (fallback forms - stride-1 3x3s with >= 128 channels take Winograd, narrower ones the folded tile above: the
            // 128x64 tile and the data gradient's 128x128 tile, whose tap state spilled 28-34 SGPRs, run on 64x64 / 96x128)
            // (nt_decide never hands 21, nor 22 to the data gradient, to this branch)
            if (cfg == 22 && MODE == 0) {
                int tiles = vspw_cdiv(p.m, 128) * vspw_cdiv(p.nout, 128);
                hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 2, 2, 0, 1, 9>), dim3(tiles, p.batch), dim3(256), 0, st, p);
            } else if (cfg == 31) {
                int tiles = vspw_cdiv(p.m, 96) * vspw_cdiv(p.nout, 128);
                hipLaunchKernelGGL((igemm_nt_v2_kernel<1, 3, 1, MODE, 1, 9>), dim3(tiles, p.batch), dim3(256), 0, st, p);
            } else if (cfg == 12) {
                int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 128);
                hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 1, 2, MODE, 1, 9>), dim3(tiles, p.batch), dim3(256), 0, st, p);
            } else {
                int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 64);
                hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 1, 1, MODE, 1, 9>), dim3(tiles, p.batch), dim3(256), 0, st, p);
            }
            return;
        }
        // other filter shapes (tap-outer K order)
        if (cfg == 22) {
            int tiles = vspw_cdiv(p.m, 128) * vspw_cdiv(p.nout, 128);
            static const int nbuf1_max_k = getenv("VSPW_NBUF1_MAXK") ? atoi(getenv("VSPW_NBUF1_MAXK")) : 1024;
            if (p.kdim <= nbuf1_max_k)
                hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 2, 2, MODE, 1>), dim3(tiles, p.batch), dim3(256), 0, st, p);
            else
                hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 2, 2, MODE, 2>), dim3(tiles, p.batch), dim3(256), 0, st, p);
        } else if (cfg == 31) {
            int tiles = vspw_cdiv(p.m, 96) * vspw_cdiv(p.nout, 128);
            hipLaunchKernelGGL((igemm_nt_v2_kernel<1, 3, 1, MODE, 1>), dim3(tiles, p.batch), dim3(256), 0, st, p);
        } else if (cfg == 12) {
            int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 128);
            hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 1, 2, MODE, 1>), dim3(tiles, p.batch), dim3(256), 0, st, p);
        } else if (cfg == 21) {
            int tiles = vspw_cdiv(p.m, 128) * vspw_cdiv(p.nout, 64);
            hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 2, 1, MODE, 1>), dim3(tiles, p.batch), dim3(256), 0, st, p);
        } else {
            int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 64);
            hipLaunchKernelGGL((igemm_nt_v2_kernel<2, 1, 1, MODE, 1>), dim3(tiles, p.batch), dim3(256), 0, st, p);
        }
    }
}

// Final (tile code, use-v2) decision for a launch; shared by the launcher and vspw_conv2d_stats_partials so that the
// number of per-tile BN partial rows the caller allocates always matches the kernel that runs.
// measured (profiles/r01_*): the straight-line v2 pipeline wins 5-12 % everywhere; the 128x128 tile uses two LDS
// buffers, the smaller tiles one (two would cap residency at 2 workgroups/CU and lose)
static int nt_decide(const IgemmNT& p, bool& v2) {
    int cfg = nt_pick_tile(p.m, p.nout, p.batch);
    static const int force_cfg = getenv("VSPW_NT_CFG") ? atoi(getenv("VSPW_NT_CFG")) : 0;  // experiments: 22 / 31 / 12 / 21 / 11
    if (force_cfg) cfg = force_cfg;
    // the fused BatchNorm-backward front end makes the epilogue a long memory phase (three extra operand streams): the
    // 96-row tile (48 accumulators, one more resident workgroup to overlap it with) beats 128x128 there (-10 %)
    if (p.relu_src != nullptr && cfg == 22) cfg = 31;
    // 32-bit byte offsets relative to the first image a tile touches / the tile's first weight row
    const long long img_elems = (long long)p.h * p.w * p.lds;
    const long long span = (128 / ((long long)p.oh * p.ow) + 2) * img_elems;
    // byte offsets (incl. the scalar channel / k base) must stay below the 2 GiB out-of-range marker of the buffer loads
    v2 = p.vec && p.c % BK == 0 && span < (1LL << 28) && (long long)p.kdim < (1LL << 21) &&
         true;  // forward (any stride) and data gradient (any stride: inexact taps are masked per tap)
    if (!v2 && (cfg == 31 || cfg == 22)) cfg = 12;  // the generic kernel has no 96-row instantiation (and its 128x128 one spilled)
    if (v2 && nt_folds(p)) cfg = 11;  // direct 3x3 with folded chains: see nt_folds
    // fallback direct 3x3 forms of the v2 kernel (tap-inner K order, no folded chains): the 128x64 tile and the data
    // gradient's 128x128 tile kept their tap state in 28-34 spilled SGPRs - they run on 64x64 / 96x128 instead.  (The
    // statistics-partials counts follow this function: one place decides.)
    if (v2 && nt_tap_inner() && p.kh == 3 && p.kw == 3) {
        if (cfg == 21) cfg = 11;
        if (cfg == 22 && p.mode != 0) cfg = 31;
    }
    return cfg;
}

static int launch_igemm_nt(const IgemmNT& pin, hipStream_t st) {
    bool v2;
    IgemmNT p = pin;
    const int cfg = nt_decide(p, v2);
    if (v2) {
        const bool pw = p.kh * p.kw == 1 && p.stride == 1 && p.pad == 0 && p.padw == 0;
        p.chunk_tiles = 0;
        if (pw && p.wino_d == 0 && accum_chunk() >= BK && p.kdim >= 2 * accum_chunk()) p.chunk_tiles = accum_chunk() / BK;
        if (pw)
            launch_nt_v2<2>(p, cfg, st);  // forward and data gradient of a pointwise conv are the same plain GEMM
        else if (p.mode == 0)
            launch_nt_v2<0>(p, cfg, st);
        else
            launch_nt_v2<1>(p, cfg, st);
        return vspw_launch_status();
    }
    if (cfg == 12) {
        int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 128);
        hipLaunchKernelGGL((igemm_nt_kernel<1, 2>), dim3(tiles, p.batch), dim3(256), 0, st, p);
    } else if (cfg == 21) {
        int tiles = vspw_cdiv(p.m, 128) * vspw_cdiv(p.nout, 64);
        hipLaunchKernelGGL((igemm_nt_kernel<2, 1>), dim3(tiles, p.batch), dim3(256), 0, st, p);
    } else {
        int tiles = vspw_cdiv(p.m, 64) * vspw_cdiv(p.nout, 64);
        hipLaunchKernelGGL((igemm_nt_kernel<1, 1>), dim3(tiles, p.batch), dim3(256), 0, st, p);
    }
    return vspw_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[co][tap*C + ci] = sum_p dY[p][co] * X[src(p, tap)][ci]   (TN GEMM, split-K)
// ---------------------------------------------------------------------------------------------
struct IgemmTN {
    const float* dy;  // [P][k]
    const float* x;   // [nb][h][w][c]
    float* part;      // [splits][k][ncols]
    int nb, h, w, c;
    int oh, ow;
    int kh, kw, stride, pad, padw, dil;
    int k;       // Cout (GEMM M)
    int ncols;   // kh*kw*c (GEMM N)
    int P;       // nb*oh*ow (GEMM K)
    int chunk;   // pixels per split (multiple of BK)
    int vec_a;   // k % 4 == 0
    int vec_b;   // c % 4 == 0
    // gather mode 5 (pointwise + affine dY): dY element = coef[0][co]*dy + coef[1][co]*dy2 + coef[2][co]  (see IgemmNT)
    const float* dy2;
    const float* coef;
    // batched plain GEMM (vspw_bmm_tn): blockIdx.z = batch index, element strides of dy / x / part between batches
    int batch;
    long long bs_dy, bs_x, bs_part;
};

__global__ __launch_bounds__(256) void igemm_tn_kernel(IgemmTN pin) {
    IgemmTN p = pin;
    if (pin.batch > 1) {
        p.dy += (size_t)blockIdx.z * pin.bs_dy;
        p.x += (size_t)blockIdx.z * pin.bs_x;
        p.part += (size_t)blockIdx.z * pin.bs_part;
    }
    __shared__ __attribute__((aligned(16))) float As[BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tiles_n = (p.ncols + BN - 1) / BN;
    const int tile_n = blockIdx.x % tiles_n;
    const int tile_m = blockIdx.x / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int p_begin = split * p.chunk;
    const int p_end = min(p.P, p_begin + p.chunk);

    const int krow = tid >> 5;        // 0..7 (+8*i)
    const int c4 = (tid & 31) * 4;    // 0..124

    // B column decode (k independent).
    int b_ky[4], b_kx[4], b_ci[4];
    bool b_colok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int n = n0 + c4 + e;
        b_colok[e] = n < p.ncols;
        int nn = b_colok[e] ? n : 0;
        int tap = nn / p.c;
        b_ci[e] = nn - tap * p.c;
        b_ky[e] = tap / p.kw;
        b_kx[e] = tap - b_ky[e] * p.kw;
    }
    const int ohw = p.oh * p.ow;

    f32x4 ra[4], rb[4];
    auto load_tile = [&](int pk0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pp = pk0 + krow + 8 * i;
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (pp < p_end) {
                const int co = m0 + c4;
                if (p.vec_a) {
                    if (co < p.k) av = *reinterpret_cast<const f32x4*>(p.dy + (size_t)pp * p.k + co);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.k) av[e] = p.dy[(size_t)pp * p.k + co + e];
                }
                const int n = pp / ohw;
                const int r = pp - n * ohw;
                const int oy = r / p.ow;
                const int ox = r - oy * p.ow;
                const int by = oy * p.stride - p.pad, bx = ox * p.stride - p.padw;
                if (p.vec_b) {
                    if (b_colok[0]) {
                        int sy = by + b_ky[0] * p.dil, sx = bx + b_kx[0] * p.dil;
                        if (sy >= 0 && sy < p.h && sx >= 0 && sx < p.w)
                            bv = *reinterpret_cast<const f32x4*>(
                                p.x + (((size_t)n * p.h + sy) * p.w + sx) * (size_t)p.c + b_ci[0]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (b_colok[e]) {
                            int sy = by + b_ky[e] * p.dil, sx = bx + b_kx[e] * p.dil;
                            if (sy >= 0 && sy < p.h && sx >= 0 && sx < p.w)
                                bv[e] = p.x[(((size_t)n * p.h + sy) * p.w + sx) * (size_t)p.c + b_ci[e]];
                        }
                    }
                }
            }
            ra[i] = av;
            rb[i] = bv;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p_end - p_begin + BK - 1) / BK;
    if (nk > 0) load_tile(p_begin);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&As[(krow + 8 * i) * BM + c4]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[(krow + 8 * i) * BN + c4]) = rb[i];
        }
        __syncthreads();
        if (kt + 1 < nk) load_tile(p_begin + (kt + 1) * BK);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[(2 * s + lh) * BM + wm * 64 + i * 32 + l31];
                b[i] = Bs[(2 * s + lh) * BN + wn * 64 + i * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    float* out = p.part + (size_t)split * p.k * p.ncols;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        if (col >= p.ncols) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < p.k) out[(size_t)row * p.ncols + col] = acc[i][j][r];
            }
        }
    }
}

// v2 of the TN kernel for the vector path (Cout % 4 == 0, Cin % 4 == 0, 32-bit offsets): same pipeline as
// igemm_nt_v2_kernel - straight-line loop body, always-valid addresses (zeroing applied when the registers are written
// to LDS), the next tile's global loads issued in the shadow of the MFMAs.
// Gather mode G (how the 32 pixels of a K-tile are turned into addresses of x):
//   0  generic, two integer divisions per staged row per K-tile (narrow feature maps, ow < 32)
//   1  generic, incremental (image, y, x) update (ow >= 32)
//   2  linear: stride 1 and same-size output (oh == h, ow == w: every 3x3 conv of layers 1-4 but the three strided
//      ones), so the source pixel of (output pixel pp, tap) is pp + tap offset - the per-tile part of every address
//      is a workgroup-uniform base bump (scalar registers) and the per-thread part a constant 32-bit offset; only the
//      in-image test of the tap still needs the (y, x) of the row, updated incrementally (ow >= 32)
//   3  pointwise (1x1, stride 1, no padding): linear and always in-image, no pixel coordinates at all
// VALU work in this loop is not hidden behind the fp32 MFMAs (tools/probe/mfma_ablate.hip), hence the modes.
#ifndef TN_NBUF
#define TN_NBUF 1
#endif
// WM / WN = MFMA 32x32 blocks per wave along Cout / along (tap, Cin); 2x2 waves -> tile (64*WM) x (64*WN).
// Narrow weight matrices (Cout <= 64 or KH*KW*Cin <= 64: the stem, layer1, 1x1 convs to/from 64 channels) use
// WM = 1 / WN = 1 so that no half of the MFMA tile is spent on padding.
template <int G, int NBUF, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_tn_v2_kernel(IgemmTN pin) {
    IgemmTN p = pin;
    if (pin.batch > 1) {
        p.dy += (size_t)blockIdx.z * pin.bs_dy;
        p.x += (size_t)blockIdx.z * pin.bs_x;
        p.part += (size_t)blockIdx.z * pin.bs_part;
    }
    TN_STAMP(0);
    NT_PRIO(NT_PRIO_EDGE);  // prologue / epilogue at raised priority (see NT_PRIO)
    constexpr bool LIN = G >= 2;
    constexpr bool WIDE = G >= 1;
    constexpr bool UNI = G == 4;  // linear + tap uniform per workgroup + rows uniform per half-wave: scalar validity
    constexpr bool TAPS = G == 2 || G == 4;
    constexpr bool PWG = G == 3 || G == 5;  // pointwise: no pixel coordinates at all
    constexpr bool AFF = G == 5;            // pointwise + affine dY operand (every pixel chunk is a multiple of BK)
    constexpr int TM = 64 * WM, TN = 64 * WN;
    constexpr int A4 = TM / 4, B4 = TN / 4;          // float4 per staged row
    constexpr int RPA = 256 / A4, RPB = 256 / B4;    // rows covered per pass
    constexpr int PA = BK / RPA, PB = BK / RPB;      // passes (float4 per thread) = 2 or 4
    __shared__ __attribute__((aligned(16))) float As[NBUF][BK * TM];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][BK * TN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware order: workgroup b runs on XCD b % 8; give each XCD a contiguous range of (split, tile) pairs so that
    // all tiles of one pixel chunk - which read the same dY and X rows - share an L2 (measured before: 7x the
    // algorithmic bytes fetched from the fabric, each chunk being pulled into all eight L2s)
    const int tiles_n = (p.ncols + TN - 1) / TN;
    const int ntiles = gridDim.x;
    const int vb = xcd_remap(blockIdx.y * ntiles + blockIdx.x, ntiles * gridDim.y);
    const int tile = vb % ntiles;
    const int tile_n = tile % tiles_n;
    const int tile_m = tile / tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const int split = vb / ntiles;
    const int p_begin = split * p.chunk;
    const int p_end = min(p.P, p_begin + p.chunk);

    const int krow_a = tid / A4, ca4 = (tid % A4) * 4;
    const int krow_b = tid / B4, cb4 = (tid % B4) * 4;
    // UNI (TN = 128: one staged row = one half-wave): wave v stages rows 8v .. 8v+7, pass i rows 8v+2i (lanes 0-31) and
    // 8v+2i+1 (lanes 32-63) - the pixel of a staged row is then wave-uniform and its in-image test runs on the SCALAR
    // unit (the K loop keeps one v_cndmask per pass on the vector ALU)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto brow = [&](int i) { return UNI ? wave_u * 8 + 2 * i + lh : krow_b + RPB * i; };

    // this thread's A column (co) and B column (tap, ci): fixed for the whole kernel.  The linear modes clamp them to
    // the last valid float4 instead of masking: rows / columns of dW past the matrix are computed and never stored.
    const int co = LIN ? min(m0 + ca4, p.k - 4) : m0 + ca4;
    const bool a_colok = co < p.k;
    const int ncol = LIN ? min(n0 + cb4, p.ncols - 4) : n0 + cb4;
    const bool b_colok = ncol < p.ncols;
    int b_ci, b_dy, b_dx;
    {
        const int nn = b_colok ? ncol : 0;
        const int tap = nn / p.c;
        b_ci = nn - tap * p.c;
        const int ky = tap / p.kw;
        const int kx = tap - ky * p.kw;
        b_dy = ky * p.dil - p.pad;
        b_dx = kx * p.dil - p.padw;
    }
    const int ohw = p.oh * p.ow;

    // pixel state of the PB rows this thread stages for B: (image, oy, ox) advance by BK pixels per K-tile.  When the
    // row is at least BK wide the update is a compare/select chain; otherwise two integer divisions.  A only needs
    // the linear pixel index.
    int pk0 = p_begin;
    int r_n[PB], r_oy[PB], r_ox[PB];
    int su_oy = 0, su_ox = 0, su_dy = 0, su_dx = 0;  // UNI: (oy, ox) of pixel pk0 + 8*wave, tap displacement (scalars)
    if (UNI) {
        const int tap = __builtin_amdgcn_readfirstlane(n0 / p.c);
        const int ky = tap / p.kw;
        su_dy = ky * p.dil - p.pad;
        su_dx = (tap - ky * p.kw) * p.dil - p.padw;
        const int r = (p_begin + wave_u * 8) % ohw;
        su_oy = r / p.ow;
        su_ox = r - su_oy * p.ow;
    }
    if (!PWG && !UNI) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int pp = p_begin + krow_b + RPB * i;
            r_n[i] = pp / ohw;
            const int r = pp - r_n[i] * ohw;
            r_oy[i] = r / p.ow;
            r_ox[i] = r - r_oy[i] * p.ow;
        }
    }
    // linear modes: uniform bases (bumped per K-tile) + constant per-thread byte offsets.  xb points (pad rows + pad
    // pixels) before the tile's first pixel so that every tap offset is non-negative; it is only dereferenced with
    // offsets that land inside the tensor.
    const char* dyb = reinterpret_cast<const char*>(p.dy) + (size_t)p_begin * p.k * 4;
    const char* dyb2 = reinterpret_cast<const char*>(AFF ? p.dy2 : p.dy) + (size_t)p_begin * p.k * 4;
    f32x4 cfa = {1.f, 1.f, 1.f, 1.f}, cfb = {0.f, 0.f, 0.f, 0.f}, cfc = cfb;  // this thread's dY columns never change
    if (AFF) {
        cfa = *reinterpret_cast<const f32x4*>(p.coef + co);
        cfb = *reinterpret_cast<const f32x4*>(p.coef + p.k + co);
        cfc = *reinterpret_cast<const f32x4*>(p.coef + 2 * p.k + co);
    }
    const char* xb = reinterpret_cast<const char*>(p.x) + ((long long)p_begin - p.pad * p.w - p.padw) * p.c * 4;
    unsigned a_voff[PA], b_voff[PB];
    // pixels from xb to the last staged pixel's farthest tap, minus the staged pixels themselves: 2*pad rows + 2*padw
    // pixels (every tap offset lies in [0, 2*(pad*w + padw)] pixels)
    const int tap_px = 2 * (p.pad * p.w + p.padw);
    if (LIN) {
#pragma unroll
        for (int i = 0; i < PA; ++i) a_voff[i] = (unsigned)((krow_a + RPA * i) * p.k + co) * 4u;
        const int tapoff = ((b_dy + p.pad) * p.w + (b_dx + p.padw)) * p.c + b_ci;
#pragma unroll
        for (int i = 0; i < PB; ++i) b_voff[i] = (unsigned)(brow(i) * p.c + tapoff) * 4u;
    }
    auto advance = [&]() {
        pk0 += BK;
        if (LIN) {
            dyb += (size_t)BK * p.k * 4;
            xb += (size_t)BK * p.c * 4;
        }
        if (AFF) dyb2 += (size_t)BK * p.k * 4;
        if (PWG) return;
        if (UNI) {  // scalar: ow >= BK, at most one row wrap per step
            su_ox += BK;
            if (su_ox >= p.ow) {
                su_ox -= p.ow;
                su_oy = (su_oy + 1 >= p.oh) ? 0 : su_oy + 1;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if (WIDE) {  // ow >= BK: at most one row wrap per step
                r_ox[i] += BK;
                const bool wx = r_ox[i] >= p.ow;
                r_ox[i] = wx ? r_ox[i] - p.ow : r_ox[i];
                r_oy[i] = wx ? r_oy[i] + 1 : r_oy[i];
                const bool wy = r_oy[i] >= p.oh;
                r_oy[i] = wy ? 0 : r_oy[i];
                if (!LIN) r_n[i] = wy ? r_n[i] + 1 : r_n[i];
            } else {
                const int pp = pk0 + krow_b + RPB * i;
                r_n[i] = pp / ohw;
                const int r = pp - r_n[i] * ohw;
                r_oy[i] = r / p.ow;
                r_ox[i] = r - r_oy[i] * p.ow;
            }
        }
    };
    f32x4 ra[PA], rb[PB], ra2[AFF ? PA : 1];
    bool oka[PA], okb[PB];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned TN_OOR = 0x80000000u;
    auto load_tile = [&]() {
        if (LIN) {
            // Raw buffer loads: the descriptors are rebuilt per K-tile from scalars (base = first pixel of the tile,
            // num_records = bytes left in this workgroup's pixel chunk), so rows past the chunk are out of range and
            // come back as zeros from the hardware bounds check; an out-of-image 3x3 tap is the out-of-range offset
            // TN_OOR.  No select touches the loaded data and no address arithmetic runs on the vector ALU - which
            // matters because fp32 MFMA executes on the SIMD's f32 vector lanes: VALU instructions in this loop are
            // subtracted from MFMA issue time (the select form spent 45 / 89 VALU instructions per 64 MFMAs).
            const int rows_left = p_end - pk0;  // uniform
            if (rows_left <= 0) return;         // prefetch past the chunk: nothing to fetch, registers never consumed
            const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(dyb), 0, rows_left * p.k * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < PA; ++i)
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[i], 0, 0));
            if (AFF) {
                const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(dyb2), 0, rows_left * p.k * 4, 0x00020000);
#pragma unroll
                for (int i = 0; i < PA; ++i)
                    ra2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a2, a_voff[i], 0, 0));
            }
            // x: the tile's rows start (pad rows + pad pixels) before its first pixel (xb), so the chunk's last pixel
            // with the largest tap offset ends tap_px pixels after rows_left pixels
            // (never past the end of x: the last chunk's rows beyond P are out of range and read as zeros; an interior
            // chunk's rows beyond rows_left are masked below - a uniform branch taken only in a chunk's last K-tile)
            const int x_left_px = p.P - (pk0 - tap_px / 2);  // pixels from xb to the end of x (scalar)
            const int want_px = rows_left + tap_px;
            const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(xb), 0, (want_px < x_left_px ? want_px : x_left_px) * p.c * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                unsigned off = b_voff[i];
                if (UNI) {
                    unsigned kill[2];  // 0 or TN_OOR per half-wave row, computed on the scalar unit
#pragma unroll
                    for (int e = 0; e < 2; ++e) {  // rows 8*wave + 2i + e
                        const int j = 2 * i + e;
                        int x = su_ox + j, y = su_oy;
                        if (x >= p.ow) {
                            x -= p.ow;
                            y = (y + 1 >= p.oh) ? 0 : y + 1;
                        }
                        const bool ok = ((unsigned)(y + su_dy) < (unsigned)p.h) & ((unsigned)(x + su_dx) < (unsigned)p.w) &
                                        (wave_u * 8 + j < rows_left);
                        kill[e] = ok ? 0u : TN_OOR;
                    }
                    off |= lh ? kill[1] : kill[0];  // valid offsets are < 2 GiB: the top bit makes them out of range
                } else {
                    if (G == 2) {
                        const int sy = r_oy[i] + b_dy, sx = r_ox[i] + b_dx;
                        off = (((unsigned)sy < (unsigned)p.h) & ((unsigned)sx < (unsigned)p.w)) ? off : TN_OOR;
                        // interior chunks: rows past rows_left would read real pixels (times a zero dY row)
                        off = (krow_b + RPB * i < rows_left) ? off : TN_OOR;
                    }
                }
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, off, 0, 0));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int pp = pk0 + krow_a + RPA * i;
            oka[i] = (pp < p_end) & a_colok;
            ra[i] = *reinterpret_cast<const f32x4*>(p.dy + (oka[i] ? pp * p.k + co : 0));
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const bool pin = (pk0 + krow_b + RPB * i) < p_end;
            const int sy = r_oy[i] * p.stride + b_dy, sx = r_ox[i] * p.stride + b_dx;
            okb[i] = pin & b_colok & ((unsigned)sy < (unsigned)p.h) & ((unsigned)sx < (unsigned)p.w);
            rb[i] = *reinterpret_cast<const f32x4*>(p.x + (okb[i] ? ((r_n[i] * p.h + sy) * p.w + sx) * p.c + b_ci : 0));
        }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto store_tile = [&](float* Ad, float* Bd) {
        if (AFF) {
#pragma unroll
            for (int i = 0; i < PA; ++i) ra[i] = cfa * ra[i] + (cfb * ra2[i] + cfc);
        }
#pragma unroll
        for (int i = 0; i < PA; ++i)
            *reinterpret_cast<f32x4*>(&Ad[(krow_a + RPA * i) * TM + ca4]) = (LIN || oka[i]) ? ra[i] : zero4;
#pragma unroll
        for (int i = 0; i < PB; ++i)
            *reinterpret_cast<f32x4*>(&Bd[brow(i) * TN + cb4]) = (LIN || okb[i]) ? rb[i] : zero4;
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p_end - p_begin + BK - 1) / BK;
    if (nk > 0) {
        load_tile();
        if (NBUF == 2) {
            store_tile(As[0], Bs[0]);
            advance();
            load_tile();  // rows past p_end are zeroed at store time
            __syncthreads();
        }
    }
    TN_STAMP(1);
    NT_PRIO(0);
    for (int kt = 0; kt < nk; ++kt) {
        NT_PROGRESS_PRIO(kt, nk);
        const int cur = (NBUF == 2) ? (kt & 1) : 0;
        if (NBUF == 1) {
            store_tile(As[0], Bs[0]);
            __syncthreads();
            advance();
        }
        const float* Ac = As[cur];
        const float* Bc = Bs[cur];
        // Fragments.  A wave's MFMA block i covers rows 2*l31 + i (WM == 2) rather than i*32 + l31: the two values a
        // lane needs for one k are then 8 contiguous bytes, and two k-steps (LDS rows 2 apart) come from ONE
        // ds_read2st64_b64 - 16 LDS reads per 64 MFMAs instead of 32, a wait every other MFMA group.  Same for B / WN.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int sp = 0; sp < BK / 4; ++sp) {
            float a0[WM], a1[WM], b0[WN], b1[WN];
            if (WM == 2) {
                const f32x2 t0 = *reinterpret_cast<const f32x2*>(&Ac[(4 * sp + lh) * TM + wm * 64 + 2 * l31]);
                const f32x2 t1 = *reinterpret_cast<const f32x2*>(&Ac[(4 * sp + 2 + lh) * TM + wm * 64 + 2 * l31]);
                a0[0] = t0[0]; a0[WM - 1] = t0[1]; a1[0] = t1[0]; a1[WM - 1] = t1[1];
            } else {
                a0[0] = Ac[(4 * sp + lh) * TM + wm * 32 + l31];
                a1[0] = Ac[(4 * sp + 2 + lh) * TM + wm * 32 + l31];
            }
            if (WN == 2) {
                const f32x2 t0 = *reinterpret_cast<const f32x2*>(&Bc[(4 * sp + lh) * TN + wn * 64 + 2 * l31]);
                const f32x2 t1 = *reinterpret_cast<const f32x2*>(&Bc[(4 * sp + 2 + lh) * TN + wn * 64 + 2 * l31]);
                b0[0] = t0[0]; b0[WN - 1] = t0[1]; b1[0] = t1[0]; b1[WN - 1] = t1[1];
            } else {
                b0[0] = Bc[(4 * sp + lh) * TN + wn * 32 + l31];
                b1[0] = Bc[(4 * sp + 2 + lh) * TN + wn * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
            if (sp == 1) {
                if (NBUF == 2) {
                    store_tile(As[(NBUF - 1) & (cur ^ 1)], Bs[(NBUF - 1) & (cur ^ 1)]);  // tile kt+1
                    advance();
                }
                load_tile();  // NBUF 2: tile kt+2 (a whole iteration to land); NBUF 1: tile kt+1
            }
        }
        __syncthreads();
    }

    TN_STAMP(2);
    NT_PRIO(NT_PRIO_EDGE);
    float* out = p.part + (size_t)split * p.k * p.ncols;
    // (row / column of accumulator element r of block (i, j) under the interleaved fragment mapping above)
    if (WN == 2 && (p.ncols & 1) == 0) {
        // blocks j = 0, 1 of a lane are columns 2*l31 and 2*l31 + 1: one 8-byte store per (i, r) - half the store
        // instructions of the element-wise form, and every one fills whole 32-byte sectors
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const int col = n0 + wn * 64 + 2 * l31;
        if (col < p.ncols) {
#pragma unroll
            for (int i = 0; i < WM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int row = m0 + wm * 32 * WM + (WM == 2 ? 2 * rr + i : rr);
                    const f32x2 v = {acc[i][0][r], acc[i][WN - 1][r]};
                    if (row < p.k) *reinterpret_cast<f32x2*>(&out[(size_t)row * p.ncols + col]) = v;
                }
            }
        }
#ifdef VSPW_NT_TIMING
        __builtin_amdgcn_s_waitcnt(0);
        TN_STAMP(3);
        if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 8192) {
            unsigned id, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            vspw_nt_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 5 + 4] = ((unsigned long long)xcc << 32) | id;
        }
#endif
        return;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn * 32 * WN + (WN == 2 ? 2 * l31 + j : l31);
        if (col >= p.ncols) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int row = m0 + wm * 32 * WM + (WM == 2 ? 2 * rr + i : rr);
                if (row < p.k) out[(size_t)row * p.ncols + col] = acc[i][j][r];
            }
        }
    }
}

// dW = sum over split-K partial slabs (fixed order: deterministic).  float4 lanes, 4 slabs in flight per step.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            long long n, int splits) {
    part += (size_t)blockIdx.y * splits * n;  // batched GEMMs: one slab group and one output per blockIdx.y
    out += (size_t)blockIdx.y * n;
    const long long n4 = (n & 3) ? 0 : (n >> 2);  // slabs are 16-B aligned only when n % 4 == 0
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
    for (; i < n4; i += stride) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 4 <= splits; z += 4) {
            const f32x4 a = p4[(size_t)z * n4 + i];
            const f32x4 b = p4[(size_t)(z + 1) * n4 + i];
            const f32x4 c = p4[(size_t)(z + 2) * n4 + i];
            const f32x4 d = p4[(size_t)(z + 3) * n4 + i];
            s += a;
            s += b;
            s += c;
            s += d;
        }
        for (; z < splits; ++z) s += p4[(size_t)z * n4 + i];
        reinterpret_cast<f32x4*>(out)[i] = s;
    }
    // scalar path: everything when n % 4 != 0, nothing otherwise
    for (long long j = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += part[(size_t)z * n + j];
        out[j] = s;
    }
}

// [r][t][s] -> [s][t][r]   (weights [Cout][taps][Cin] -> [Cin][taps][Cout])
__global__ void weight_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int T, int S) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int r0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int r = r0 + i, s = s0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && s < S) ? in[((size_t)r * T + t) * S + s] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int s = s0 + i, r = r0 + threadIdx.x;
        if (r < R && s < S) out[((size_t)s * T + t) * R + r] = tile[threadIdx.x][i];
    }
}

// The same transpose for MANY weight tensors in one launch (every convolution's data gradient needs the
// [Cin][taps][Cout] copy of its weights once per step, after the optimizer rewrote them): entries[] (device) is sorted
// by tile0 = number of 32x32 tiles of all preceding tensors; a workgroup finds its tensor by binary search.
__global__ __launch_bounds__(256) void weight_transpose_multi_kernel(const vspw_wt_entry* __restrict__ entries,
                                                                     int n_entries) {
    __shared__ float tile[32][33];
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
    const int R = e.k, T = e.taps, S = e.c;
    const int ts = (S + 31) / 32, tr = (R + 31) / 32;
    int local = (int)(b - e.tile0);
    const int sx = local % ts;
    local /= ts;
    const int ry = local % tr;
    const int t = local / tr;
    const int r0 = ry * 32, s0 = sx * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ in = e.w;
    float* __restrict__ out = e.wT;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, sc = s0 + tx;
        tile[i][tx] = (r < R && sc < S) ? in[((size_t)r * T + t) * S + sc] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int sc = s0 + i, r = r0 + tx;
        if (r < R && sc < S) out[((size_t)sc * T + t) * R + r] = tile[tx][i];
    }
}

extern "C" long long vspw_weight_transpose_tiles(int k, int taps, int c) {
    if (k <= 0 || taps <= 0 || c <= 0) return 0;
    return (long long)taps * ((k + 31) / 32) * ((c + 31) / 32);
}

extern "C" int vspw_weight_transpose_multi(const vspw_wt_entry* entries, int n_entries, long long total_tiles,
                                           void* stream) {
    if (!entries || n_entries <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffLL) return VSPW_EINVAL;
    hipLaunchKernelGGL(weight_transpose_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, vspw_stream(stream),
                       entries, n_entries);
    return vspw_launch_status();
}

// NCHW -> NHWC for the images crossing the boundary.  One thread per pixel for small C (the 3-channel frames): plane
// reads are coalesced across threads, the C outputs of a thread are contiguous; grid.y = image (no integer division).
template <int C>
__global__ __launch_bounds__(256) void nchw_to_nhwc_small_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 long long HW) {
    const long long n = blockIdx.y;
    const float* src = in + n * C * HW;
    float* dst = out + n * C * HW;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += stride) {
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = src[c * HW + p];
#pragma unroll
        for (int c = 0; c < C; ++c) dst[p * C + c] = v[c];
    }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, long long HW) {
    long long total = (long long)N * C * HW;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        int c = (int)(i % C);
        long long r = i / C;
        long long hw = r % HW;
        long long n = r / HW;
        out[i] = in[((size_t)n * C + c) * HW + hw];
    }
}

static int conv_geometry_ok(const vspw_conv_desc* d) {
    if (!d) return 0;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->k <= 0) return 0;
    if (d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->dil <= 0 || d->pad < 0 || d->pad_w < 0) return 0;
    int oh = (d->h + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
    int ow = (d->w + 2 * d->pad_w - d->dil * (d->kw - 1) - 1) / d->stride + 1;
    return oh == d->oh && ow == d->ow && oh > 0 && ow > 0;
}

static bool fill_fwd_params(const vspw_conv_desc* d, IgemmNT& p) {
    p.src = nullptr; p.wt = nullptr; p.bias = nullptr; p.dst = nullptr; p.stat_part = nullptr; p.addend = nullptr;
    p.relu_src = nullptr; p.bn_y = nullptr; p.bn_mean = nullptr; p.bn_invstd = nullptr;
    p.src2 = nullptr; p.coef = nullptr; p.zout = nullptr;
    p.batch = 1; p.bs_src = p.bs_wt = p.bs_dst = 0;
    p.wino_d = p.wino_th = p.wino_tw = 0;
#ifdef VSPW_NT_DBG
    p.dbg = getenv("VSPW_NT_DBG") ? atoi(getenv("VSPW_NT_DBG")) : 0;
#endif
    p.nb = d->n; p.h = d->h; p.w = d->w; p.c = d->c;
    p.oh = d->oh; p.ow = d->ow;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad; p.padw = d->pad_w; p.dil = d->dil;
    p.mode = 0;
    p.nout = d->k; p.ldd = d->k;
    long long m = (long long)d->n * d->oh * d->ow;
    if (m > 0x7fffffffLL) return false;
    p.m = (int)m;
    p.kdim = d->kh * d->kw * d->c;
    p.vec = (d->c % 4 == 0) ? 1 : 0;
    p.lds = d->c;
    p.act = 0;
    return true;
}

extern "C" size_t vspw_conv2d_stats_partials(const vspw_conv_desc* d) {
    if (!conv_geometry_ok(d)) return 0;
    IgemmNT p;
    if (!fill_fwd_params(d, p)) return 0;
    bool v2;
    const int rows = nt_tile_rows(nt_decide(p, v2));
    return (size_t)((p.m + rows - 1) / rows);
}

// Few-row pointwise GEMM (the pyramid-pool branches: 2 ... 72 pooled pixels x 2048 -> 512 channels, models/models.py:
// 905-912 / clip_psp.py:45-56): through the MFMA tiles these are 8 workgroups walking 64 K-tiles each - 58-62 us of pure
// latency per branch (profiles/r05_final_gemm_shapes.csv).  Here a WAVE owns one output channel: its weight row lives in
// registers (lane = 4 consecutive k, +256 per pass), the input rows stream through L2, one wave_sum per output element.
#define SKINNY_MAX_ROWS 128
#define SKINNY_ROWS_PER_WG 24
template <int KP>  // KP = K / 256 passes
__global__ __launch_bounds__(256) void skinny_pointwise_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int m, int n, int k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 4 + wave;
    if (col >= n) return;
    f32x4 wr[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) wr[j] = *reinterpret_cast<const f32x4*>(w + (size_t)col * k + j * 256 + lane * 4);
    const float bv = bias ? bias[col] : 0.f;
    const int r0 = blockIdx.y * SKINNY_ROWS_PER_WG, r1 = min(m, r0 + SKINNY_ROWS_PER_WG);
    for (int r = r0; r < r1; ++r) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KP; ++j) acc += wr[j] * *reinterpret_cast<const f32x4*>(x + (size_t)r * k + j * 256 + lane * 4);
        const float s = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
        if (lane == 0) y[(size_t)r * n + col] = s + bv;
    }
}

// y [m][nout] = x [m][kred] . w [nout][kred]^T (+ bias): true when the launch was issued (few rows, a reduction of 2-16 x 256)
static bool launch_skinny(const vspw_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int nout,
                          int kred, hipStream_t st) {
    static const int enabled = getenv("VSPW_SKINNY") ? atoi(getenv("VSPW_SKINNY")) : 1;
    const long long m = (long long)d->n * d->oh * d->ow;
    if (!enabled || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->pad_w != 0 || m > SKINNY_MAX_ROWS ||
        kred % 256 != 0 || kred < 512 || kred > 4096 || nout < 64)
        return false;
    const dim3 grid(vspw_cdiv(nout, 4), vspw_cdiv(m, SKINNY_ROWS_PER_WG));
#define SKINNY(KP) hipLaunchKernelGGL((skinny_pointwise_kernel<KP>), grid, dim3(256), 0, st, x, w, bias, y, (int)m, nout, kred)
    switch (kred / 256) {
        case 2: SKINNY(2); break;
        case 3: SKINNY(3); break;
        case 4: SKINNY(4); break;
        case 6: SKINNY(6); break;
        case 8: SKINNY(8); break;
        case 12: SKINNY(12); break;
        case 16: SKINNY(16); break;
        default: return false;
    }
#undef SKINNY
    return true;
}

extern "C" int vspw_conv2d_fwd(const vspw_conv_desc* d, const float* x, const float* w, const float* bias,
                               float* y, float* stat_part, void* stream) {
    if (!conv_geometry_ok(d) || !x || !w || !y) return VSPW_EINVAL;
    if (stat_part == nullptr && launch_skinny(d, x, w, bias, y, d->k, d->c, vspw_stream(stream))) return vspw_launch_status();
    IgemmNT p;
    if (!fill_fwd_params(d, p)) return VSPW_EINVAL;
    p.src = x; p.wt = w; p.bias = bias; p.dst = y; p.stat_part = stat_part;
    return launch_igemm_nt(p, vspw_stream(stream));
}

extern "C" size_t vspw_conv2d_fwd_apply_supported(const vspw_conv_desc* d) {
    IgemmNT p;
    if (!conv_geometry_ok(d) || !fill_fwd_params(d, p)) return 0;
    bool v2;
    nt_decide(p, v2);
    return (v2 && p.kh * p.kw == 1 && p.stride == 1 && p.pad == 0 && p.padw == 0) ? 1 : 0;
}

extern "C" int vspw_conv2d_fwd_apply(const vspw_conv_desc* d, const float* y_in, const float* res_in,
                                     const float* scale_shift, float* z_out, const float* w, const float* bias, float* y,
                                     float* stat_part, void* stream) {
    if (!conv_geometry_ok(d) || !y_in || !scale_shift || !z_out || !w || !y) return VSPW_EINVAL;
    IgemmNT p;
    if (!fill_fwd_params(d, p)) return VSPW_EINVAL;
    bool v2;
    nt_decide(p, v2);
    const bool pw = p.kh * p.kw == 1 && p.stride == 1 && p.pad == 0 && p.padw == 0;
    if (!v2 || !pw) return VSPW_EINVAL;
    p.src = y_in; p.src2 = res_in; p.coef = scale_shift; p.zout = z_out;
    p.wt = w; p.bias = bias; p.dst = y; p.stat_part = stat_part;
    return launch_igemm_nt(p, vspw_stream(stream));
}

extern "C" int vspw_conv2d_fwd_ex(const vspw_conv_desc* d, const float* x, long long ldx, const float* w,
                                  const float* bias, const float* addend, int act, float* y, long long ldy,
                                  void* stream) {
    if (!conv_geometry_ok(d) || !x || !w || !y || ldx < d->c || ldy < d->k || act < 0 || act > 3) return VSPW_EINVAL;
    if (ldx > 0x7fffffffLL || ldy > 0x7fffffffLL) return VSPW_EINVAL;
    IgemmNT p;
    if (!fill_fwd_params(d, p)) return VSPW_EINVAL;
    p.src = x; p.wt = w; p.bias = bias; p.dst = y; p.stat_part = nullptr;
    p.lds = (int)ldx;
    p.ldd = (int)ldy;
    p.act = act;
    p.addend = addend;  // [m][ldy], same layout as y
    p.vec = (d->c % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0) ? 1 : 0;
    return launch_igemm_nt(p, vspw_stream(stream));
}

struct BnFront {
    const float* relu_src;
    const float* bn_y;
    const float* bn_mean;
    const float* bn_invstd;
    float* stat_part;
};
struct AffA {
    const float* y;
    const float* coef;
};
static int conv2d_bwd_data_impl(const vspw_conv_desc* d, const float* dy, const float* wT, const float* addend,
                                float* dx, void* stream, const BnFront* bn = nullptr, const AffA* aff = nullptr);
static bool fill_bwd_data_params(const vspw_conv_desc* d, IgemmNT& p);

extern "C" int vspw_conv2d_bwd_data(const vspw_conv_desc* d, const float* dy, const float* wT, float* dx,
                                    void* stream) {
    return conv2d_bwd_data_impl(d, dy, wT, nullptr, dx, stream);
}

extern "C" int vspw_conv2d_bwd_data_acc(const vspw_conv_desc* d, const float* dy, const float* wT,
                                        const float* addend, float* dx, void* stream) {
    if (!addend) return VSPW_EINVAL;
    return conv2d_bwd_data_impl(d, dy, wT, addend, dx, stream);
}

extern "C" size_t vspw_conv2d_bwd_data_bn_partials(const vspw_conv_desc* d) {
    // rows of the per-tile partial buffer of vspw_conv2d_bwd_data_bn; 0 = this geometry cannot take the fused path
    if (!conv_geometry_ok(d)) return 0;
    IgemmNT p;
    if (!fill_bwd_data_params(d, p)) return 0;
    p.relu_src = reinterpret_cast<const float*>(16);  // only tested against nullptr by the tile choice
    bool v2;
    const int rows = nt_tile_rows(nt_decide(p, v2));
    return v2 ? (size_t)((p.m + rows - 1) / rows) : 0;
}

extern "C" int vspw_conv2d_bwd_data_bn(const vspw_conv_desc* d, const float* dy, const float* wT, const float* addend,
                                       const float* relu_src, const float* bn_y, const float* bn_mean,
                                       const float* bn_invstd, float* dx, float* stat_part, void* stream) {
    if (!relu_src || !bn_y || !bn_mean || !bn_invstd || !stat_part) return VSPW_EINVAL;
    if (vspw_conv2d_bwd_data_bn_partials(d) == 0) return VSPW_EINVAL;
    BnFront bn = {relu_src, bn_y, bn_mean, bn_invstd, stat_part};
    return conv2d_bwd_data_impl(d, dy, wT, addend, dx, stream, &bn);
}

extern "C" int vspw_conv2d_bwd_data_aff(const vspw_conv_desc* d, const float* g, const float* y, const float* coef,
                                        const float* wT, const float* addend, const float* relu_src, const float* bn_y,
                                        const float* bn_mean, const float* bn_invstd, float* dx, float* stat_part,
                                        void* stream) {
    if (!g || !y || !coef) return VSPW_EINVAL;
    AffA aff = {y, coef};
    if (relu_src || bn_y || bn_mean || bn_invstd || stat_part) {
        if (!relu_src || !bn_y || !bn_mean || !bn_invstd || !stat_part) return VSPW_EINVAL;
        if (vspw_conv2d_bwd_data_bn_partials(d) == 0) return VSPW_EINVAL;
        BnFront bn = {relu_src, bn_y, bn_mean, bn_invstd, stat_part};
        return conv2d_bwd_data_impl(d, g, wT, addend, dx, stream, &bn, &aff);
    }
    return conv2d_bwd_data_impl(d, g, wT, addend, dx, stream, nullptr, &aff);
}

static bool fill_bwd_data_params(const vspw_conv_desc* d, IgemmNT& p) {
    p.src = nullptr; p.wt = nullptr; p.bias = nullptr; p.dst = nullptr; p.stat_part = nullptr; p.addend = nullptr;
    p.relu_src = nullptr; p.bn_y = nullptr; p.bn_mean = nullptr; p.bn_invstd = nullptr;
    p.src2 = nullptr; p.coef = nullptr; p.zout = nullptr;
    p.batch = 1; p.bs_src = p.bs_wt = p.bs_dst = 0;
    p.wino_d = p.wino_th = p.wino_tw = 0;
#ifdef VSPW_NT_DBG
    p.dbg = getenv("VSPW_NT_DBG") ? atoi(getenv("VSPW_NT_DBG")) : 0;
#endif
    p.nb = d->n; p.h = d->oh; p.w = d->ow; p.c = d->k;   // gather over dY
    p.oh = d->h; p.ow = d->w;                              // rows are input pixels
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad; p.padw = d->pad_w; p.dil = d->dil;
    p.mode = 1;
    p.nout = d->c; p.ldd = d->c;
    long long m = (long long)d->n * d->h * d->w;
    if (m > 0x7fffffffLL) return false;
    p.m = (int)m;
    p.kdim = d->kh * d->kw * d->k;
    p.vec = (d->k % 4 == 0) ? 1 : 0;
    p.lds = d->k;
    p.act = 0;
    return true;
}

static int conv2d_bwd_data_impl(const vspw_conv_desc* d, const float* dy, const float* wT, const float* addend,
                                float* dx, void* stream, const BnFront* bn, const AffA* aff) {
    if (!conv_geometry_ok(d) || !dy || !wT || !dx) return VSPW_EINVAL;
    // (few rows: dx [m][Cin] = dy [m][Cout] . wT [Cin][Cout]^T is the same skinny product as the forward pass)
    if (!addend && !bn && !aff && launch_skinny(d, dy, wT, nullptr, dx, d->c, d->k, vspw_stream(stream)))
        return vspw_launch_status();
    IgemmNT p;
    if (!fill_bwd_data_params(d, p)) return VSPW_EINVAL;
    p.src = dy; p.wt = wT; p.dst = dx; p.addend = addend;
    if (aff) {
        bool v2;
        nt_decide(p, v2);
        const bool pw = p.kh * p.kw == 1 && p.stride == 1 && p.pad == 0 && p.padw == 0;
        if (!v2 || !pw || !aff->y || !aff->coef) return VSPW_EINVAL;
        p.src2 = aff->y;
        p.coef = aff->coef;
    }
    if (bn) {
        p.relu_src = bn->relu_src; p.bn_y = bn->bn_y; p.bn_mean = bn->bn_mean; p.bn_invstd = bn->bn_invstd;
        p.stat_part = bn->stat_part;
    }
    return launch_igemm_nt(p, vspw_stream(stream));
}

// tile of the weight-gradient GEMM: 64 wide on a side whose extent is <= 64 (only on the v2 path)
static void wgrad_tile(const vspw_conv_desc* d, int& tm, int& tn) {
    const bool vec = (d->k % 4 == 0) && (d->c % 4 == 0);
    tm = (vec && d->k <= 64) ? 64 : BM;
    tn = (vec && d->kh * d->kw * d->c <= 64) ? 64 : BN;
}

// Split-K plan of the weight gradient: how many pixel chunks (each one workgroup per output tile, partial slabs summed
// by splitk_reduce_kernel in split order).  Workgroups are equal-sized, so the launch takes ceil(N / 256) "rounds" of
// one workgroup per CU where N / 256 would do: the old rule (N ~ 768, whatever the tile count) left 864 workgroups for
// the 1024->512 3x3 (3.4 rounds of work in 4: 114 TFLOP/s where the forward GEMM of the same shape reaches 141) and
// 770 for the stem (3.008 in 4).  Cost model per candidate s (relative to the perfectly divisible GEMM):
//     rounds(N) / (N / 256)            quantisation, N = tiles * splits
//   x 1 / occupancy(N / 256)           1 / 2 / >= 3 resident workgroups per CU hide barrier and load latency differently
//   x (1 + 6 / k_tiles)                per-workgroup prologue + epilogue, about six K-tiles' worth
//   + 100 * s / P                      partial slabs written and re-read (2 x 4 bytes per output element and split
//                                      against 2 * P flops per output element at ~125 TFLOP/s and ~4 TB/s)
static void wgrad_plan(const vspw_conv_desc* d, int& splits, int& chunk, int batch = 1) {
    const long long P = (long long)d->n * d->oh * d->ow;
    const int ncols = d->kh * d->kw * d->c;
    int tm, tn;
    wgrad_tile(d, tm, tn);
    const long long tiles = (long long)vspw_cdiv(d->k, tm) * vspw_cdiv(ncols, tn) * batch;
    long long max_splits = (P + 255) / 256;
    if (max_splits > 512) max_splits = 512;
    if (max_splits < 1) max_splits = 1;
    static const int forced = getenv("VSPW_WGRAD_SPLITS") ? atoi(getenv("VSPW_WGRAD_SPLITS")) : 0;  // diagnostic sweep
    if (forced > 0) {
        long long ch = (P + forced - 1) / forced;
        ch = ((ch + BK - 1) / BK) * BK;
        splits = (int)((P + ch - 1) / ch);
        chunk = (int)ch;
        return;
    }
    double best = 1e30;
    long long best_ch = ((P + BK - 1) / BK) * BK;
    int best_s = 1;
    for (long long want = 1; want <= max_splits; ++want) {
        long long ch = (P + want - 1) / want;
        ch = ((ch + BK - 1) / BK) * BK;
        const long long sp = (P + ch - 1) / ch;
        if (sp != want && want != 1) continue;  // this chunk size was already seen under a smaller `want`
        const double n = (double)(tiles * sp);
        const double per_cu = n / 256.0;
        const double rounds = (double)((tiles * sp + 255) / 256);
        const double occ = per_cu >= 3.0 ? 1.0 : (per_cu >= 2.0 ? 0.93 : 0.80);
        const double kt = (double)(ch / BK);
        const double cost = rounds / per_cu / occ * (1.0 + 6.0 / kt) + 100.0 * (double)sp / (double)P;
        if (cost < best) {
            best = cost;
            best_ch = ch;
            best_s = (int)sp;
        }
    }
    splits = best_s;
    chunk = (int)best_ch;
}

// Few-channel inputs (the RGB stem conv, Cin = 3): the vector gather of the v2 kernel needs Cin % 4 == 0, and the
// generic kernel's element-wise gather ran this memory-bound GEMM (147 MB of dY against 2 GFLOP) at 4.6 TFLOP/s,
// 0.43 ms per step.  Instead: copy x into a 4-channel-padded image (zeros in the pad), run the v2 kernel on the padded
// geometry, drop the pad columns of dW.  Both helpers are plain streaming kernels; the copies live in the workspace.
static bool wgrad_pads_channels(const vspw_conv_desc* d) { return d->c % 4 != 0 && d->c < 16 && d->k % 4 == 0; }
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ x, float* __restrict__ xp,
                                                           long long npix, int c, int cp) {
    const long long total = npix * cp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long px = i / cp;
        const int ch = (int)(i - px * cp);
        xp[i] = ch < c ? x[px * c + ch] : 0.f;
    }
}

__global__ __launch_bounds__(256) void strip_channels_kernel(const float* __restrict__ wp, float* __restrict__ w,
                                                             long long rows, int c, int cp) {
    const long long total = rows * c;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c;
        w[i] = wp[r * cp + (i - r * c)];
    }
}

extern "C" size_t vspw_conv2d_bwd_weight_workspace(const vspw_conv_desc* d) {
    if (!conv_geometry_ok(d)) return 0;
    if (wgrad_pads_channels(d)) {
        vspw_conv_desc dp = *d;
        dp.c = (d->c + 3) & ~3;
        return align256((size_t)d->n * d->h * d->w * dp.c * sizeof(float)) +
               align256((size_t)d->k * d->kh * d->kw * dp.c * sizeof(float)) + vspw_conv2d_bwd_weight_workspace(&dp);
    }
    int splits, chunk;
    wgrad_plan(d, splits, chunk);
    if (splits <= 1) return 0;
    return (size_t)splits * d->k * d->kh * d->kw * d->c * sizeof(float);
}

struct BatchTN {  // vspw_bmm_tn: `batch` independent GEMMs, element strides between them
    int batch;
    long long bs_dy, bs_x, bs_out;
};
static int conv2d_bwd_weight_impl(const vspw_conv_desc* d, const float* dy, const float* x, float* dw, void* ws,
                                  size_t ws_bytes, void* stream, const AffA* aff, const BatchTN* bt = nullptr);

extern "C" int vspw_conv2d_bwd_weight(const vspw_conv_desc* d, const float* dy, const float* x, float* dw,
                                      void* ws, size_t ws_bytes, void* stream) {
    return conv2d_bwd_weight_impl(d, dy, x, dw, ws, ws_bytes, stream, nullptr);
}

extern "C" int vspw_conv2d_bwd_weight_aff(const vspw_conv_desc* d, const float* g, const float* y, const float* coef,
                                          const float* x, float* dw, void* ws, size_t ws_bytes, void* stream) {
    if (!y || !coef) return VSPW_EINVAL;
    AffA aff = {y, coef};
    return conv2d_bwd_weight_impl(d, g, x, dw, ws, ws_bytes, stream, &aff);
}

static int conv2d_bwd_weight_impl(const vspw_conv_desc* d, const float* dy, const float* x, float* dw, void* ws,
                                  size_t ws_bytes, void* stream, const AffA* aff, const BatchTN* bt) {
    if (!conv_geometry_ok(d) || !dy || !x || !dw) return VSPW_EINVAL;
    const int batch = bt ? bt->batch : 1;
    if (batch < 1 || batch > 65535) return VSPW_EINVAL;
    if (wgrad_pads_channels(d) && aff == nullptr && bt == nullptr) {
        vspw_conv_desc dp = *d;
        dp.c = (d->c + 3) & ~3;
        const size_t xb = align256((size_t)d->n * d->h * d->w * dp.c * sizeof(float));
        const size_t wb = align256((size_t)d->k * d->kh * d->kw * dp.c * sizeof(float));
        if (!ws || ws_bytes < xb + wb) return VSPW_EINVAL;
        float* xp = reinterpret_cast<float*>(ws);
        float* wp = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + xb);
        hipStream_t sp = vspw_stream(stream);
        const long long npix = (long long)d->n * d->h * d->w;
        hipLaunchKernelGGL(pad_channels_kernel, dim3(vspw_stream_grid(npix * dp.c, 256)), dim3(256), 0, sp, x, xp, npix,
                           d->c, dp.c);
        int rc = vspw_launch_status();
        if (rc != VSPW_OK) return rc;
        rc = conv2d_bwd_weight_impl(&dp, dy, xp, wp, reinterpret_cast<char*>(ws) + xb + wb, ws_bytes - xb - wb, stream,
                                    nullptr);
        if (rc != VSPW_OK) return rc;
        const long long rows = (long long)d->k * d->kh * d->kw;
        hipLaunchKernelGGL(strip_channels_kernel, dim3(vspw_stream_grid(rows * d->c, 256)), dim3(256), 0, sp, wp, dw, rows,
                           d->c, dp.c);
        return vspw_launch_status();
    }
    int splits, chunk;
    wgrad_plan(d, splits, chunk, batch);
    const size_t out_elems = (size_t)d->k * d->kh * d->kw * d->c;
    size_t need = splits > 1 ? (size_t)batch * splits * out_elems * sizeof(float) : 0;
    if (need > ws_bytes || (need > 0 && !ws)) return VSPW_EINVAL;
    IgemmTN p;
    p.dy = dy; p.x = x;
    p.part = splits > 1 ? reinterpret_cast<float*>(ws) : dw;
    p.batch = batch;
    p.bs_dy = bt ? bt->bs_dy : 0;
    p.bs_x = bt ? bt->bs_x : 0;
    p.bs_part = splits > 1 ? (long long)splits * (long long)out_elems : (bt ? bt->bs_out : 0);
    p.nb = d->n; p.h = d->h; p.w = d->w; p.c = d->c;
    p.oh = d->oh; p.ow = d->ow;
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad; p.padw = d->pad_w; p.dil = d->dil;
    p.k = d->k;
    p.ncols = d->kh * d->kw * d->c;
    long long P = (long long)d->n * d->oh * d->ow;
    if (P > 0x7fffffffLL) return VSPW_EINVAL;
    p.P = (int)P;
    p.chunk = chunk;
    p.vec_a = (d->k % 4 == 0) ? 1 : 0;
    p.vec_b = (d->c % 4 == 0) ? 1 : 0;
    p.dy2 = aff ? aff->y : nullptr;
    p.coef = aff ? aff->coef : nullptr;
    const bool v2 = p.vec_a && p.vec_b && (long long)p.P * p.k < 0x7fffffffLL &&
                    (long long)d->n * d->h * d->w * d->c < 0x7fffffffLL;
    int tm, tn;
    wgrad_tile(d, tm, tn);
    if (!v2) tm = tn = 128;
    const dim3 grid(vspw_cdiv(p.k, tm) * vspw_cdiv(p.ncols, tn), splits, batch);
    hipStream_t st_ = vspw_stream(stream);
    if (aff) {
        // affine dY: pointwise, vector path, 128-row dY tiles and pixel chunks without a ragged last K-tile only
        const bool point = d->stride == 1 && d->oh == d->h && d->ow == d->w && d->kh * d->kw == 1 && d->pad == 0 &&
                           d->pad_w == 0;
        if (!v2 || !point || tm != 128 || p.P % BK != 0) return VSPW_EINVAL;
        if (tn == 128)
            hipLaunchKernelGGL((igemm_tn_v2_kernel<5, TN_NBUF, 2, 2>), grid, dim3(256), 0, st_, p);
        else
            hipLaunchKernelGGL((igemm_tn_v2_kernel<5, 1, 2, 1>), grid, dim3(256), 0, st_, p);
    } else if (!v2) {
        hipLaunchKernelGGL(igemm_tn_kernel, grid, dim3(256), 0, st_, p);
    } else {
        // gather mode (see igemm_tn_v2_kernel)
        const bool same = d->stride == 1 && d->oh == d->h && d->ow == d->w;
        const bool point = same && d->kh * d->kw == 1 && d->pad == 0 && d->pad_w == 0;
        // 4: linear gather whose 128-column tiles never straddle a filter tap (Cin % 128 == 0): scalar in-image tests
        const int g = point ? 3 : (same && p.ow >= BK) ? ((tn == 128 && d->c % 128 == 0) ? 4 : 2) : (p.ow >= BK ? 1 : 0);
        const int code = g * 100 + (tm / 64) * 10 + (tn / 64);
#define TN_CASE(G, NB, WM_, WN_) \
    case G * 100 + WM_ * 10 + WN_: \
        hipLaunchKernelGGL((igemm_tn_v2_kernel<G, NB, WM_, WN_>), grid, dim3(256), 0, st_, p); \
        break;
        switch (code) {
            TN_CASE(0, 2, 2, 2) TN_CASE(0, 1, 1, 2) TN_CASE(0, 1, 2, 1) TN_CASE(0, 1, 1, 1)
            TN_CASE(1, TN_NBUF, 2, 2) TN_CASE(1, 1, 1, 2) TN_CASE(1, 1, 2, 1) TN_CASE(1, 1, 1, 1)
            TN_CASE(2, TN_NBUF, 2, 2) TN_CASE(2, 1, 1, 2) TN_CASE(2, 1, 2, 1) TN_CASE(2, 1, 1, 1)
            TN_CASE(3, TN_NBUF, 2, 2) TN_CASE(3, 1, 1, 2) TN_CASE(3, 1, 2, 1) TN_CASE(3, 1, 1, 1)
            TN_CASE(4, TN_NBUF, 2, 2) TN_CASE(4, 1, 1, 2)
            default: return VSPW_EINVAL;
        }
#undef TN_CASE
    }
    int st = vspw_launch_status();
    if (st != VSPW_OK) return st;
    if (splits > 1) {
        long long n = (long long)p.k * p.ncols;
        if (bt && bt->bs_out != n) return VSPW_EINVAL;  // (the batched reduce writes densely packed outputs)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(vspw_stream_grid(n / 4 + 1, 256), batch), dim3(256), 0,
                           vspw_stream(stream), reinterpret_cast<const float*>(ws), dw, n, splits);
        st = vspw_launch_status();
    }
    return st;
}

// ---- batched plain GEMMs (OCR object attention / gather, reference models/ocr_modules/spatial_ocr_block.py:100-109,
// 252-274 torch.matmul / torch.bmm on [B, ., .] operands): the batch is a grid dimension of the SAME kernels
static void bmm_nt_desc(vspw_conv_desc& d, int m, int n, int k) {
    d.n = 1; d.h = m; d.w = 1; d.c = k; d.oh = m; d.ow = 1; d.k = n; d.kh = 1; d.kw = 1; d.stride = 1; d.pad = 0;
    d.dil = 1; d.pad_w = 0;
}

extern "C" int vspw_bmm_nt(const float* a, const float* bt, float* c, int batch, int m, int n, int k, void* stream) {
    if (!a || !bt || !c || batch < 1 || batch > 65535 || m < 1 || n < 1 || k < 1) return VSPW_EINVAL;
    vspw_conv_desc d;
    bmm_nt_desc(d, m, n, k);
    IgemmNT p;
    if (!conv_geometry_ok(&d) || !fill_fwd_params(&d, p)) return VSPW_EINVAL;
    p.src = a; p.wt = bt; p.dst = c;
    p.batch = batch;
    p.bs_src = (long long)m * k; p.bs_wt = (long long)n * k; p.bs_dst = (long long)m * n;
    return launch_igemm_nt(p, vspw_stream(stream));
}

static void bmm_tn_desc(vspw_conv_desc& d, int r, int m, int n) {
    d.n = 1; d.h = r; d.w = 1; d.c = n; d.oh = r; d.ow = 1; d.k = m; d.kh = 1; d.kw = 1; d.stride = 1; d.pad = 0;
    d.dil = 1; d.pad_w = 0;
}

extern "C" size_t vspw_bmm_tn_workspace(int batch, int r, int m, int n) {
    if (batch < 1 || r < 1 || m < 1 || n < 1) return 0;
    vspw_conv_desc d;
    bmm_tn_desc(d, r, m, n);
    int splits, chunk;
    wgrad_plan(&d, splits, chunk, batch);
    return splits > 1 ? (size_t)batch * splits * m * n * sizeof(float) : 0;
}

extern "C" int vspw_bmm_tn(const float* a, const float* b, float* c, int batch, int r, int m, int n, void* ws,
                           size_t ws_bytes, void* stream) {
    if (!a || !b || !c || batch < 1 || r < 1 || m < 1 || n < 1) return VSPW_EINVAL;
    vspw_conv_desc d;
    bmm_tn_desc(d, r, m, n);
    BatchTN bt{batch, (long long)r * m, (long long)r * n, (long long)m * n};
    return conv2d_bwd_weight_impl(&d, a, b, c, ws, ws_bytes, stream, nullptr, &bt);
}

extern "C" int vspw_weight_transpose(const float* w, float* wT, int k, int taps, int c, void* stream) {
    if (!w || !wT || k <= 0 || taps <= 0 || c <= 0) return VSPW_EINVAL;
    dim3 grid(vspw_cdiv(c, 32), vspw_cdiv(k, 32), taps);
    hipLaunchKernelGGL(weight_transpose_kernel, grid, dim3(32, 8), 0, vspw_stream(stream), w, wT, k, taps, c);
    return vspw_launch_status();
}

extern "C" int vspw_nchw_to_nhwc(const float* in, float* out, int n, int c, long long hw, void* stream) {
    if (!in || !out || n <= 0 || c <= 0 || hw <= 0) return VSPW_EINVAL;
    long long total = (long long)n * c * hw;
    if (c == 3 && n <= 65535) {
        hipLaunchKernelGGL(nchw_to_nhwc_small_kernel<3>, dim3(vspw_stream_grid(hw, 256), n), dim3(256), 0,
                           vspw_stream(stream), in, out, hw);
        return vspw_launch_status();
    }
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream), in,
                       out, n, c, hw);
    return vspw_launch_status();
}

// 1 when BOTH affine-operand gradient GEMMs (vspw_conv2d_bwd_data_aff, vspw_conv2d_bwd_weight_aff) accept this geometry;
// callers fall back to vspw_bn_bwd_apply + the plain gradients otherwise.
extern "C" size_t vspw_conv2d_bwd_aff_supported(const vspw_conv_desc* d) {
    if (!conv_geometry_ok(d)) return 0;
    const bool point = d->stride == 1 && d->oh == d->h && d->ow == d->w && d->kh * d->kw == 1 && d->pad == 0 &&
                       d->pad_w == 0;
    if (!point) return 0;
    IgemmNT p;
    if (!fill_bwd_data_params(d, p)) return 0;
    bool v2;
    nt_decide(p, v2);
    if (!v2) return 0;
    const long long P = (long long)d->n * d->oh * d->ow;
    const bool tn_v2 = d->k % 4 == 0 && d->c % 4 == 0 && P * d->k < 0x7fffffffLL &&
                       (long long)d->n * d->h * d->w * d->c < 0x7fffffffLL;
    int tm, tn;
    wgrad_tile(d, tm, tn);
    return (tn_v2 && tm == 128 && P % BK == 0) ? 1 : 0;
}

// M[xi] = (B^T d B)[xi] . U[xi]^T for the 16 Winograd positions in one launch, the input transform evaluated while the A
// operand is staged (IgemmNT::wino_d): src = x (forward) or dY (data gradient), NHWC with `channels` channels;
// u [16][rows][channels]; m [16][T][rows].
extern "C" int vspw_wino_gemm_fused_ex(const vspw_conv_desc* d, const float* src, long long ldx, int channels,
                                       const float* u, int rows, float* m, void* stream);
extern "C" int vspw_wino_gemm_fused(const vspw_conv_desc* d, const float* src, int channels, const float* u, int rows,
                                    float* m, void* stream) {
    return vspw_wino_gemm_fused_ex(d, src, channels, channels, u, rows, m, stream);
}

// ... with the source read at pixel stride ldx >= channels (the first `channels` channels of a wider NHWC buffer).
extern "C" int vspw_wino_gemm_fused_ex(const vspw_conv_desc* d, const float* src, long long ldx, int channels,
                                       const float* u, int rows, float* m, void* stream) {
    if (ldx < channels || (ldx & 3) || ldx > 0x7fffffff) return VSPW_EINVAL;
    if (!d || !src || !u || !m || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->dil < 1 || d->pad != d->dil ||
        d->pad_w != d->dil || d->oh != d->h || d->ow != d->w || channels % BK != 0 || rows % 4 != 0)
        return VSPW_EINVAL;
    const int dl = d->dil;
    const int th = ((d->h + dl - 1) / dl + 1) / 2, tw = ((d->w + dl - 1) / dl + 1) / 2;
    const long long tpi = (long long)dl * dl * th * tw, T = tpi * d->n;
    if (T > 0x3fffffffLL) return VSPW_EINVAL;
    vspw_conv_desc g;
    bmm_nt_desc(g, (int)T, rows, channels);
    IgemmNT p;
    if (!conv_geometry_ok(&g) || !fill_fwd_params(&g, p)) return VSPW_EINVAL;
    p.src = src; p.wt = u; p.dst = m;
    p.nb = d->n; p.h = d->h; p.w = d->w; p.oh = (int)tpi; p.ow = 1;
    p.batch = 16;
    p.bs_src = 0; p.bs_wt = (long long)rows * channels; p.bs_dst = T * rows;
    p.wino_d = dl; p.wino_th = th; p.wino_tw = tw;
    p.lds = (int)ldx;
    bool v2;
    int cfg = nt_decide(p, v2);
    if (!v2) return VSPW_EINVAL;
    static const int force = getenv("VSPW_WINO_TILE") ? atoi(getenv("VSPW_WINO_TILE")) : 0;
    cfg = force ? force : ((cfg == 11 || cfg == 12) ? 12 : 31);
    launch_nt_v2<2>(p, cfg, vspw_stream(stream));
    return vspw_launch_status();
}
