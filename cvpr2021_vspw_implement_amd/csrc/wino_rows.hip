// Winograd F(2x2, 3x3): the four GEMMs of one transform ROW in one workgroup, half of the output transform in registers.
//
// winograd.hip runs the 16 transform positions xi = (a, b) as 16 independent GEMMs M[xi] = V[xi] U[xi]^T and leaves the
// whole output transform Y = A^T M A to a second kernel: M (16 planes, 4x the convolution's output) is written once and
// read once.  A^T M A is separable - Y[i][j] = sum_a A^T[i][a] (sum_b M[a][b] A^T[j][b]) - and the inner sum only mixes
// the four positions of one row a:
//     P[a][0] = M[a][0] + M[a][1] + M[a][2],      P[a][1] = M[a][1] - M[a][2] - M[a][3].
// Here ONE workgroup owns a (tile block x output-channel block) of row a and runs the four GEMMs back to back through
// one software pipeline (4 * Cin / 32 K-tiles instead of Cin / 32: prologue, epilogue and the dispatch gap between
// workgroups are paid once per four GEMMs) with only TWO accumulator sets, in the order b = 1, 2, 0, 3:
//     X <- M[a][1];  Y <- M[a][2];  (X, Y) <- (X + Y, X - Y)   [2 vector ops per accumulator register and workgroup]
//     X += V[a][0] U[a][0]^T  (the MFMAs keep accumulating)  = P[a][0]
//     Y += (-V[a][3]) U[a][3]^T  (operand negated while it is staged)  = P[a][1]
// It writes the 8 planes P [4][2][Tpad][rows] - HALF of M - and the second kernel (wino_output_kernel<.., ROWS>)
// finishes Y[i][j] = sum_a A^T[i][a] P[a][j] reading half as much.
// Fusing the other half as well would need all 16 positions' accumulators (or a 16-GEMM serial walk per workgroup:
// 282 workgroups for the layer3 shape) - see DESIGN.md section 3.
//
// The MFMA loop is the pointwise loop of conv_igemm.hip (raw buffer loads -> registers -> LDS rows of 36 floats ->
// ds_read_b128 fragments -> v_mfma_f32_32x32x2_f32; one LDS buffer, the next K-tile's loads in flight during the MFMAs).
// FUSED: the A operand is (B^T d B)[a][b] of the tile's 4x4 input patch, combined from four pixels while it is staged
// (V is never written) - the data-gradient / inference form, conv_igemm.hip's AFF 4.
#include "common.h"
#include <cstdlib>

#define WR_BK 32
#define WR_LDA 36

struct WinoRowsP {
    const float* a;  // plain: V [16][T][c]; FUSED: NHWC source [nb][h][w][lds]
    const float* u;  // [16][rows][c]
    float* tp;       // [4][2][tpad][rows]
    int T, tpad, c, rows;
    int nb, h, w, lds, tpi, d, th, tw;  // FUSED only: source geometry, tiles per image, dilation, tiles per sub-grid
    int dbg;  // experiments (vspw_wino_rows_config(tile + 100 * dbg)): 1 = no stores, 2 = no K loop
};

template <int WGM, int WM, int WN, int FUSED, int MINB>
__global__ __launch_bounds__(256, MINB) void igemm_nt_wrows_kernel(WinoRowsP p) {
    constexpr int WGN = 4 / WGM;
    constexpr int TM = 32 * WM * WGM, TN = 32 * WN * WGN;
    constexpr int RA = TM / 32, RB = TN / 32;
    constexpr unsigned OOR = 0x80000000u;
    // operand tiles, reused by the epilogue as per-wave 32 x LDA transposition scratch
    __shared__ __attribute__((aligned(16))) float smem[(TM + TN) * WR_LDA];
    static_assert((TM + TN) * WR_LDA >= 4 * 32 * WR_LDA, "epilogue scratch: 32 x LDA floats per wave");
    float* As = smem;
    float* Bs = smem + TM * WR_LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (WGM == 2) ? (wave >> 1) : 0, wn = (WGM == 2) ? (wave & 1) : wave;
    const int l31 = lane & 31, lh = lane >> 5;
    __builtin_amdgcn_s_setprio(2);

    const int tiles_n = p.rows / TN;
    int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int ar = vb & 3;  // transform row: the four rows of a tile block are neighbours in the XCD's tile range
    vb >>= 2;
    const int tile_n = vb % tiles_n, tile_m = vb / tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;

    const size_t plane_a = (size_t)p.T * p.c, plane_u = (size_t)p.rows * p.c;
    const int img0 = FUSED ? m0 / p.tpi : 0;
    const char* a0 = FUSED ? reinterpret_cast<const char*>(p.a + (size_t)img0 * p.h * p.w * p.lds)
                           : reinterpret_cast<const char*>(p.a + (size_t)(ar * 4) * plane_a);
    // (the scalar offset that selects the position's plane may or may not be part of the hardware range check: the
    // plain operands' descriptors span all four planes of the row)
    const long long a_rem = FUSED ? (long long)(p.nb - img0) * p.h * p.w * p.lds * 4 : (long long)plane_a * 16;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(a0), 0, (int)(unsigned)(a_rem < (long long)OOR ? a_rem : (long long)OOR), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.u + (size_t)(ar * 4) * plane_u + (size_t)n0 * p.c)), 0,
        (int)(unsigned)(((long long)plane_u * 4 - (long long)n0 * p.c) * 4), 0x00020000);

    unsigned a_voff[RA], b_voff[RB];
    unsigned a_w4[FUSED ? 4 : 1][RA];
    float w4c[4] = {0.f, 0.f, 0.f, 0.f};
    int t_gy0[FUSED ? RA : 1], t_gx[FUSED ? RA : 1], t_row[FUSED ? RA : 1], t_sy[FUSED ? RA : 1], t_sx[FUSED ? RA : 1];
    (void)t_gy0; (void)t_gx; (void)t_row; (void)t_sy; (void)t_sx;
    // B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]: row q has two non-zeros, at patch index I0(q), I1(q)
    auto bt_i0 = [](int q) { return q == 0 ? 0 : 1; };
    auto bt_i1 = [](int q) { return q == 3 ? 3 : 2; };
    auto bt_s0 = [](int q) { return q == 2 ? -1.f : 1.f; };
    auto bt_s1 = [](int q) { return (q == 0 || q == 3) ? -1.f : 1.f; };
    auto set_b = [&](int b) {  // FUSED: the four patch pixels of (B^T d B)[ar][b] for each staged row
        if constexpr (FUSED) {
            const int ia0 = bt_i0(ar), ia1 = bt_i1(ar), jb0 = bt_i0(b), jb1 = bt_i1(b);
            const float sg3 = b == 3 ? -1.f : 1.f;  // position 3 enters P[a][1] with a minus sign
            const float sa0 = sg3 * bt_s0(ar), sa1 = sg3 * bt_s1(ar), sb0 = bt_s0(b), sb1 = bt_s1(b);
            w4c[0] = sa0 * sb0; w4c[1] = sa0 * sb1; w4c[2] = sa1 * sb0; w4c[3] = sa1 * sb1;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int gy0 = t_gy0[i] + ia0, gy1 = t_gy0[i] + ia1, gx0 = t_gx[i] + jb0, gx1 = t_gx[i] + jb1;
                const int py0 = gy0 * p.d + t_sy[i], py1 = gy1 * p.d + t_sy[i];
                const int px0 = gx0 * p.d + t_sx[i], px1 = gx1 * p.d + t_sx[i];
                const bool oy0 = (gy0 >= 0) & (py0 < p.h), oy1 = (gy1 >= 0) & (py1 < p.h);
                const bool ox0 = (gx0 >= 0) & (px0 < p.w), ox1 = (gx1 >= 0) & (px1 < p.w);
                const int rb0 = (t_row[i] + py0) * p.w, rb1 = (t_row[i] + py1) * p.w;
                a_w4[0][i] = (oy0 & ox0) ? (unsigned)((rb0 + px0) * p.lds + lcol) * 4u : OOR;
                a_w4[1][i] = (oy0 & ox1) ? (unsigned)((rb0 + px1) * p.lds + lcol) * 4u : OOR;
                a_w4[2][i] = (oy1 & ox0) ? (unsigned)((rb1 + px0) * p.lds + lcol) * 4u : OOR;
                a_w4[3][i] = (oy1 & ox1) ? (unsigned)((rb1 + px1) * p.lds + lcol) * 4u : OOR;
            }
        }
    };
    if constexpr (FUSED) {
        const int per = p.th * p.tw;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int t = min(m0 + lrow + 32 * i, p.T - 1);
            const int img = t / p.tpi;
            int r = t - img * p.tpi;
            const int sg = r / per;
            r -= sg * per;
            t_sy[i] = sg / p.d;
            t_sx[i] = sg - t_sy[i] * p.d;
            const int ty = r / p.tw, tx = r - ty * p.tw;
            t_gy0[i] = 2 * ty - 1;
            t_gx[i] = 2 * tx - 1;
            t_row[i] = (img - img0) * p.h;
            a_voff[i] = 0;
        }
        set_b(1);
    } else {
#pragma unroll
        for (int i = 0; i < RA; ++i) a_voff[i] = (unsigned)(min(m0 + lrow + 32 * i, p.T - 1) * p.c + lcol) * 4u;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) b_voff[i] = (unsigned)((lrow + 32 * i) * p.c + lcol) * 4u;

    // state of the NEXT K-tile to fetch: step bq of the position order 1, 2, 0, 3, channel base kb
    int kb = 0, bq = 0;
    bool more = true;
    bool neg = false;  // plain operand: the staged tile belongs to position 3
    auto b_of = [](int q) { return q == 0 ? 1 : (q == 1 ? 2 : (q == 2 ? 0 : 3)); };
    const unsigned pa4 = FUSED ? 0u : (unsigned)(plane_a * 4), pu4 = (unsigned)(plane_u * 4);
    f32x4 ra[RA], rb[RB], rw[FUSED ? 3 : 1][RA];
    auto load_tile = [&]() {
        if (more) {
            const unsigned bb = (unsigned)b_of(bq);
            const unsigned sa = bb * pa4 + (unsigned)kb * 4u, sb = bb * pu4 + (unsigned)kb * 4u;
            if constexpr (FUSED) {
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_w4[0][i], sa, 0));
#pragma unroll
                    for (int q = 1; q < 4; ++q)
                        rw[q - 1][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_w4[q][i], sa, 0));
                }
            } else {
#pragma unroll
                for (int i = 0; i < RA; ++i)
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_voff[i], sa, 0));
            }
#pragma unroll
            for (int i = 0; i < RB; ++i)
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, b_voff[i], sb, 0));
        }
    };
    auto advance = [&]() {
        kb += WR_BK;
        if (kb == p.c) {
            kb = 0;
            ++bq;
            more = bq < 4;
            if (FUSED && more) set_b(b_of(bq));
            if (!FUSED) neg = bq == 3;
        }
    };
    // (advance() runs after the staging registers of the previous tile were stored: w4c / neg always belong to the
    // tile the registers hold)
    auto store_tile = [&]() {
        if constexpr (FUSED) {
#pragma unroll
            for (int i = 0; i < RA; ++i)
                ra[i] = (w4c[0] * ra[i] + w4c[1] * rw[0][i]) + (w4c[2] * rw[1][i] + w4c[3] * rw[2][i]);
        } else if (neg) {  // (uniform branch, the last quarter of the K-tiles)
#pragma unroll
            for (int i = 0; i < RA; ++i) ra[i] = -ra[i];
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&As[(lrow + 32 * i) * WR_LDA + lcol]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * i) * WR_LDA + lcol]) = rb[i];
    };

    f32x16 P0[WM][WN], P1[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) P0[i][j][r] = P1[i][j][r] = 0.f;
    const int nkb = (p.dbg & 2) ? 0 : p.c / WR_BK;
    auto gemm = [&](f32x16 (&acc)[WM][WN]) {  // one transform position: Cin / 32 K-tiles of the shared pipeline
        for (int kt = 0; kt < nkb; ++kt) {
            store_tile();
            __syncthreads();
            advance();
#pragma unroll
            for (int kc = 0; kc < WR_BK / 8; ++kc) {
                f32x4 fa[WM], fb[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    fa[i] = *reinterpret_cast<const f32x4*>(&As[(wm * 32 * WM + i * 32 + l31) * WR_LDA + kc * 8 + 4 * lh]);
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    fb[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn * 32 * WN + j * 32 + l31) * WR_LDA + kc * 8 + 4 * lh]);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
                if (kc == 0) load_tile();
            }
            __syncthreads();
        }
    };
    // stores: every tile is interior (tpad rows, rows % TN == 0): 32x32 blocks through a wave-private LDS scratch so
    // that a lane stores 16 bytes (4 consecutive channels).  Raw buffer stores: the wave's block origin is the scalar
    // base, one per-lane offset register serves every block.
    float* scr = smem + wave * (32 * WR_LDA);  // (the operand tiles are dead behind the loop's last barrier)
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    const size_t plane_t = (size_t)p.tpad * p.rows;
    const unsigned st_voff = (unsigned)(erow * p.rows + ec4) * 4u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto store_plane = [&](f32x16 (&acc)[WM][WN], int j01) {
        float* org = p.tp + (size_t)(ar * 2 + j01) * plane_t + (size_t)(m0 + wm * 32 * WM) * p.rows + n0 + wn * 32 * WN;
        const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(org, 0, 0x7ffffffc, 0x00020000);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * WR_LDA + l31] = acc[i][j][r];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(&scr[(rq * 8 + erow) * WR_LDA + ec4])),
                        rs_o, st_voff, (unsigned)(((i * 32 + rq * 8) * p.rows + j * 32) * 4), 0);
                // one block at a time: unfenced, the scheduler reads every block back from the scratch before the first
                // store (16 registers per block) and spills the accumulators
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    __builtin_amdgcn_s_setprio(0);
    load_tile();
    gemm(P0);  // M[a][1]
    gemm(P1);  // M[a][2]
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const f32x16 x = P0[i][j], y = P1[i][j];
            P0[i][j] = x + y;
            P1[i][j] = x - y;
        }
    gemm(P0);  // + M[a][0]
    gemm(P1);  // - M[a][3]
    __builtin_amdgcn_s_setprio(2);
    // (tried twice: storing P[a][0] - final at this point - before the fourth GEMM, so that its stores drain behind the
    // MFMAs instead of at the end of a one-round launch where every workgroup stores at once (15 of 169 us,
    // tools/diag/wino_rows_probe.py "no stores"): with plain and with raw-buffer stores the 96-row tile then spills
    // 88-140 registers at three workgroups per CU (157 -> 205 us))
    if (p.dbg & 1) {  // keep the accumulators alive through a never-true store
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += P0[i][j][r] + P1[i][j][r];
        if (t == 123.456f) p.tp[tid] = t;
        return;
    }
    store_plane(P0, 0);
    store_plane(P1, 1);
}

// Tile choice (code as in conv_igemm.hip's nt_pick_tile): 64 x 128 (12), 96 x 128 (31) at three workgroups per CU when
// V is read from memory, two with the fused operand (its four staged pixels per row need the registers); 128 x 128 (22)
// at two.  vspw_wino_rows_config(code) / VSPW_WROWS_TILE force one (experiments); 0 = automatic.
static int wr_forced = -1;
static int wr_dbg = 0;
static int wr_tile_rows(int cfg) { return cfg == 22 ? 128 : (cfg == 31 ? 96 : 64); }
static int wr_tile(long long T, int rows, bool fused = false) {
    if (wr_forced < 0) wr_forced = getenv("VSPW_WROWS_TILE") ? atoi(getenv("VSPW_WROWS_TILE")) : 0;
    if (wr_forced == 12 || wr_forced == 31 || (wr_forced == 22 && !fused)) return wr_forced;
    // makespan model of nt_pick_tile: the busiest CU runs ceil(workgroups / 256) of them; per-tile efficiencies fitted
    // to tools/diag/wino_rows_probe.py (256 -> 256, 512 -> 512 d2 / d4, 1024 -> 512 at 60 x 60 x 10 frames)
    const int rows_[3] = {128, 96, 64}, code[3] = {22, 31, 12};
    const double eff[3] = {1.0, 0.95, 0.92};
    int best = 1;
    double best_cost = 1e300;
    for (int i = fused ? 1 : 0; i < 3; ++i) {
        const long long wg = 4 * ((T + rows_[i] - 1) / rows_[i]) * (rows / 128);
        const double cost = (double)((wg + 255) / 256) * rows_[i] / eff[i];
        if (cost < best_cost * 0.999) {
            best_cost = cost;
            best = i;
        }
    }
    return code[best];
}

extern "C" int vspw_wino_rows_config(int tile) {
    const int dbg = tile / 100;
    tile %= 100;
    if ((tile != 0 && tile != 12 && tile != 31 && tile != 22) || dbg < 0 || dbg > 3) return VSPW_EINVAL;
    wr_forced = tile;
    wr_dbg = dbg;
    return VSPW_OK;
}

static bool wr_geom(const vspw_conv_desc* d, int channels, int rows, WinoRowsP& p, bool fused) {
    if (!d || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->dil < 1 || d->pad != d->dil || d->pad_w != d->dil ||
        d->oh != d->h || d->ow != d->w || d->n < 1 || channels < WR_BK || channels % WR_BK != 0 || rows < 128 ||
        rows % 128 != 0)
        return false;
    const int dl = d->dil;
    const int th = ((d->h + dl - 1) / dl + 1) / 2, tw = ((d->w + dl - 1) / dl + 1) / 2;
    const long long tpi = (long long)dl * dl * th * tw, T = tpi * d->n;
    if (T > 0x3fffffffLL) return false;
    const int tm = wr_tile_rows(wr_tile(T, rows, fused));
    const long long tpad = (T + tm - 1) / tm * tm;
    // 32-bit byte offsets: four operand planes behind one scalar base, per-lane offsets inside one plane / one image span
    if (4 * T * channels * 4 >= 0x7fffffffLL || 4LL * rows * channels * 4 >= 0x7fffffffLL) return false;
    const long long span = (128 / tpi + 2) * (long long)d->h * d->w * channels * 4;
    if (span >= (1LL << 30) || tpad * rows / 64 * 4 > 0x3fffffffLL) return false;
    p.T = (int)T; p.tpad = (int)tpad; p.c = channels; p.rows = rows; p.dbg = wr_dbg;
    p.nb = d->n; p.h = d->h; p.w = d->w; p.lds = channels; p.tpi = (int)tpi; p.d = dl; p.th = th; p.tw = tw;
    return true;
}

// Padded tile count of the P planes ([4][2][tpad][rows] floats) for this geometry; 0: not supported (callers take the
// 16-GEMM path of winograd.hip).
extern "C" long long vspw_wino_rows_tpad(const vspw_conv_desc* d, int channels, int rows, int fused) {
    WinoRowsP p;
    return wr_geom(d, channels, rows, p, fused != 0) ? p.tpad : 0;
}

// 1 when this form is expected to beat the 16-GEMM form (measured, tools/diag/wino_rows_probe.py: it needs enough
// workgroups - a quarter of the 16-GEMM grid - to fill the chip, and the halved M traffic stops mattering against long
// reductions): the dispatch rule of ops._wino_conv.  A forced tile (vspw_wino_rows_config) means yes.
extern "C" int vspw_wino_rows_prefer(const vspw_conv_desc* d, int channels, int rows, int fused) {
    WinoRowsP p;
    if (!wr_geom(d, channels, rows, p, fused != 0)) return 0;
    if (wr_forced > 0) return 1;
    const int tm = wr_tile_rows(wr_tile(p.T, rows, fused != 0));
    const long long wg = 4LL * (p.tpad / tm) * (rows / 128);
    // (VSPW_WROWS_MAXROWS, experiments: widest output of the plain form.  Measured on TCB-OCR, whose head has a
    // 2 048-row data gradient: limiting it to 1 024 rows is slower by 0.3-0.5 ms per step)
    static const int max_rows = getenv("VSPW_WROWS_MAXROWS") ? atoi(getenv("VSPW_WROWS_MAXROWS")) : 1 << 30;
    return fused ? (wg >= 350 && channels <= 2048) : (wg >= 700 && channels <= 512 && p.T >= 4096 && rows <= max_rows);
}

template <int FUSED>
static int wr_launch(WinoRowsP& p, hipStream_t st) {
    const int cfg = wr_tile(p.T, p.rows, FUSED != 0);
    const int tm = wr_tile_rows(cfg);
    const long long grid = 4LL * (p.tpad / tm) * (p.rows / 128);
    if (grid > 0x7fffffffLL || p.tpad % tm != 0) return VSPW_EINVAL;
    if (cfg == 22) {
        if constexpr (FUSED)
            return VSPW_EINVAL;  // (the fused operand's staging registers do not fit next to 128 accumulators)
        else
            hipLaunchKernelGGL((igemm_nt_wrows_kernel<2, 2, 2, 0, 2>), dim3((unsigned)grid), dim3(256), 0, st, p);
    } else if (cfg == 31) {
        hipLaunchKernelGGL((igemm_nt_wrows_kernel<1, 3, 1, FUSED, FUSED ? 2 : 3>), dim3((unsigned)grid), dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL((igemm_nt_wrows_kernel<2, 1, 2, FUSED, FUSED ? 2 : 3>), dim3((unsigned)grid), dim3(256), 0, st, p);
    }
    return vspw_launch_status();
}

// P[a][j] = sum_b (V[a][b] U[a][b]^T) A^T[j][b]:  v [16][T][channels], u [16][rows][channels] -> tp [4][2][tpad][rows]
extern "C" int vspw_wino_gemm_rows(const vspw_conv_desc* d, const float* v, int channels, const float* u, int rows,
                                   float* tp, void* stream) {
    WinoRowsP p;
    if (!v || !u || !tp || !wr_geom(d, channels, rows, p, false)) return VSPW_EINVAL;
    p.a = v; p.u = u; p.tp = tp;
    return wr_launch<0>(p, vspw_stream(stream));
}

// ... with V evaluated from the NHWC tensor src ([n][h][w][channels]: x, or dY for the data gradient) while the operand
// is staged (vspw_wino_gemm_fused's operand form).
extern "C" int vspw_wino_gemm_fused_rows_ex(const vspw_conv_desc* d, const float* src, long long ldx, int channels,
                                            const float* u, int rows, float* tp, void* stream);
extern "C" int vspw_wino_gemm_fused_rows(const vspw_conv_desc* d, const float* src, int channels, const float* u,
                                         int rows, float* tp, void* stream) {
    return vspw_wino_gemm_fused_rows_ex(d, src, channels, channels, u, rows, tp, stream);
}

// ... with the source read at pixel stride ldx >= channels (the first `channels` channels of a wider NHWC buffer: the
// flow network's concatenation buffers, cf. vspw_wino_gemm_fused_ex).
extern "C" int vspw_wino_gemm_fused_rows_ex(const vspw_conv_desc* d, const float* src, long long ldx, int channels,
                                            const float* u, int rows, float* tp, void* stream) {
    WinoRowsP p;
    if (!src || !u || !tp || ldx < channels || (ldx & 3) || ldx > 0x7fffffff || !wr_geom(d, channels, rows, p, true))
        return VSPW_EINVAL;
    if ((long long)(128 / p.tpi + 2) * d->h * d->w * ldx * 4 >= (1LL << 30)) return VSPW_EINVAL;
    p.a = src; p.u = u; p.tp = tp;
    p.lds = (int)ldx;
    return wr_launch<1>(p, vspw_stream(stream));
}
