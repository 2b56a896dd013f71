// Ablation probe for the 128x128 (2x2 waves, 2x2 MFMA 32x32x2 fp32 blocks) NT GEMM pipeline: which part of the loop
// keeps the MFMA pipe from its issue-rate peak?  Stand-alone (hipcc tools/probe/mfma_ablate.hip -o mfma_ablate); not
// part of the library.  Flags: 1 = global loads, 2 = ds_write, 4 = barriers, 8 = ds_read fragments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <dlfcn.h>
#include "../../include/vspw_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32, LDA = 36;

template <int FLAGS, int NBUF>
__global__ __launch_bounds__(256) void gemm_nt(const float* __restrict__ A, const float* __restrict__ B,
                                               float* __restrict__ C, int M, int N, int K) {
    constexpr int TM = 128, TN = 128;
    __shared__ __attribute__((aligned(16))) float As[NBUF][TM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][TN * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const int tiles_n = N / TN;
    int bid = blockIdx.x;
    if (FLAGS & 16) {
        const int nblocks = gridDim.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nblocks >> 3, r = nblocks & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + idx;
    }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    const float* ap = A + (size_t)(m0 + lrow) * K + lcol;
    const float* bp = B + (size_t)(n0 + lrow) * K + lcol;
    f32x4 ra[4], rb[4];
    unsigned okbits = (M > 5) ? 15u : (unsigned)tid;
    auto load_tile = [&](int k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap + (size_t)32 * i * K + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp + (size_t)32 * i * K + k);
    };
    auto store_tile = [&](float* Ad, float* Bd) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(&Ad[(lrow + 32 * i) * LDA + lcol]) =
                (!(FLAGS & 32) || ((okbits >> i) & 1u)) ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&Bd[(lrow + 32 * i) * LDA + lcol]) = rb[i];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;
    load_tile(0);
    store_tile(As[0], Bs[0]);
    if (NBUF == 2) store_tile(As[NBUF - 1], Bs[NBUF - 1]);
    load_tile(BK);
    __syncthreads();
    f32x4 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[0][(wm * 64 + i * 32 + l31) * LDA + 4 * lh]);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bs[0][(wn * 64 + j * 32 + l31) * LDA + 4 * lh]);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = (NBUF == 2) ? (kt & 1) : 0;
        if (NBUF == 1) {
            if (FLAGS & 2) store_tile(As[0], Bs[0]);
            if ((FLAGS & 64) && (FLAGS & 1)) {  // early issue: right after the registers are free
                int k = (kt + 1) * BK;
                load_tile(k < K ? k : 0);
            }
            if (FLAGS & 4) __syncthreads();
        }
        const float* Ac = As[cur];
        const float* Bc = Bs[cur];
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            if (FLAGS & 8) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i] = *reinterpret_cast<const f32x4*>(&Ac[(wm * 64 + i * 32 + l31) * LDA + kc * 8 + 4 * lh]);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    b[j] = *reinterpret_cast<const f32x4*>(&Bc[(wn * 64 + j * 32 + l31) * LDA + kc * 8 + 4 * lh]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            if (kc == 0) {
                if (NBUF == 2 && (FLAGS & 2)) store_tile(As[(NBUF - 1) & (cur ^ 1)], Bs[(NBUF - 1) & (cur ^ 1)]);
                if ((FLAGS & 1) && !((FLAGS & 64) && NBUF == 1)) {
                    int k = (kt + (NBUF == 2 ? 2 : 1)) * BK;
                    load_tile(k < K ? k : 0);
                }
            }
        }
        if (FLAGS & 4) __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                C[(size_t)row * N + col] = acc[i][j][r];
            }
    }
}

template <int FLAGS, int NBUF>
static void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid((M / 128) * (N / 128));
    for (int i = 0; i < 2; ++i) gemm_nt<FLAGS, NBUF><<<grid, 256>>>(A, B, C, M, N, K);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) gemm_nt<FLAGS, NBUF><<<grid, 256>>>(A, B, C, M, N, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-44s nbuf %d  %8.3f ms  %7.1f TFLOP/s\n", name, NBUF, ms, 2.0 * M * N * K / ms * 1e-9);
}

int main() {
    const int M = 16384, N = 4096, K = 4096;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMalloc(&B, (size_t)N * K * 4);
    hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20) * 1e-4f - 0.2f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    run<15, 2>("full (gload + ds_write + barrier + ds_read)", A, B, C, M, N, K);
    run<14, 2>("no global loads", A, B, C, M, N, K);
    run<12, 2>("no global loads, no ds_write", A, B, C, M, N, K);
    run<8, 2>("ds_read + mfma only (no barrier)", A, B, C, M, N, K);
    run<4, 2>("barrier + mfma only", A, B, C, M, N, K);
    run<0, 2>("mfma only", A, B, C, M, N, K);
    run<15, 1>("full (gload + ds_write + barrier + ds_read)", A, B, C, M, N, K);
    run<14, 1>("no global loads", A, B, C, M, N, K);
    run<12, 1>("no global loads, no ds_write", A, B, C, M, N, K);
    run<0, 1>("mfma only", A, B, C, M, N, K);
    run<15 + 16, 1>("full + xcd remap", A, B, C, M, N, K);
    run<15 + 32, 1>("full + zero select at ds_write", A, B, C, M, N, K);
    run<15 + 48, 1>("full + xcd remap + select", A, B, C, M, N, K);
    run<15 + 16, 2>("full + xcd remap", A, B, C, M, N, K);
    run<15 + 64, 1>("full, loads issued at the top", A, B, C, M, N, K);
    run<15, 1>("full (repeat)", A, B, C, M, N, K);
    run<15 + 64, 1>("full, loads issued at the top (repeat)", A, B, C, M, N, K);
    // the library's conv kernel on the same GEMM (1x1 conv, n=1, 128x128 pixels, 4096 -> 4096 channels)
    void* lib = dlopen("cvpr2021_vspw_implement_amd/lib/libvspw_hip.so", RTLD_NOW);
    if (lib) {
        typedef int (*fwd_t)(const vspw_conv_desc*, const float*, const float*, const float*, float*, float*, void*);
        fwd_t fwd = (fwd_t)dlsym(lib, "vspw_conv2d_fwd");
        for (int kdim : {4096, 1024, 256}) {
            vspw_conv_desc d = {1, 128, 128, kdim, 128, 128, 4096, 1, 1, 1, 0, 1};
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int i = 0; i < 2; ++i) fwd(&d, A, B, nullptr, C, nullptr, nullptr);
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) fwd(&d, A, B, nullptr, C, nullptr, nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            printf("library vspw_conv2d_fwd 1x1 c=%d k=4096 m=16384   %8.3f ms  %7.1f TFLOP/s\n", kdim, ms, 2.0 * M * N * kdim / ms * 1e-9);
        }
    } else printf("dlopen failed: %s\n", dlerror());
    return 0;
}
