// Ablation probe for the weight-gradient ("TN") loop: C[M][N] = sum_k A[k][M] * B[k][N], both operands k-major in
// memory (k = pixels), staged [k][m] in LDS and read back as scalar fragments (ds_read2_b32) - the structure of
// igemm_tn_v2_kernel.  FLAGS: 1 = global loads, 2 = ds_write, 4 = barriers, 8 = ds_read fragments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;

template <int FLAGS, int PAIR>
__global__ __launch_bounds__(256) void gemm_tn(const float* __restrict__ A, const float* __restrict__ B,
                                               float* __restrict__ C, int M, int N, int K, int chunk) {
    constexpr int TM = 128, TN = 128;
    __shared__ __attribute__((aligned(16))) float As[BK * TM];
    __shared__ __attribute__((aligned(16))) float Bs[BK * TN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const int tiles_n = N / TN;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const int k_begin = blockIdx.y * chunk;
    const int krow = tid >> 5, c4 = (tid & 31) * 4;  // 8 rows per pass, 4 passes
    const float* ap = A + (size_t)(k_begin + krow) * M + m0 + c4;
    const float* bp = B + (size_t)(k_begin + krow) * N + n0 + c4;
    f32x4 ra[4], rb[4];
    auto load_tile = [&](int k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap + (size_t)(k + 8 * i) * M);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp + (size_t)(k + 8 * i) * N);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&As[(krow + 8 * i) * TM + c4]) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&Bs[(krow + 8 * i) * TN + c4]) = rb[i];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = chunk / BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    float a[2] = {As[lh * TM + wm * 64 + l31], As[lh * TM + wm * 64 + 32 + l31]};
    float b[2] = {Bs[lh * TN + wn * 64 + l31], Bs[lh * TN + wn * 64 + 32 + l31]};
    for (int kt = 0; kt < nk; ++kt) {
        if (FLAGS & 2) store_tile();
        if (FLAGS & 4) __syncthreads();
        if (PAIR) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int sp = 0; sp < BK / 4; ++sp) {
                // two MFMA k-steps per fetch: rows (4sp + lh) and (4sp + 2 + lh), 8 bytes each -> ds_read2_b64
                const f32x2 a0 = *reinterpret_cast<const f32x2*>(&As[(4 * sp + lh) * TM + wm * 64 + 2 * l31]);
                const f32x2 a1 = *reinterpret_cast<const f32x2*>(&As[(4 * sp + 2 + lh) * TM + wm * 64 + 2 * l31]);
                const f32x2 b0 = *reinterpret_cast<const f32x2*>(&Bs[(4 * sp + lh) * TN + wn * 64 + 2 * l31]);
                const f32x2 b1 = *reinterpret_cast<const f32x2*>(&Bs[(4 * sp + 2 + lh) * TN + wn * 64 + 2 * l31]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
                if (sp == 1 && (FLAGS & 1)) {
                    int k = (kt + 1) * BK;
                    load_tile(k < chunk ? k : 0);
                }
            }
        } else {
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            if (FLAGS & 8) {
                if (PAIR) {
                    // MFMA block i of a wave covers rows 2*l31 + i (instead of i*32 + l31): one 8-byte read feeds both
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 a2 = *reinterpret_cast<const f32x2*>(&As[(2 * s + lh) * TM + wm * 64 + 2 * l31]);
                    const f32x2 b2 = *reinterpret_cast<const f32x2*>(&Bs[(2 * s + lh) * TN + wn * 64 + 2 * l31]);
                    a[0] = a2[0]; a[1] = a2[1]; b[0] = b2[0]; b[1] = b2[1];
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i] = As[(2 * s + lh) * TM + wm * 64 + i * 32 + l31];
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[j] = Bs[(2 * s + lh) * TN + wn * 64 + j * 32 + l31];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            if (s == 3 && (FLAGS & 1)) {
                int k = (kt + 1) * BK;
                load_tile(k < chunk ? k : 0);
            }
        }
        }
        if (FLAGS & 4) __syncthreads();
    }
    float* out = C + (size_t)blockIdx.y * M * N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[(size_t)row * N + col] = acc[i][j][r];
            }
    }
}

template <int FLAGS, int PAIR = 0>
static void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K, int splits) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid((M / 128) * (N / 128), splits);
    const int chunk = K / splits;
    for (int i = 0; i < 2; ++i) gemm_tn<FLAGS, PAIR><<<grid, 256>>>(A, B, C, M, N, K, chunk);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) gemm_tn<FLAGS, PAIR><<<grid, 256>>>(A, B, C, M, N, K, chunk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-46s splits %3d  %8.3f ms  %7.1f TFLOP/s\n", name, splits, ms, 2.0 * M * N * K / ms * 1e-9);
}

int main() {
    // a layer3 3x3 weight gradient: Cout 256 x (9*256) columns over 36 000 pixels -> here M=256, N=2304, K=36864
    const int M = 256, N = 2304, K = 36864;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)K * M * 4);
    hipMalloc(&B, (size_t)K * N * 4);
    hipMalloc(&C, (size_t)64 * M * N * 4);
    std::vector<float> h((size_t)K * N);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20) * 1e-4f - 0.2f;
    hipMemcpy(A, h.data(), (size_t)K * M * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice);
    for (int splits : {18, 24}) {
        run<15>("full", A, B, C, M, N, K, splits);
        run<15, 1>("full, 8-byte paired fragments", A, B, C, M, N, K, splits);
        run<14>("no global loads", A, B, C, M, N, K, splits);
        run<12>("no global loads, no ds_write", A, B, C, M, N, K, splits);
        run<8>("ds_read + mfma only (no barrier)", A, B, C, M, N, K, splits);
        run<0>("mfma only", A, B, C, M, N, K, splits);
    }
    return 0;
}
