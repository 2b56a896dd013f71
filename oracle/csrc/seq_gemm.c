/* ORACLE (test infrastructure, NOT the product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this).
 *
 * C[m][n] = sum_k A[m][k] * B[n][k] in float32 where every output element is ONE fused-multiply-add chain in k order
 * (acc = fmaf(a, b, acc), k = 0, 1, 2, ...): the accumulation order of a matrix-core k-loop.  The reference's
 * arithmetic lives in ATen's CPU GEMMs (README.md:13 pins PyTorch 1.3.1: MKL / Eigen kernels that block the k loop and
 * add block results), and numpy's OpenBLAS does the same; neither order is part of conv2d's / matmul's semantics
 * (reference call sites: models/resnet.py:61-66, models/ocr_modules/spatial_ocr_block.py:61,105,267,274,
 * models/non_local.py:116,139).  oracle/np_ops.py switches its float32 GEMMs to this routine (set_gemm("sequential"))
 * when a test needs the float32 noise floor of the k-sequential order instead of OpenBLAS's blocked one: the rounding
 * error of a chain grows like sqrt(K), and K reaches 36 864 on this path.
 * `chunk` > 0 restarts the chain every `chunk` k and adds the partial results in order (a split-K reduction).
 *
 * Vectorised ACROSS output columns (8 floats per register), so the per-element order is untouched.
 * Build: gcc -O3 -mavx2 -mfma -fopenmp -shared -fPIC (oracle/build_oracle.py). */
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

#define MR 6
#define NV 2 /* ymm registers per row: 16 columns */

static void kernel(const float* A, long lda, const float* Bt, long ldbt, float* C, long ldc, int mr, long k0, long k1,
                   int first) {
    /* rows m..m+mr of A, 16 columns of Bt ([K][Npad], row k holds B[:, k]); accumulates k0..k1 into C tile */
    __m256 acc[MR][NV];
    for (int i = 0; i < MR; ++i)
        for (int j = 0; j < NV; ++j) acc[i][j] = _mm256_setzero_ps();
    for (long k = k0; k < k1; ++k) {
        const __m256 b0 = _mm256_loadu_ps(Bt + k * ldbt), b1 = _mm256_loadu_ps(Bt + k * ldbt + 8);
        for (int i = 0; i < mr; ++i) {
            const __m256 a = _mm256_broadcast_ss(A + i * lda + k);
            acc[i][0] = _mm256_fmadd_ps(a, b0, acc[i][0]);
            acc[i][1] = _mm256_fmadd_ps(a, b1, acc[i][1]);
        }
    }
    for (int i = 0; i < mr; ++i)
        for (int j = 0; j < NV; ++j) {
            float* c = C + i * ldc + 8 * j;
            if (first)
                _mm256_storeu_ps(c, acc[i][j]);
            else
                _mm256_storeu_ps(c, _mm256_add_ps(_mm256_loadu_ps(c), acc[i][j]));
        }
}

int seq_gemm_nt_f32(const float* A, const float* B, float* C, long M, long N, long K, long chunk) {
    if (M <= 0 || N <= 0) return 0;
    const long Np = (N + 15) / 16 * 16;
    float* Bt = (float*)aligned_alloc(64, (size_t)((K > 0 ? K : 1) * Np) * sizeof(float));
    float* Cp = (float*)aligned_alloc(64, (size_t)(M * Np) * sizeof(float));
    if (!Bt || !Cp) {
        free(Bt);
        free(Cp);
        return -1;
    }
    if (chunk <= 0 || chunk > K) chunk = K > 0 ? K : 1;
#pragma omp parallel for schedule(static)
    for (long k = 0; k < K; ++k) {
        float* row = Bt + k * Np;
        for (long n = 0; n < N; ++n) row[n] = B[n * K + k];
        for (long n = N; n < Np; ++n) row[n] = 0.0f;
    }
    const long mblocks = (M + MR - 1) / MR, nblocks = Np / 16;
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
    for (long mb = 0; mb < mblocks; ++mb)
        for (long nb = 0; nb < nblocks; ++nb) {
            const long m = mb * MR;
            const int mr = (int)(M - m < MR ? M - m : MR);
            float* c = Cp + m * Np + nb * 16;
            if (K <= 0) {
                for (int i = 0; i < mr; ++i) memset(c + i * Np, 0, 16 * sizeof(float));
                continue;
            }
            for (long k0 = 0; k0 < K; k0 += chunk)
                kernel(A + m * K, K, Bt + nb * 16, Np, c, Np, mr, k0, k0 + chunk < K ? k0 + chunk : K, k0 == 0);
        }
#pragma omp parallel for schedule(static)
    for (long m = 0; m < M; ++m) memcpy(C + m * N, Cp + m * Np, (size_t)N * sizeof(float));
    free(Bt);
    free(Cp);
    return 0;
}
