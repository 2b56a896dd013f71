// Fused non-local "dot" affinity for gfx950:   out[b][i][:] = scale * sum_j (q[b][i][:] . k[b][j][:]) * v[b][j][:]
//
// Replaces, for NLBlockND(mode='dot') of the reference (models/non_local.py:105-143, used by Non_local2d / Non_local3d,
// models/non_local_models.py:19-72,124-151), the chain
//     f = torch.matmul(theta_x, phi_x); f_div_C = f / N; y = torch.matmul(f_div_C, g_x)
// WITHOUT materialising the N x N affinity f (N = T*H*W positions: 25 200^2 fp32 = 2.5 GB per sample at T = 7) and,
// with the operands permuted, its three gradients (f has no softmax, so every one of them has the same shape):
//     y       = nl(theta, phi, g)        d theta = nl(dy, g, phi)
//     d g     = nl(phi, theta, dy)       d phi   = nl(g, dy, theta)          (all with scale = 1/N)
//
// Structure (flash-attention-like streaming, but plain sums instead of an online softmax): a workgroup owns 128 query
// rows (4 waves x 32) and one chunk of the key range; it streams 32-key tiles of k and v through a double-buffered LDS
// ring and keeps everything else in registers:
//   1. S^T[key][query] = k_tile . q^T      fp32 MFMA 32x32x2, A = k rows from LDS (ds_read_b128), B = the wave's q rows,
//                                          loaded once from global memory straight into B-operand registers;
//   2. P = scale * S^T                     (the reference divides f by N before the second matmul; same place here);
//   3. out[query][ch] += P . v_tile        S^T was computed transposed precisely so that its accumulator registers ARE
//                                          the A-operand layout of this second MFMA (lane = query, register = key): no
//                                          LDS round trip, no shuffles.  B = v rows from LDS.
// Per tile and wave: 128 MFMAs against 16 ds_read_b128 + 32 ds_read2_b32.  Partial outputs of the key chunks go to a
// workspace and are summed in chunk order by a second kernel (deterministic; N x C x chunks floats instead of N x N).
// Summation order: channels in 8-channel groups (4 + 4 per lane half) inside a dot product, keys in order within a
// tile up to the MFMA pairing (key j with key j+4), tiles in order, chunks in order - fixed, hence bit-reproducible.
#include "common.h"

#define NL_TQ 128   // queries per workgroup (32 per wave)
#define NL_TK 32    // keys per tile

template <int CB>  // C = 32 * CB channels
__global__ __launch_bounds__(256, 2) void nl_dot_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ out,
                                                        int n, float scale, int chunk_keys, long long out_chunk_stride) {
    constexpr int C = 32 * CB;
    constexpr int LDK = C + 4;                 // 16 consecutive rows hit 16 distinct 16-byte slots (ds_read_b128)
    constexpr int F4 = NL_TK * C / 4 / 256;    // float4 per thread per staged tensor (CB: 1, 2, 4)
    __shared__ __attribute__((aligned(16))) float Ks[2][NL_TK * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[2][NL_TK * C];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z;
    const int chunk = blockIdx.y;
    const int q0 = blockIdx.x * NL_TQ + wave * 32;
    const int key_begin = chunk * chunk_keys;
    const int key_end = min(n, key_begin + chunk_keys);
    const float* qb = q + (size_t)b * n * C;
    const float* kb = k + (size_t)b * n * C;
    const float* vb = v + (size_t)b * n * C;

    // this wave's 32 query rows as B-operand registers: lane (l31, lh) holds q[q0 + l31][8*kc + 4*lh + s]
    f32x4 qf[C / 8];
    {
        const float* qrow = qb + (size_t)min(q0 + l31, n - 1) * C + 4 * lh;
#pragma unroll
        for (int kc = 0; kc < C / 8; ++kc) qf[kc] = *reinterpret_cast<const f32x4*>(qrow + 8 * kc);
    }

    f32x16 acc[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // staging: thread t moves float4 number t + 256*i of the [NL_TK][C] tile (row = idx / (C/4))
    f32x4 rk[F4], rv[F4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int key0) {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / (C / 4), c4 = (idx % (C / 4)) * 4;
            const int key = key0 + row;
            const bool ok = key < key_end;
            const size_t off = (size_t)(ok ? key : key_begin) * C + c4;
            const f32x4 a = *reinterpret_cast<const f32x4*>(kb + off);
            const f32x4 c = *reinterpret_cast<const f32x4*>(vb + off);
            rk[i] = ok ? a : zero4;  // keys past the chunk contribute exactly zero (and 0 * finite v = 0)
            rv[i] = ok ? c : zero4;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / (C / 4), c4 = (idx % (C / 4)) * 4;
            *reinterpret_cast<f32x4*>(&Ks[buf][row * LDK + c4]) = rk[i];
            *reinterpret_cast<f32x4*>(&Vs[buf][row * C + c4]) = rv[i];
        }
    };

    const int ntiles = (key_end - key_begin + NL_TK - 1) / NL_TK;
    if (ntiles > 0) {
        load_tile(key_begin);
        store_tile(0);
        if (ntiles > 1) load_tile(key_begin + NL_TK);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const float* Kc = Ks[cur];
        const float* Vc = Vs[cur];
        // 1. S^T = k_tile . q^T  (rows = 32 keys, cols = this wave's 32 queries)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < C / 8; ++kc) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&Kc[l31 * LDK + 8 * kc + 4 * lh]);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[kc][e], s, 0, 0, 0);
            if (kc == 1 && t + 1 < ntiles) {
                // tile t+1 has been in flight since the previous iteration: registers -> the other LDS buffer (its
                // last readers passed the barrier that ended iteration t-1); then start fetching tile t+2
                store_tile(cur ^ 1);
                if (t + 2 < ntiles) load_tile(key_begin + (t + 2) * NL_TK);
            }
        }
        // 2. P = scale * S^T ;  3. out += P . v_tile
        //    register r of lane (l31, lh) is P[query l31][key (r&3) + 8*(r>>2) + 4*lh]: exactly an A operand
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = s[r] * scale;
            const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int j = 0; j < CB; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, Vc[key * C + 32 * j + l31], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }

    // D layout: register r of lane (l31, lh) = out[query (r&3) + 8*(r>>2) + 4*lh][channel 32*j + l31]
    float* ob = out + (size_t)chunk * out_chunk_stride + (size_t)b * n * C;
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < n) ob[(size_t)row * C + 32 * j + l31] = acc[j][r];
        }
}

// out = sum over chunks of part[chunk] in chunk order (float4 lanes)
__global__ __launch_bounds__(256) void nl_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                        long long n4, int chunks) {
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        f32x4 s = p4[i];
        for (int c = 1; c < chunks; ++c) s += p4[(size_t)c * n4 + i];
        reinterpret_cast<f32x4*>(out)[i] = s;
    }
}

// key-range chunks: enough workgroups to fill 256 CUs x 2 several times over, at least 4 tiles per chunk
static void nl_plan(int b, int n, int& chunks, int& chunk_keys) {
    const long long qtiles = (long long)vspw_cdiv(n, NL_TQ) * b;
    long long want = (4096 + qtiles - 1) / qtiles;
    const long long max_chunks = (n + 4 * NL_TK - 1) / (4 * NL_TK);
    if (want > max_chunks) want = max_chunks;
    if (want < 1) want = 1;
    long long ck = (n + want - 1) / want;
    ck = ((ck + NL_TK - 1) / NL_TK) * NL_TK;
    chunk_keys = (int)ck;
    chunks = (int)((n + ck - 1) / ck);
}

extern "C" size_t vspw_nl_dot_workspace(int b, int n, int c) {
    if (b <= 0 || n <= 0 || c <= 0) return 0;
    int chunks, ck;
    nl_plan(b, n, chunks, ck);
    return chunks > 1 ? (size_t)chunks * b * n * c * sizeof(float) : 0;
}

extern "C" int vspw_nl_dot(const float* q, const float* k, const float* v, float* out, int b, int n, int c,
                           float scale, void* ws, size_t ws_bytes, void* stream) {
    if (!q || !k || !v || !out || b <= 0 || n <= 0) return VSPW_EINVAL;
    if (c != 32 && c != 64 && c != 128) return VSPW_EINVAL;
    if (b > 65535 || (long long)b * n * c > 0x7fffffffLL) return VSPW_EINVAL;
    int chunks, ck;
    nl_plan(b, n, chunks, ck);
    const size_t need = chunks > 1 ? (size_t)chunks * b * n * c * sizeof(float) : 0;
    if (need > ws_bytes || (need > 0 && !ws)) return VSPW_EINVAL;
    if (chunks > 65535) return VSPW_EINVAL;
    float* dst = chunks > 1 ? reinterpret_cast<float*>(ws) : out;
    const long long stride = (long long)b * n * c;
    const dim3 grid(vspw_cdiv(n, NL_TQ), chunks, b);
    hipStream_t st = vspw_stream(stream);
    if (c == 128)
        hipLaunchKernelGGL(nl_dot_kernel<4>, grid, dim3(256), 0, st, q, k, v, dst, n, scale, ck, stride);
    else if (c == 64)
        hipLaunchKernelGGL(nl_dot_kernel<2>, grid, dim3(256), 0, st, q, k, v, dst, n, scale, ck, stride);
    else
        hipLaunchKernelGGL(nl_dot_kernel<1>, grid, dim3(256), 0, st, q, k, v, dst, n, scale, ck, stride);
    int rc = vspw_launch_status();
    if (rc != VSPW_OK || chunks == 1) return rc;
    const long long n4 = stride / 4;
    hipLaunchKernelGGL(nl_reduce_kernel, dim3(vspw_stream_grid(n4, 256)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(ws), out, n4, chunks);
    return vspw_launch_status();
}
