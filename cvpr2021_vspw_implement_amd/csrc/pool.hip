// Pooling kernels (NHWC fp32, HBM-bound gathers; no atomics: every adjoint is a gather).
//   maxpool 3x3/s2/p1       : models/resnet.py:109 (stem)
//   adaptive average pool   : models/clip_psp.py:85-87,160-166; models/models.py:947,972 (PPM pyramid)
//   temporal mean over T    : models/clip_psp.py:181-188 (Temporal Context Blending of TCB-PSP),
//                             models/ocr_modules/spatial_ocr_block.py:108-109 (TCB-OCR context mean)
#include "common.h"

// ---- max pool ------------------------------------------------------------------------------------
// ATen scans the window row-major and keeps the FIRST maximum (strict '>' , NaN propagates); the winning tap
// index is stored so that the backward pass routes the gradient exactly like the reference.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int n, int h, int w, int c,
                                                          int oh, int ow) {
    const long long total = (long long)n * oh * ow * c;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int ch = (int)(i % c);
        long long r = i / c;
        const int ox = (int)(r % ow);
        r /= ow;
        const int oy = (int)(r % oh);
        const int img = (int)(r / oh);
        float best = -INFINITY;
        int bi = -1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= h) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= w) continue;
                const float v = x[(((size_t)img * h + iy) * w + ix) * c + ch];
                if (bi < 0 || v > best || v != v) {
                    best = v;
                    bi = ky * 3 + kx;
                }
            }
        }
        y[i] = best;
        idx[i] = (uint8_t)bi;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int n, int h, int w, int c, int oh,
                                                          int ow) {
    const long long total = (long long)n * h * w * c;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int ch = (int)(i % c);
        long long r = i / c;
        const int ix = (int)(r % w);
        r /= w;
        const int iy = (int)(r % h);
        const int img = (int)(r / h);
        float g = 0.f;
        // windows containing (iy, ix): oy with oy*2-1+ky == iy, ky in 0..2
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (ty & 1)) continue;
            const int oy = ty >> 1;
            if (oy >= oh) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (tx & 1)) continue;
                const int ox = tx >> 1;
                if (ox >= ow) continue;
                const size_t o = (((size_t)img * oh + oy) * ow + ox) * c + ch;
                if (idx[o] == (uint8_t)(ky * 3 + kx)) g += dy[o];
            }
        }
        dx[i] = g;
    }
}

// float4 variants (c % 4 == 0: the 128-channel stem output): one thread = 4 channels of one pixel, one 32-bit
// pixel decode per 16 bytes instead of 64-bit div/mod per element; grid.y = image.
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void maxpool_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           uint8_t* __restrict__ idx, int h, int w, int c4, int oh,
                                                           int ow) {
    const int img = blockIdx.y;
    const int total = oh * ow * c4;
    const f32x4* xb = reinterpret_cast<const f32x4*>(x) + (size_t)img * h * w * c4;
    f32x4* yb = reinterpret_cast<f32x4*>(y) + (size_t)img * total;
    u8x4* ib = reinterpret_cast<u8x4*>(idx) + (size_t)img * total;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ch = i % c4;
        const int r = i / c4;
        const int ox = r % ow, oy = r / ow;
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= h) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= w) continue;
                const f32x4 v = xb[((size_t)iy * w + ix) * c4 + ch];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (bi[e] < 0 || v[e] > best[e] || v[e] != v[e]) {  // first maximum wins ties; NaN propagates
                        best[e] = v[e];
                        bi[e] = ky * 3 + kx;
                    }
            }
        }
        yb[i] = best;
        u8x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (unsigned char)bi[e];
        ib[i] = o;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd4_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                           float* __restrict__ dx, int h, int w, int c4, int oh, int ow) {
    const int img = blockIdx.y;
    const int total = h * w * c4;
    const f32x4* gb = reinterpret_cast<const f32x4*>(dy) + (size_t)img * oh * ow * c4;
    const u8x4* ib = reinterpret_cast<const u8x4*>(idx) + (size_t)img * oh * ow * c4;
    f32x4* db = reinterpret_cast<f32x4*>(dx) + (size_t)img * total;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ch = i % c4;
        const int r = i / c4;
        const int ix = r % w, iy = r / w;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // windows containing (iy, ix): oy with oy*2-1+ky == iy, ky in 0..2 (same visiting order as the scalar kernel)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (ty & 1)) continue;
            const int oy = ty >> 1;
            if (oy >= oh) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (tx & 1)) continue;
                const int ox = tx >> 1;
                if (ox >= ow) continue;
                const size_t o = ((size_t)oy * ow + ox) * c4 + ch;
                const u8x4 id = ib[o];
                const f32x4 v = gb[o];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (id[e] == (unsigned char)(ky * 3 + kx)) g[e] += v[e];
            }
        }
        db[i] = g;
    }
}

// ---- adaptive average pool ---------------------------------------------------------------------------
// bin i covers [floor(i*H/s), ceil((i+1)*H/s)) (ATen start_index/end_index); bins overlap when s does not divide H.
__device__ __forceinline__ int bin_start(int i, int in, int s) { return (int)(((long long)i * in) / s); }
__device__ __forceinline__ int bin_end(int i, int in, int s) { return (int)((((long long)(i + 1)) * in + s - 1) / s); }

#define AP_TX 64
#define AP_TY 4
__global__ __launch_bounds__(AP_TX * AP_TY) void adaptive_avgpool_fwd_kernel(const float* __restrict__ x,
                                                                            float* __restrict__ y, int n, int h, int w,
                                                                            int c, int s) {
    __shared__ float red[AP_TY][AP_TX * 4];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int bin = blockIdx.y;
    const int img = blockIdx.z;
    const int by = bin / s, bx = bin - by * s;
    const int y0 = bin_start(by, h, s), y1 = bin_end(by, h, s);
    const int x0 = bin_start(bx, w, s), x1 = bin_end(bx, w, s);
    const int bw = x1 - x0;
    const int cnt = (y1 - y0) * bw;
    const int c0 = (blockIdx.x * AP_TX + tx) * 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < c) {
        for (int p = ty; p < cnt; p += AP_TY) {
            const int py = y0 + p / bw, px = x0 + p % bw;
            const float* src = x + (((size_t)img * h + py) * w + px) * c + c0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < c) a[e] += src[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ty][tx * 4 + e] = a[e];
    __syncthreads();
    if (ty == 0 && c0 < c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c0 + e < c) {
                float v = red[0][tx * 4 + e];
#pragma unroll
                for (int j = 1; j < AP_TY; ++j) v += red[j][tx * 4 + e];
                y[(((size_t)img * s + by) * s + bx) * c + c0 + e] = v / (float)cnt;
            }
        }
    }
}

// Fused forward of the PPM pyramid (all scales, ONE pass over x).  Stage 1: per (image, image row, 256-channel chunk)
// the row's sum over every x-bin of every scale -> rowpart[img][row][slot][c] (slot = x-bin index, scales concatenated;
// thread.y = scale, so the row is read once from HBM and re-read from L1 by the other scales).  Stage 2: sum the rows of
// every y-bin and divide by the bin population.  n*h*(c/256) workgroups instead of (c/256)*s*s*n, which for s = 1..3 left
// most of the chip idle.
struct PoolFwdMulti {
    float* y[4];
    int s[4];
    int slot0[4];  // first x-bin slot of scale k
    int ns, nslots;
};
__global__ __launch_bounds__(256) void pyramid_pool_rows_kernel(const float* __restrict__ x, float* __restrict__ rowpart,
                                                                PoolFwdMulti p, int h, int w, int c) {
    const int tx = threadIdx.x, k = threadIdx.y;
    const int row = blockIdx.y, img = blockIdx.z;
    const int c0 = (blockIdx.x * 64 + tx) * 4;
    if (k >= p.ns || c0 >= c) return;
    const int s = p.s[k];
    const float* src = x + (((size_t)img * h + row) * w) * c + c0;
    float* dst = rowpart + (((size_t)img * h + row) * p.nslots + p.slot0[k]) * c + c0;
    for (int bx = 0; bx < s; ++bx) {
        const int x0 = bin_start(bx, w, s), x1 = bin_end(bx, w, s);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int px = x0; px < x1; ++px) a += *reinterpret_cast<const f32x4*>(src + (size_t)px * c);
        *reinterpret_cast<f32x4*>(dst + (size_t)bx * c) = a;
    }
}
__global__ __launch_bounds__(256) void pyramid_pool_bins_kernel(const float* __restrict__ rowpart, PoolFwdMulti p, int n,
                                                                int h, int w, int c) {
    // one thread per (image, scale, by, bx, float4 of channels)
    const int c4 = c >> 2;
    int nbins = 0;
    for (int k = 0; k < p.ns; ++k) nbins += p.s[k] * p.s[k];
    const long long total = (long long)n * nbins * c4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int ch = (int)(i % c4) * 4;
        long long r = i / c4;
        int bin = (int)(r % nbins);
        const int img = (int)(r / nbins);
        int k = 0;
        while (bin >= p.s[k] * p.s[k]) {
            bin -= p.s[k] * p.s[k];
            ++k;
        }
        const int s = p.s[k];
        const int by = bin / s, bx = bin - by * s;
        const int y0 = bin_start(by, h, s), y1 = bin_end(by, h, s);
        const int x0 = bin_start(bx, w, s), x1 = bin_end(bx, w, s);
        const float* src = rowpart + (((size_t)img * h) * p.nslots + p.slot0[k] + bx) * c + ch;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int yy = y0; yy < y1; ++yy) a += *reinterpret_cast<const f32x4*>(src + (size_t)yy * p.nslots * c);
        const float cnt = (float)((y1 - y0) * (x1 - x0));
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = a[e] / cnt;
        *reinterpret_cast<f32x4*>(p.y[k] + (((size_t)img * s + by) * s + bx) * c + ch) = o;
    }
}

__global__ __launch_bounds__(256) void adaptive_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                   int n, int h, int w, int c, int s, int accumulate) {
    const long long total = (long long)n * h * w * c;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int ch = (int)(i % c);
        long long r = i / c;
        const int ix = (int)(r % w);
        r /= w;
        const int iy = (int)(r % h);
        const int img = (int)(r / h);
        float g = 0.f;
        // candidate bins: those whose [start,end) contains the coordinate (at most 2 per axis)
        int byc = (int)(((long long)iy * s) / h);
        int bxc = (int)(((long long)ix * s) / w);
        for (int by = max(0, byc - 1); by <= min(s - 1, byc + 1); ++by) {
            const int y0 = bin_start(by, h, s), y1 = bin_end(by, h, s);
            if (iy < y0 || iy >= y1) continue;
            for (int bx = max(0, bxc - 1); bx <= min(s - 1, bxc + 1); ++bx) {
                const int x0 = bin_start(bx, w, s), x1 = bin_end(bx, w, s);
                if (ix < x0 || ix >= x1) continue;
                const float cnt = (float)((y1 - y0) * (x1 - x0));
                g += dy[(((size_t)img * s + by) * s + bx) * c + ch] / cnt;
            }
        }
        dx[i] = accumulate ? dx[i] + g : g;
    }
}

// Fused adjoint of the whole pyramid: dx = sum over up to 4 scales of the adaptive-pool adjoint, optionally composed with
// the temporal-mean adjoint (T > 1: dy_i is the gradient of the BLENDED [B][s][s][c] tensor and every frame t of clip b
// receives dy_i[b]/T).  One pass over dx instead of one per scale.
struct PoolBwdMulti {
    const float* dy[4];
    int s[4];
    int ns;
};

#define POOL_BWD_XCHUNK 8
// One workgroup per (image, row, 8-pixel column chunk): the scale / row-bin / column-bin loops are workgroup-uniform (scalar registers, the
// integer divisions of the bin edges are done once per row or per pixel, not per element); the 256 threads sweep the
// channel float4s of one pixel at a time.  The pooled gradients are a few hundred KB (L2-resident): the kernel is a
// pure streaming write of dx.  Same summation order as before: scales in order, row bins, then column bins.
__global__ __launch_bounds__(256) void adaptive_avgpool_bwd_multi_kernel(PoolBwdMulti p, float* __restrict__ dx, int n,
                                                                         int h, int w, int c, int T, int B) {
    // Tap lists (which pooled cells reach a pixel, with which weight) are built ONCE per workgroup, one pixel of the
    // chunk per thread, into LDS: the bin-edge integer divisions used to be evaluated by every thread for every element
    // (measured: 423 us for 295 MB, 0.7 TB/s - instruction-bound, not memory-bound).  Summation order unchanged:
    // scales in order, row bins, then column bins.
    __shared__ int s_nb[POOL_BWD_XCHUNK];
    __shared__ int s_off[POOL_BWD_XCHUNK][16];
    __shared__ int s_k[POOL_BWD_XCHUNK][16];
    __shared__ float s_wt[POOL_BWD_XCHUNK][16];
    const int cw = c / 4;
    const int iy = blockIdx.x % h;
    const int img = blockIdx.x / h;
    const int src = (T > 1) ? (img % B) : img;
    const float invT = 1.f / (float)T;
    f32x4* drow = reinterpret_cast<f32x4*>(dx) + ((size_t)img * h + iy) * w * cw;
    const int x_lo = blockIdx.y * POOL_BWD_XCHUNK, x_hi = min(w, x_lo + POOL_BWD_XCHUNK);
    if ((int)threadIdx.x < x_hi - x_lo) {
        const int ix = x_lo + threadIdx.x;
        int nb = 0;
        for (int k = 0; k < p.ns; ++k) {
            const int s = p.s[k];
            const int byc = (iy * s) / h;
            const int bxc = (ix * s) / w;
            for (int by = max(0, byc - 1); by <= min(s - 1, byc + 1); ++by) {
                const int y0 = bin_start(by, h, s), y1 = bin_end(by, h, s);
                if (iy < y0 || iy >= y1) continue;
                for (int bx = max(0, bxc - 1); bx <= min(s - 1, bxc + 1); ++bx) {
                    const int x0 = bin_start(bx, w, s), x1 = bin_end(bx, w, s);
                    if (ix < x0 || ix >= x1) continue;
                    if (nb < 16) {
                        s_off[threadIdx.x][nb] = ((src * s + by) * s + bx) * cw;
                        s_k[threadIdx.x][nb] = k;
                        s_wt[threadIdx.x][nb] = 1.f / (float)((y1 - y0) * (x1 - x0));
                        ++nb;
                    }
                }
            }
        }
        s_nb[threadIdx.x] = nb;
    }
    __syncthreads();
    for (int ix = x_lo; ix < x_hi; ++ix) {
        const int q = ix - x_lo;
        const int nb = s_nb[q];
        for (int ch = threadIdx.x; ch < cw; ch += blockDim.x) {
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < nb; ++j)
                g += s_wt[q][j] * reinterpret_cast<const f32x4*>(p.dy[s_k[q][j]])[s_off[q][j] + ch];
            if (T > 1) g *= invT;
            drow[(size_t)ix * cw + ch] = g;
        }
    }
}

// ---- temporal mean ---------------------------------------------------------------------------------
// x is [T][B][inner] (frames stacked frame-major along the batch, as torch.cat(clip_imgs) makes them).
// The reference concatenates [current, others...] and takes torch.mean over that axis; fp32 summation order
// here is t = T-1 (current frame, the LAST chunk of the batch) first, then t = 0..T-2, as the reference's cat.
__global__ __launch_bounds__(256) void temporal_mean_fwd_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ wts, float* __restrict__ y,
                                                                int T, int B, long long inner) {
    const long long total = (long long)B * inner;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float invT = 1.f / (float)T;
    for (; i < total; i += stride) {
        const long long b = i / inner;
        float a = 0.f;
        for (int j = 0; j < T; ++j) {
            const int t = (j == 0) ? (T - 1) : (j - 1);
            float v = x[(size_t)t * total + i];
            if (wts) v *= wts[b * T + j];
            a += v;
        }
        y[i] = a * invT;
    }
}

__global__ __launch_bounds__(256) void temporal_mean_bwd_kernel(const float* __restrict__ dy,
                                                                const float* __restrict__ wts, float* __restrict__ dx,
                                                                int T, int B, long long inner) {
    const long long total = (long long)B * inner;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float invT = 1.f / (float)T;
    for (; i < total; i += stride) {
        const long long b = i / inner;
        const float g = dy[i] * invT;
        for (int j = 0; j < T; ++j) {
            const int t = (j == 0) ? (T - 1) : (j - 1);
            dx[(size_t)t * total + i] = wts ? g * wts[b * T + j] : g;
        }
    }
}

extern "C" int vspw_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, int oh,
                                     int ow, void* stream) {
    if (!x || !y || !idx || n <= 0 || h <= 0 || w <= 0 || c <= 0) return VSPW_EINVAL;
    if (oh != (h + 2 - 3) / 2 + 1 || ow != (w + 2 - 3) / 2 + 1) return VSPW_EINVAL;
    long long total = (long long)n * oh * ow * c;
    if (c % 4 == 0 && n <= 65535 && (long long)h * w * c < 0x7fffffffLL) {
        hipLaunchKernelGGL(maxpool_fwd4_kernel, dim3(vspw_stream_grid((long long)oh * ow * (c / 4), 256), n), dim3(256), 0,
                           vspw_stream(stream), x, y, idx, h, w, c / 4, oh, ow);
        return vspw_launch_status();
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream), x, y,
                       idx, n, h, w, c, oh, ow);
    return vspw_launch_status();
}

extern "C" int vspw_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int w, int c,
                                     int oh, int ow, void* stream) {
    if (!dy || !dx || !idx || n <= 0 || h <= 0 || w <= 0 || c <= 0) return VSPW_EINVAL;
    if (oh != (h + 2 - 3) / 2 + 1 || ow != (w + 2 - 3) / 2 + 1) return VSPW_EINVAL;
    long long total = (long long)n * h * w * c;
    if (c % 4 == 0 && n <= 65535 && (long long)h * w * c < 0x7fffffffLL) {
        hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(vspw_stream_grid((long long)h * w * (c / 4), 256), n), dim3(256), 0,
                           vspw_stream(stream), dy, idx, dx, h, w, c / 4, oh, ow);
        return vspw_launch_status();
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream), dy,
                       idx, dx, n, h, w, c, oh, ow);
    return vspw_launch_status();
}

extern "C" int vspw_adaptive_avgpool_fwd(const float* x, float* y, int n, int h, int w, int c, int s, void* stream) {
    if (!x || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0 || s <= 0 || n > 65535) return VSPW_EINVAL;
    dim3 grid(vspw_cdiv(c, AP_TX * 4), s * s, n);
    hipLaunchKernelGGL(adaptive_avgpool_fwd_kernel, grid, dim3(AP_TX, AP_TY), 0, vspw_stream(stream), x, y, n, h, w, c,
                       s);
    return vspw_launch_status();
}

extern "C" size_t vspw_pyramid_pool_fwd_workspace(const int* scales, int nscales, int n, int h, int c) {
    if (!scales || nscales <= 0 || nscales > 4 || n <= 0 || h <= 0 || c <= 0) return 0;
    long long slots = 0;
    for (int k = 0; k < nscales; ++k) slots += scales[k];
    return (size_t)n * h * slots * c * sizeof(float);
}

extern "C" int vspw_pyramid_pool_fwd(const float* x, const int* scales, int nscales, float* const* y, int n, int h,
                                     int w, int c, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !scales || !y || nscales <= 0 || nscales > 4 || n <= 0 || n > 65535 || h <= 0 || h > 65535 || w <= 0 ||
        c <= 0 || (c & 3))
        return VSPW_EINVAL;
    PoolFwdMulti p;
    p.ns = nscales;
    p.nslots = 0;
    for (int k = 0; k < 4; ++k) {
        p.y[k] = k < nscales ? y[k] : nullptr;
        p.s[k] = k < nscales ? scales[k] : 1;
        p.slot0[k] = p.nslots;
        if (k < nscales) {
            if (!y[k] || scales[k] <= 0) return VSPW_EINVAL;
            p.nslots += scales[k];
        }
    }
    if (!ws || ws_bytes < (size_t)n * h * p.nslots * c * sizeof(float)) return VSPW_EINVAL;
    float* rowpart = reinterpret_cast<float*>(ws);
    hipStream_t st = vspw_stream(stream);
    hipLaunchKernelGGL(pyramid_pool_rows_kernel, dim3(vspw_cdiv(c, 256), h, n), dim3(64, 4), 0, st, x, rowpart, p, h, w,
                       c);
    long long nbins = 0;
    for (int k = 0; k < nscales; ++k) nbins += (long long)scales[k] * scales[k];
    hipLaunchKernelGGL(pyramid_pool_bins_kernel, dim3(vspw_stream_grid((long long)n * nbins * (c / 4), 256)), dim3(256), 0,
                       st, rowpart, p, n, h, w, c);
    return vspw_launch_status();
}

extern "C" int vspw_adaptive_avgpool_bwd(const float* dy, float* dx, int n, int h, int w, int c, int s,
                                         int accumulate, void* stream) {
    if (!dy || !dx || n <= 0 || h <= 0 || w <= 0 || c <= 0 || s <= 0) return VSPW_EINVAL;
    long long total = (long long)n * h * w * c;
    hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0,
                       vspw_stream(stream), dy, dx, n, h, w, c, s, accumulate);
    return vspw_launch_status();
}

extern "C" int vspw_pyramid_pool_bwd(const float* const* dy, const int* scales, int nscales, float* dx, int n, int h,
                                     int w, int c, int T, void* stream) {
    if (!dy || !scales || !dx || nscales <= 0 || nscales > 4 || n <= 0 || h <= 0 || w <= 0 || c <= 0 || T <= 0)
        return VSPW_EINVAL;
    if (c % 4 != 0 || n % T != 0) return VSPW_EINVAL;
    PoolBwdMulti p;
    p.ns = nscales;
    for (int k = 0; k < 4; ++k) {
        p.dy[k] = k < nscales ? dy[k] : nullptr;
        p.s[k] = k < nscales ? scales[k] : 1;
        if (k < nscales && (!dy[k] || scales[k] <= 0)) return VSPW_EINVAL;
    }
    if ((long long)n * h > 0x7fffffffLL) return VSPW_EINVAL;
    hipLaunchKernelGGL(adaptive_avgpool_bwd_multi_kernel, dim3((unsigned)(n * h), vspw_cdiv(w, POOL_BWD_XCHUNK)), dim3(256),
                       0, vspw_stream(stream), p, dx, n, h, w, c, T, n / T);
    return vspw_launch_status();
}

// dwts[b][j] = (1/T) * sum_i dy[b][i] * x[t(j)][b][i]   (gradient of the psp_weight temporal softmax weights)
__global__ __launch_bounds__(256) void temporal_mean_wgrad_kernel(const float* __restrict__ dy,
                                                                  const float* __restrict__ x,
                                                                  float* __restrict__ dw, int T, int B, long long inner,
                                                                  int accumulate) {
    __shared__ double red[4];
    const int b = blockIdx.x / T, j = blockIdx.x % T;
    const int t = (j == 0) ? (T - 1) : (j - 1);
    const float* px = x + ((size_t)t * B + b) * inner;
    const float* pg = dy + (size_t)b * inner;
    double s = 0;
    for (long long i = threadIdx.x; i < inner; i += blockDim.x) s += (double)pg[i] * (double)px[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)((red[0] + red[1] + red[2] + red[3]) / (double)T);
        dw[blockIdx.x] = accumulate ? dw[blockIdx.x] + v : v;
    }
}

extern "C" int vspw_temporal_mean_wgrad(const float* dy, const float* x, float* dw, int T, int B, long long inner,
                                        int accumulate, void* stream) {
    if (!dy || !x || !dw || T <= 0 || B <= 0 || inner <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(temporal_mean_wgrad_kernel, dim3(B * T), dim3(256), 0, vspw_stream(stream), dy, x, dw, T, B,
                       inner, accumulate);
    return vspw_launch_status();
}

extern "C" int vspw_temporal_mean_fwd(const float* x, const float* wts, float* y, int T, int B, long long inner,
                                      void* stream) {
    if (!x || !y || T <= 0 || B <= 0 || inner <= 0) return VSPW_EINVAL;
    long long total = (long long)B * inner;
    hipLaunchKernelGGL(temporal_mean_fwd_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream),
                       x, wts, y, T, B, inner);
    return vspw_launch_status();
}

extern "C" int vspw_temporal_mean_bwd(const float* dy, const float* wts, float* dx, int T, int B, long long inner,
                                      void* stream) {
    if (!dy || !dx || T <= 0 || B <= 0 || inner <= 0) return VSPW_EINVAL;
    long long total = (long long)B * inner;
    hipLaunchKernelGGL(temporal_mean_bwd_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream),
                       dy, wts, dx, T, B, inner);
    return vspw_launch_status();
}

// ---- 2x2 average pool, NHWC ------------------------------------------------------------------------
// F.avg_pool2d(emb, (2,2)) of the non-local decoders' `downsample` switch (models/non_local_models.py:30-32,136-137):
// stride 2, no padding, floor output size (an odd trailing row / column is dropped); ATen sums the window row-major
// and divides by 4.  One thread per 4 channels; the adjoint is a gather (each input pixel belongs to one window).
__global__ __launch_bounds__(256) void avgpool2x2_nhwc_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  int n, int h, int w, int c4, int oh, int ow) {
    const long long total = (long long)n * oh * ow * c4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* __restrict__ y4 = reinterpret_cast<f32x4*>(y);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)(i % c4);
        long long r = i / c4;
        const int ox = (int)(r % ow);
        r /= ow;
        const int oy = (int)(r % oh);
        const int img = (int)(r / oh);
        const size_t base = (((size_t)img * h + 2 * oy) * w + 2 * ox) * c4 + ch;
        f32x4 acc = x4[base];
        acc += x4[base + c4];
        acc += x4[base + (size_t)w * c4];
        acc += x4[base + (size_t)w * c4 + c4];
        y4[i] = acc * 0.25f;
    }
}

__global__ __launch_bounds__(256) void avgpool2x2_nhwc_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                  int n, int h, int w, int c4, int oh, int ow) {
    const long long total = (long long)n * h * w * c4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(dy);
    f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dx);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)(i % c4);
        long long r = i / c4;
        const int ix = (int)(r % w);
        r /= w;
        const int iy = (int)(r % h);
        const int img = (int)(r / h);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((iy >> 1) < oh && (ix >> 1) < ow)
            v = g4[(((size_t)img * oh + (iy >> 1)) * ow + (ix >> 1)) * c4 + ch] * 0.25f;
        d4[i] = v;
    }
}

extern "C" int vspw_avgpool2x2_nhwc_fwd(const float* x, float* y, int n, int h, int w, int c, void* stream) {
    if (!x || !y || n <= 0 || h < 2 || w < 2 || c <= 0 || (c & 3)) return VSPW_EINVAL;
    const int oh = h / 2, ow = w / 2;
    const long long total = (long long)n * oh * ow * (c / 4);
    hipLaunchKernelGGL(avgpool2x2_nhwc_fwd_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream),
                       x, y, n, h, w, c / 4, oh, ow);
    return vspw_launch_status();
}

extern "C" int vspw_avgpool2x2_nhwc_bwd(const float* dy, float* dx, int n, int h, int w, int c, void* stream) {
    if (!dy || !dx || n <= 0 || h < 2 || w < 2 || c <= 0 || (c & 3)) return VSPW_EINVAL;
    const int oh = h / 2, ow = w / 2;
    const long long total = (long long)n * h * w * (c / 4);
    hipLaunchKernelGGL(avgpool2x2_nhwc_bwd_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream),
                       dy, dx, n, h, w, c / 4, oh, ow);
    return vspw_launch_status();
}
