// Softmax family and the fused segmentation loss / inference head, NHWC fp32.
//
//   softmax_lastdim : F.log_softmax(dim=1) at feature resolution (models/clip_psp.py:198,212; models/models.py:930,990)
//                     and the object-attention softmax(dim=-1) (models/ocr_modules/spatial_ocr_block.py:268-270)
//   softmax_pixels  : F.softmax(probs, dim=2) over H*W (spatial_ocr_block.py:60,104)
//   seg_nll_*       : F.interpolate(log-probs -> label size, bilinear) + nn.NLLLoss(ignore_index=255) + pixel_acc
//                     (models/clip_psp.py:198-216, 92-98; models/models.py:92-107, 65-71) without ever materialising
//                     the [N,K,H,W] up-sampled tensor (228 MB per 2 frames at 479x479, K=124)
//   upsample_softmax: inference head, interpolate logits to segSize then softmax (models/clip_psp.py:190-194)
// In NHWC the class vector of a pixel is contiguous, so one wavefront owns one pixel and reduces over K with
// cross-lane shuffles.
#include "common.h"

__global__ __launch_bounds__(256) void softmax_lastdim_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  long long rows, int k, float alpha, int logm) {
    const int lane = threadIdx.x & 63;
    long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; row < rows; row += stride) {
        const float* px = x + (size_t)row * k;
        float m = -INFINITY;
        for (int j = lane; j < k; j += 64) m = fmaxf(m, alpha * px[j]);
        m = wave_max(m);
        float s = 0.f;
        for (int j = lane; j < k; j += 64) s += expf(alpha * px[j] - m);
        s = wave_sum(s);
        float* py = y + (size_t)row * k;
        if (logm) {
            const float ls = logf(s);
            for (int j = lane; j < k; j += 64) py[j] = alpha * px[j] - m - ls;
        } else {
            const float inv = 1.f / s;
            for (int j = lane; j < k; j += 64) py[j] = expf(alpha * px[j] - m) * inv;
        }
    }
}

__global__ __launch_bounds__(256) void softmax_lastdim_bwd_kernel(const float* __restrict__ dy,
                                                                  const float* __restrict__ y, float* __restrict__ dx,
                                                                  long long rows, int k, float alpha, int logm) {
    const int lane = threadIdx.x & 63;
    long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; row < rows; row += stride) {
        const float* pg = dy + (size_t)row * k;
        const float* py = y + (size_t)row * k;
        float* pd = dx + (size_t)row * k;
        float s = 0.f;
        if (logm) {
            for (int j = lane; j < k; j += 64) s += pg[j];
            s = wave_sum(s);
            for (int j = lane; j < k; j += 64) pd[j] = alpha * (pg[j] - expf(py[j]) * s);
        } else {
            for (int j = lane; j < k; j += 64) s += pg[j] * py[j];
            s = wave_sum(s);
            for (int j = lane; j < k; j += 64) pd[j] = alpha * py[j] * (pg[j] - s);
        }
    }
}

// softmax over the pixels of each (sample, class) column of a [B][HW][K] tensor: 16 classes (64 bytes) x 64 row groups
// per workgroup - the tensors are small (3 600 x 124 per sample) and the launch is latency-bound: 4 x 10 workgroups of
// 8 row groups took 333 us forward / 260 us backward on the OCR bench step
#define SP_TX 16
#define SP_TY 64
__global__ __launch_bounds__(SP_TX * SP_TY) void softmax_pixels_fwd_kernel(const float* __restrict__ x,
                                                                          float* __restrict__ y, int hw, int k,
                                                                          float alpha) {
    __shared__ float red[SP_TY][SP_TX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int ch = blockIdx.x * SP_TX + tx;
    const bool ok = ch < k;
    const float* px = x + (size_t)blockIdx.y * hw * k;
    float* py = y + (size_t)blockIdx.y * hw * k;
    float m = -INFINITY;
    if (ok)
        for (int r = ty; r < hw; r += SP_TY) m = fmaxf(m, alpha * px[(size_t)r * k + ch]);
    red[ty][tx] = m;
    __syncthreads();
    m = red[0][tx];
#pragma unroll
    for (int j = 1; j < SP_TY; ++j) m = fmaxf(m, red[j][tx]);
    __syncthreads();
    float s = 0.f;
    if (ok)
        for (int r = ty; r < hw; r += SP_TY) s += expf(alpha * px[(size_t)r * k + ch] - m);
    red[ty][tx] = s;
    __syncthreads();
    s = red[0][tx];
#pragma unroll
    for (int j = 1; j < SP_TY; ++j) s += red[j][tx];
    const float inv = 1.f / s;
    if (ok)
        for (int r = ty; r < hw; r += SP_TY) py[(size_t)r * k + ch] = expf(alpha * px[(size_t)r * k + ch] - m) * inv;
}

__global__ __launch_bounds__(SP_TX * SP_TY) void softmax_pixels_bwd_kernel(const float* __restrict__ dy,
                                                                          const float* __restrict__ y,
                                                                          float* __restrict__ dx, int hw, int k,
                                                                          float alpha) {
    __shared__ float red[SP_TY][SP_TX];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int ch = blockIdx.x * SP_TX + tx;
    const bool ok = ch < k;
    const size_t base = (size_t)blockIdx.y * hw * k;
    float s = 0.f;
    if (ok)
        for (int r = ty; r < hw; r += SP_TY) s += dy[base + (size_t)r * k + ch] * y[base + (size_t)r * k + ch];
    red[ty][tx] = s;
    __syncthreads();
    s = red[0][tx];
#pragma unroll
    for (int j = 1; j < SP_TY; ++j) s += red[j][tx];
    if (ok)
        for (int r = ty; r < hw; r += SP_TY) {
            const size_t o = base + (size_t)r * k + ch;
            dx[o] = alpha * y[o] * (dy[o] - s);
        }
}

// ---- fused interpolate + NLL + pixel accuracy ----------------------------------------------------------
// One wavefront per output pixel when the arg-max is wanted (lanes stride over K); one thread per pixel otherwise.
template <typename LT>
__global__ __launch_bounds__(256) void seg_nll_fwd_acc_kernel(const float* __restrict__ logp,
                                                              const LT* __restrict__ label,
                                                              double* __restrict__ out, int n, int h, int w, int k,
                                                              int H, int W, int ignore, float sy, float sx) {
    const int lane = threadIdx.x & 63;
    const long long total = (long long)n * H * W;
    long long pix = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    double loss = 0, cnt = 0, acc = 0, valid = 0;
    for (; pix < total; pix += stride) {
        const int ox = (int)(pix % W);
        long long r = pix / W;
        const int oy = (int)(r % H);
        const int img = (int)(r / H);
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(oy, sy, h, y0, y1, ly);
        bilinear_src(ox, sx, w, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* p00 = logp + (((size_t)img * h + y0) * w + x0) * k;
        const float* p01 = logp + (((size_t)img * h + y0) * w + x1) * k;
        const float* p10 = logp + (((size_t)img * h + y1) * w + x0) * k;
        const float* p11 = logp + (((size_t)img * h + y1) * w + x1) * k;
        const long long lab = (long long)label[pix];  // float labels: label.long() of the reference (truncation)
        float best = -INFINITY;
        int bi = 0x7fffffff;
        float vlab = 0.f;
        for (int j = lane; j < k; j += 64) {
            const float v = hy * (hx * p00[j] + lx * p01[j]) + ly * (hx * p10[j] + lx * p11[j]);
            if (v > best) {
                best = v;
                bi = j;
            }
            if ((long long)j == lab) vlab = v;
        }
        // wave arg-max, lowest index wins ties (torch.max returns the first maximal index)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) {
                best = ob;
                bi = oi;
            }
        }
        vlab = wave_sum(vlab);
        if (lane == 0) {
            if (lab >= 0) {
                valid += 1.0;
                if ((long long)bi == lab) acc += 1.0;
            }
            if (lab != (long long)ignore && lab >= 0 && lab < k) {
                loss -= (double)vlab;
                cnt += 1.0;
            }
        }
    }
    // block reduce (lane 0 of each wave holds the partials)
    __shared__ double red[4][4];
    const int wv = threadIdx.x >> 6;
    if (lane == 0) {
        red[wv][0] = loss;
        red[wv][1] = cnt;
        red[wv][2] = acc;
        red[wv][3] = valid;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double a = 0;
        for (int j = 0; j < (int)(blockDim.x >> 6); ++j) a += red[j][threadIdx.x];
        if (threadIdx.x == 0) a = rint(a * VSPW_NLL_FIXED);  // integer-valued: the atomic sum is order-independent
        if (a != 0) atomicAdd(&out[threadIdx.x], a);
    }
}

template <typename LT>
__global__ __launch_bounds__(256) void seg_nll_fwd_kernel(const float* __restrict__ logp,
                                                          const LT* __restrict__ label, double* __restrict__ out,
                                                          int n, int h, int w, int k, int H, int W, int ignore,
                                                          float sy, float sx) {
    const long long total = (long long)n * H * W;
    long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    double loss = 0, cnt = 0, valid = 0;
    for (; pix < total; pix += stride) {
        const long long lab = (long long)label[pix];  // float labels: label.long() of the reference (truncation)
        if (lab >= 0) valid += 1.0;
        if (lab == (long long)ignore || lab < 0 || lab >= k) continue;
        const int ox = (int)(pix % W);
        long long r = pix / W;
        const int oy = (int)(r % H);
        const int img = (int)(r / H);
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(oy, sy, h, y0, y1, ly);
        bilinear_src(ox, sx, w, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float a = logp[(((size_t)img * h + y0) * w + x0) * k + lab];
        const float b = logp[(((size_t)img * h + y0) * w + x1) * k + lab];
        const float c = logp[(((size_t)img * h + y1) * w + x0) * k + lab];
        const float d = logp[(((size_t)img * h + y1) * w + x1) * k + lab];
        loss -= (double)(hy * (hx * a + lx * b) + ly * (hx * c + lx * d));
        cnt += 1.0;
    }
    loss = wave_sum_d(loss);
    cnt = wave_sum_d(cnt);
    valid = wave_sum_d(valid);
    __shared__ double red[4][3];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[wv][0] = loss;
        red[wv][1] = cnt;
        red[wv][2] = valid;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a = 0;
        for (int j = 0; j < (int)(blockDim.x >> 6); ++j) a += red[j][threadIdx.x];
        const int slot = threadIdx.x == 2 ? 3 : threadIdx.x;
        if (threadIdx.x == 0) a = rint(a * VSPW_NLL_FIXED);  // integer-valued: the atomic sum is order-independent
        if (a != 0) atomicAdd(&out[slot], a);
    }
}

// One workgroup per source (feature-resolution) pixel: gather the bilinear adjoint of the per-pixel NLL gradient
// into K class bins (64-bit fixed-point LDS atomics: order-independent, bit-reproducible), then apply the log-softmax Jacobian and write dlogits[K].
#define NLL_MAXK 1024
template <typename LT>
__global__ __launch_bounds__(256) void seg_nll_bwd_kernel(const float* __restrict__ logp,
                                                          const LT* __restrict__ label,
                                                          const double* __restrict__ fwd_out,
                                                          const float* __restrict__ gscale,
                                                          float* __restrict__ dlogits, int n, int h, int w, int k,
                                                          int H, int W, int ignore, float sy, float sx, int jacobian) {
    __shared__ float bins[NLL_MAXK];
    __shared__ unsigned long long ibins[NLL_MAXK];
    __shared__ float wsum[4];
    const int tid = threadIdx.x;
    const long long sp = blockIdx.x;
    const int ix = (int)(sp % w);
    long long r = sp / w;
    const int iy = (int)(r % h);
    const int img = (int)(r / h);
    for (int j = tid; j < k; j += blockDim.x) ibins[j] = 0ull;
    __syncthreads();
    const float ry = (float)H / (float)h, rx = (float)W / (float)w;
    int oy_lo = (int)floorf(((float)iy - 1.f + 0.5f) * ry - 0.5f) - 1;
    int oy_hi = (int)ceilf(((float)iy + 1.f + 0.5f) * ry - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 1.f + 0.5f) * rx - 0.5f) - 1;
    int ox_hi = (int)ceilf(((float)ix + 1.f + 0.5f) * rx - 0.5f) + 1;
    if (iy == 0) oy_lo = 0;
    if (iy == h - 1) oy_hi = H - 1;
    if (ix == 0) ox_lo = 0;
    if (ix == w - 1) ox_hi = W - 1;
    oy_lo = max(oy_lo, 0);
    ox_lo = max(ox_lo, 0);
    oy_hi = min(oy_hi, H - 1);
    ox_hi = min(ox_hi, W - 1);
    const int fw = ox_hi - ox_lo + 1, fh = oy_hi - oy_lo + 1;
    // Class bins are accumulated with 64-bit INTEGER LDS atomics on weights scaled by 2^40: integer addition is
    // associative, so the result does not depend on the order the lanes arrive in (bit-reproducible), and the
    // bilinear weights (products of two fp32 fractions, >= 2^-16 apart from exact zeros) are represented exactly up to
    // 2^-40 - the bin sums are MORE accurate than an fp32 accumulation.  A footprint has < 2^12 pixels: no overflow.
    for (int q = tid; q < fw * fh; q += blockDim.x) {
        const int oy = oy_lo + q / fw, ox = ox_lo + q % fw;
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(oy, sy, h, y0, y1, ly);
        float wy = 0.f;
        if (y0 == iy) wy += 1.f - ly;
        if (y1 == iy) wy += ly;
        if (wy == 0.f) continue;
        bilinear_src(ox, sx, w, x0, x1, lx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        if (wx == 0.f) continue;
        const long long lab = (long long)label[((size_t)img * H + oy) * W + ox];
        if (lab == (long long)ignore || lab < 0 || lab >= k) continue;
        atomicAdd(&ibins[(int)lab], (unsigned long long)((double)(wy * wx) * 1099511627776.0 + 0.5));
    }
    __syncthreads();
    for (int j = tid; j < k; j += blockDim.x) bins[j] = (float)((double)ibins[j] * (1.0 / 1099511627776.0));
    __syncthreads();
    const double cnt = fwd_out[1];
    const float g = cnt > 0 ? -gscale[0] / (float)cnt : 0.f;
    float s = 0.f;
    for (int j = tid; j < k; j += blockDim.x) s += bins[j];
    s = wave_sum(s);
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    s = 0.f;
    for (int j = 0; j < (int)(blockDim.x >> 6); ++j) s += wsum[j];
    s *= g;  // sum_k dlogp_k
    const size_t base = (size_t)sp * k;
    if (jacobian) {
        for (int j = tid; j < k; j += blockDim.x) dlogits[base + j] = g * bins[j] - expf(logp[base + j]) * s;
    } else {
        for (int j = tid; j < k; j += blockDim.x) dlogits[base + j] = g * bins[j];
    }
}

__global__ __launch_bounds__(256) void upsample_softmax_kernel(const float* __restrict__ logits,
                                                               float* __restrict__ probs, int n, int h, int w, int k,
                                                               int H, int W, float sy, float sx) {
    const int lane = threadIdx.x & 63;
    const long long total = (long long)n * H * W;
    long long pix = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; pix < total; pix += stride) {
        const int ox = (int)(pix % W);
        long long r = pix / W;
        const int oy = (int)(r % H);
        const int img = (int)(r / H);
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(oy, sy, h, y0, y1, ly);
        bilinear_src(ox, sx, w, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* p00 = logits + (((size_t)img * h + y0) * w + x0) * k;
        const float* p01 = logits + (((size_t)img * h + y0) * w + x1) * k;
        const float* p10 = logits + (((size_t)img * h + y1) * w + x0) * k;
        const float* p11 = logits + (((size_t)img * h + y1) * w + x1) * k;
        float* dst = probs + (size_t)pix * k;
        if (k <= 256) {
            // interpolate once, keep the (at most 4) values of this lane in registers: the three softmax passes then
            // see bit-identical inputs (re-evaluating the blend may contract differently and, for huge logits, break
            // max >= v)
            float vals[4];
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = lane + 64 * t;
                vals[t] = (j < k) ? hy * (hx * p00[j] + lx * p01[j]) + ly * (hx * p10[j] + lx * p11[j]) : -INFINITY;
                m = fmaxf(m, vals[t]);
            }
            m = wave_max(m);
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                vals[t] = (lane + 64 * t < k) ? expf(vals[t] - m) : 0.f;
                s += vals[t];
            }
            s = wave_sum(s);
            const float inv = 1.f / s;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (lane + 64 * t < k) dst[lane + 64 * t] = vals[t] * inv;
        } else {
            float m = -INFINITY;
            for (int j = lane; j < k; j += 64)
                m = fmaxf(m, hy * (hx * p00[j] + lx * p01[j]) + ly * (hx * p10[j] + lx * p11[j]));
            m = wave_max(m);
            float s = 0.f;
            for (int j = lane; j < k; j += 64)
                s += expf(fminf(hy * (hx * p00[j] + lx * p01[j]) + ly * (hx * p10[j] + lx * p11[j]) - m, 0.f));
            s = wave_sum(s);
            const float inv = 1.f / s;
            for (int j = lane; j < k; j += 64)
                dst[j] = expf(fminf(hy * (hx * p00[j] + lx * p01[j]) + ly * (hx * p10[j] + lx * p11[j]) - m, 0.f)) * inv;
        }
    }
}

__global__ void zero_f64_kernel(double* p, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

static int wave_grid(long long items) {
    long long g = (items + 3) / 4;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}

extern "C" int vspw_softmax_lastdim_fwd(const float* x, float* y, long long rows, int k, float alpha, int log,
                                        void* stream) {
    if (!x || !y || rows <= 0 || k <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(softmax_lastdim_fwd_kernel, dim3(wave_grid(rows)), dim3(256), 0, vspw_stream(stream), x, y, rows,
                       k, alpha, log);
    return vspw_launch_status();
}

extern "C" int vspw_softmax_lastdim_bwd(const float* dy, const float* y, float* dx, long long rows, int k,
                                        float alpha, int log, void* stream) {
    if (!dy || !y || !dx || rows <= 0 || k <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(softmax_lastdim_bwd_kernel, dim3(wave_grid(rows)), dim3(256), 0, vspw_stream(stream), dy, y, dx,
                       rows, k, alpha, log);
    return vspw_launch_status();
}

extern "C" int vspw_softmax_pixels_fwd(const float* x, float* y, int b, int hw, int k, float alpha, void* stream) {
    if (!x || !y || b <= 0 || hw <= 0 || k <= 0 || b > 65535) return VSPW_EINVAL;
    hipLaunchKernelGGL(softmax_pixels_fwd_kernel, dim3(vspw_cdiv(k, SP_TX), b), dim3(SP_TX, SP_TY), 0,
                       vspw_stream(stream), x, y, hw, k, alpha);
    return vspw_launch_status();
}

extern "C" int vspw_softmax_pixels_bwd(const float* dy, const float* y, float* dx, int b, int hw, int k, float alpha,
                                       void* stream) {
    if (!dy || !y || !dx || b <= 0 || hw <= 0 || k <= 0 || b > 65535) return VSPW_EINVAL;
    hipLaunchKernelGGL(softmax_pixels_bwd_kernel, dim3(vspw_cdiv(k, SP_TX), b), dim3(SP_TX, SP_TY), 0,
                       vspw_stream(stream), dy, y, dx, hw, k, alpha);
    return vspw_launch_status();
}

extern "C" int vspw_seg_nll_fwd(const float* logp, const void* label, int label_f32, double* out, int n, int h, int w,
                                int k, int H, int W, int ignore_index, int want_acc, void* stream) {
    if (!logp || !label || !out || n <= 0 || h <= 0 || w <= 0 || k <= 0 || H <= 0 || W <= 0) return VSPW_EINVAL;
    const long long total = (long long)n * H * W;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int64_t* li = reinterpret_cast<const int64_t*>(label);
    const float* lf = reinterpret_cast<const float*>(label);
    hipStream_t st = vspw_stream(stream);
    if (want_acc) {
        if (label_f32)
            hipLaunchKernelGGL(seg_nll_fwd_acc_kernel<float>, dim3(wave_grid(total)), dim3(256), 0, st, logp, lf, out, n, h,
                               w, k, H, W, ignore_index, sy, sx);
        else
            hipLaunchKernelGGL(seg_nll_fwd_acc_kernel<int64_t>, dim3(wave_grid(total)), dim3(256), 0, st, logp, li, out, n,
                               h, w, k, H, W, ignore_index, sy, sx);
    } else {
        if (label_f32)
            hipLaunchKernelGGL(seg_nll_fwd_kernel<float>, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, st, logp, lf,
                               out, n, h, w, k, H, W, ignore_index, sy, sx);
        else
            hipLaunchKernelGGL(seg_nll_fwd_kernel<int64_t>, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, st, logp, li,
                               out, n, h, w, k, H, W, ignore_index, sy, sx);
    }
    return vspw_launch_status();
}

extern "C" int vspw_seg_nll_bwd(const float* logp, const void* label, int label_f32, const double* fwd_out,
                                const float* gscale, float* dlogits, int n, int h, int w, int k, int H, int W,
                                int ignore_index, int lsm_jacobian, void* stream) {
    if (!logp || !label || !fwd_out || !gscale || !dlogits) return VSPW_EINVAL;
    if (n <= 0 || h <= 0 || w <= 0 || k <= 0 || k > NLL_MAXK || H <= 0 || W <= 0) return VSPW_EINVAL;
    const long long blocks = (long long)n * h * w;
    if (blocks > 0x7fffffffLL) return VSPW_EINVAL;
    if (label_f32)
        hipLaunchKernelGGL(seg_nll_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, vspw_stream(stream), logp,
                           reinterpret_cast<const float*>(label), fwd_out, gscale, dlogits, n, h, w, k, H, W,
                           ignore_index, (float)h / (float)H, (float)w / (float)W, lsm_jacobian);
    else
        hipLaunchKernelGGL(seg_nll_bwd_kernel<int64_t>, dim3((unsigned)blocks), dim3(256), 0, vspw_stream(stream), logp,
                           reinterpret_cast<const int64_t*>(label), fwd_out, gscale, dlogits, n, h, w, k, H, W,
                           ignore_index, (float)h / (float)H, (float)w / (float)W, lsm_jacobian);
    return vspw_launch_status();
}

extern "C" int vspw_upsample_softmax(const float* logits, float* probs, int n, int h, int w, int k, int H, int W,
                                     void* stream) {
    if (!logits || !probs || n <= 0 || h <= 0 || w <= 0 || k <= 0 || H <= 0 || W <= 0) return VSPW_EINVAL;
    const long long total = (long long)n * H * W;
    hipLaunchKernelGGL(upsample_softmax_kernel, dim3(wave_grid(total)), dim3(256), 0, vspw_stream(stream), logits, probs,
                       n, h, w, k, H, W, (float)h / (float)H, (float)w / (float)W);
    return vspw_launch_status();
}

extern "C" int vspw_zero_f64(double* p, long long n, void* stream) {
    if (!p || n <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(zero_f64_kernel, dim3(vspw_cdiv(n, 256)), dim3(256), 0, vspw_stream(stream), p, n);
    return vspw_launch_status();
}
