// Kernels of the frozen RAFT optical-flow network that feeds the NetWarp flow-warp (models/netwarp.py:170-176 runs
// RAFT_core/raft.py:75-127 with iters=20, test_mode=True under torch.no_grad()).  Forward only.  All activations NHWC
// ([pixel rows][channels]); the convolutions go through vspw_conv2d_fwd_ex (conv_igemm.hip), the all-pairs
// correlation through the same NT GEMM, everything else is here:
//   * instance norm statistics + the fused  relu(residual + relu(x*scale + shift))  apply used by both encoders
//     (RAFT_core/extractor.py:44-56,168-190: InstanceNorm2d in fnet, eval-mode BatchNorm2d in cnet);
//   * 2x2 average pooling of the correlation volume over its last two dims (corr.py:27-29);
//   * the 4-level, radius-4 bilinear correlation lookup (corr.py:31-52 + utils/utils.py:57-71);
//   * the SepConvGRU gating elementwise ops (update.py:44-60);
//   * convex-combination 8x upsampling of the flow (raft.py:57-68).
// All of these are HBM/latency-bound gathers and elementwise passes: coalesced along the channel (last) dimension.
#include "common.h"

// ------------------------------------------------------------------------------------------------ instance norm
// partial[n][chunk][2][c]: per row-chunk column sums of x and x*x of image n.  float4 lanes along the channels
// (c % 4 == 0), 256 / (c/4) row lanes; a chunk is INORM_ROWS rows so that even the half-resolution maps give > 1000
// workgroups.
#define INORM_ROWS 256
__global__ __launch_bounds__(256) void inorm_stats_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                          int hw, int c) {
    __shared__ f32x4 red[2][256];
    const int n = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int tid = threadIdx.x;
    const int lanes_c = c >> 2;
    const int row_lanes = 256 / lanes_c;
    const int cl = tid % lanes_c, rl = tid / lanes_c;
    const int r0 = chunk * INORM_ROWS, r1 = min(hw, r0 + INORM_ROWS);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (rl < row_lanes) {
        const f32x4* xp = reinterpret_cast<const f32x4*>(x + (size_t)n * hw * c) + cl;
        for (int r = r0 + rl; r < r1; r += row_lanes) {
            const f32x4 v = xp[(size_t)r * lanes_c];
            s += v;
            q += v * v;
        }
    }
    red[0][tid] = s;
    red[1][tid] = q;
    __syncthreads();
    if (tid < lanes_c) {
        f32x4 ss = {0.f, 0.f, 0.f, 0.f}, qq = {0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < row_lanes; ++l) {
            ss += red[0][l * lanes_c + tid];
            qq += red[1][l * lanes_c + tid];
        }
        float* out = partial + ((size_t)n * nchunk + chunk) * 2 * c;
        reinterpret_cast<f32x4*>(out)[tid] = ss;
        reinterpret_cast<f32x4*>(out + c)[tid] = qq;
    }
}

// scale[n][c] = 1/sqrt(var_biased + eps), shift[n][c] = -mean*scale  (nn.InstanceNorm2d defaults: no affine, no
// running statistics, eps 1e-5; extractor.py:27-31,131).  fp64 combine of the fp32 chunk partials: one workgroup per
// image, 256 / c chunk lanes per channel.
__global__ __launch_bounds__(256) void inorm_finalize_kernel(const float* __restrict__ partial,
                                                             float* __restrict__ scale, float* __restrict__ shift,
                                                             int nchunk, int hw, int c, float eps) {
    __shared__ double red[2][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int lanes = 256 / c;  // c <= 256
    const int ch = tid % c, ln = tid / c;
    double s = 0.0, q = 0.0;
    if (ln < lanes) {
        for (int k = ln; k < nchunk; k += lanes) {
            const float* p = partial + ((size_t)n * nchunk + k) * 2 * c;
            s += (double)p[ch];
            q += (double)p[c + ch];
        }
    }
    red[0][tid] = s;
    red[1][tid] = q;
    __syncthreads();
    if (tid < c) {
        s = 0.0;
        q = 0.0;
        for (int l = 0; l < lanes; ++l) {
            s += red[0][l * c + tid];
            q += red[1][l * c + tid];
        }
        const double mean = s / hw;
        double var = q / hw - mean * mean;
        if (var < 0.0) var = 0.0;
        const float inv = (float)(1.0 / sqrt(var + (double)eps));
        scale[(size_t)n * c + tid] = inv;
        shift[(size_t)n * c + tid] = (float)(-mean) * inv;
    }
}

// y = x*scale + shift; if relu_in: y = max(y,0); if res: y += res; if relu_out: y = max(y,0).
// scale/shift are indexed [image * coef_stride + channel] (coef_stride 0: shared across images = eval BatchNorm).
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ res, float* __restrict__ y,
                                                         long long total4, int hw, int c, int coef_stride,
                                                         int relu_in, int relu_out) {
    // float4 lanes (c % 4 == 0); 32-bit index math per element group
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int c4 = c >> 2;
    const long long per_img4 = (long long)hw * c4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (; i < total4; i += stride) {
        const long long img = i / per_img4;
        const int ch = (int)((i - img * per_img4) % c4) * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + img * coef_stride + ch);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + img * coef_stride + ch);
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i] * sc + sh;
        if (relu_in) v = __builtin_elementwise_max(v, zero);
        if (res) v += reinterpret_cast<const f32x4*>(res)[i];
        if (relu_out) v = __builtin_elementwise_max(v, zero);
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// ------------------------------------------------------------------------------------------ correlation pyramid
// out[p][y][x] = mean of the 2x2 block of in[p] (F.avg_pool2d(corr, 2, stride=2), floor sizes).
__global__ __launch_bounds__(256) void avgpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         long long planes, int h, int w) {
    const int oh = h / 2, ow = w / 2;
    const long long total = planes * oh * ow;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int x = (int)(i % ow);
        const long long t = i / ow;
        const int y = (int)(t % oh);
        const long long p = t / oh;
        const float* s = in + (p * h + 2 * y) * w + 2 * x;
        out[i] = (s[0] + s[1] + s[w] + s[w + 1]) * 0.25f;
    }
}

// grid_sample(bilinear, zeros padding, align_corners=True) of one [h][w] plane at pixel coordinates (x, y), following
// utils/utils.py:57-63 (normalise: 2*x/(W-1)-1) and ATen's un-normalisation ((g+1)/2*(W-1)) step by step so that the
// sampling position carries the same rounding as the reference's.
__device__ __forceinline__ float corr_sample(const float* __restrict__ plane, int h, int w, float x, float y) {
    const float gx = 2.f * x / (float)(w - 1) - 1.f;
    const float gy = 2.f * y / (float)(h - 1) - 1.f;
    const float ix = ((gx + 1.f) / 2.f) * (float)(w - 1);
    const float iy = ((gy + 1.f) / 2.f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float x1 = fx + 1.f, y1 = fy + 1.f;  // ATen: nw = (ix_se - ix)*(iy_se - iy), ...
    const float w00 = (x1 - ix) * (y1 - iy), w01 = (ix - fx) * (y1 - iy), w10 = (x1 - ix) * (iy - fy),
                w11 = (ix - fx) * (iy - fy);
    float v = 0.f;
    const bool xin0 = (x0 >= 0) & (x0 < w), xin1 = (x0 + 1 >= 0) & (x0 + 1 < w);
    const bool yin0 = (y0 >= 0) & (y0 < h), yin1 = (y0 + 1 >= 0) & (y0 + 1 < h);
    if (yin0 & xin0) v += plane[y0 * w + x0] * w00;
    if (yin0 & xin1) v += plane[y0 * w + x0 + 1] * w01;
    if (yin1 & xin0) v += plane[(y0 + 1) * w + x0] * w10;
    if (yin1 & xin1) v += plane[(y0 + 1) * w + x0 + 1] * w11;
    return v;
}

// One workgroup per query pixel (b, i): out[row][l*81 + a*9 + c] = sample(level l plane of that pixel, at
// (cx/2^l + (a-4), cy/2^l + (c-4))).  The reference adds delta[..., 0] = dy-grid value (varying along the FIRST window
// axis) to the x coordinate (corr.py:39-45): window axis a moves x, axis c moves y; kept, the trained weights expect it.
struct CorrPyr {
    const float* lvl[4];
    int h[4], w[4];
};
__global__ __launch_bounds__(128) void corr_lookup_kernel(CorrPyr pyr, const float* __restrict__ flow, int ldf,
                                                          float* __restrict__ out, int ldo, int h1, int w1) {
    const long long row = blockIdx.x;  // b*h1*w1 + i
    const int i = (int)(row % ((long long)h1 * w1));
    const float cx = (float)(i % w1) + flow[row * ldf + 0];
    const float cy = (float)(i / w1) + flow[row * ldf + 1];
    for (int k = threadIdx.x; k < 4 * 81; k += blockDim.x) {
        const int l = k / 81, r = k - l * 81;
        const int a = r / 9, c = r - a * 9;
        const float sc = 1.f / (float)(1 << l);
        const float x = cx * sc + (float)(a - 4);  // coords / 2**i, exact for powers of two
        const float y = cy * sc + (float)(c - 4);
        const float* plane = pyr.lvl[l] + row * (long long)pyr.h[l] * pyr.w[l];
        out[row * ldo + k] = corr_sample(plane, pyr.h[l], pyr.w[l], x, y);
    }
}

// ------------------------------------------------------------------------------------------------------ GRU gates
// zr[row][0..c) = z, zr[row][c..2c) = r (one conv with the z and r filters stacked).  out = r * h
__global__ __launch_bounds__(256) void gru_rh_kernel(const float* __restrict__ zr, int ldzr, const float* __restrict__ h,
                                                     int ldh, float* __restrict__ out, int ldo, long long rows, int c) {
    const long long total = rows * c;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long long row = i / c;
        const int j = (int)(i - row * c);
        out[row * ldo + j] = zr[row * ldzr + c + j] * h[row * ldh + j];
    }
}
// h = (1 - z) * h + z * q   (update.py:52,59), in place
__global__ __launch_bounds__(256) void gru_update_kernel(const float* __restrict__ zr, int ldzr,
                                                         const float* __restrict__ q, int ldq, float* __restrict__ h,
                                                         int ldh, long long rows, int c) {
    const long long total = rows * c;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long long row = i / c;
        const int j = (int)(i - row * c);
        const float z = zr[row * ldzr + j];
        const float hv = h[row * ldh + j];
        h[row * ldh + j] = (1.f - z) * hv + z * q[row * ldq + j];
    }
}

// -------------------------------------------------------------------------------------------- convex upsampling
// raft.py:57-68: mask [n][h][w][9*64] (channel = k*64 + i*8 + j), softmax over k of mask_scale*mask, applied to the
// 3x3 neighbourhood (zero padded) of 8*flow; out NCHW [n][2][8h][8w].  One 64-thread workgroup per coarse pixel.
__global__ __launch_bounds__(64) void convex_upsample_kernel(const float* __restrict__ flow, int ldf,
                                                             const float* __restrict__ mask, int ldm,
                                                             float mask_scale, float* __restrict__ out, int h, int w) {
    const long long row = blockIdx.x;
    const int hw = h * w;
    const int n = (int)(row / hw), p = (int)(row % hw);
    const int y = p / w, x = p % w;
    const int t = threadIdx.x, si = t >> 3, sj = t & 7;
    float m[9];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        m[k] = mask_scale * mask[row * ldm + k * 64 + t];
        mx = fmaxf(mx, m[k]);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        m[k] = expf(m[k] - mx);
        den += m[k];
    }
    float ux = 0.f, uy = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            const long long nb = (long long)n * hw + yy * w + xx;
            const float wk = m[k] / den;
            ux += wk * (8.f * flow[nb * ldf + 0]);
            uy += wk * (8.f * flow[nb * ldf + 1]);
        }
    }
    const int H = 8 * h, W = 8 * w;
    const size_t o = ((size_t)n * 2 * H + (8 * y + si)) * W + 8 * x + sj;
    out[o] = ux;
    out[o + (size_t)H * W] = uy;
}

// ------------------------------------------------------------------------------------------------ thin convolutions
// FlowHead.conv2 (3x3, 256 -> 2 channels, update.py:13-14) is a GEMM in name only: it fills 2 of the 64 columns of an MFMA
// tile (round 5: 54 us per iteration through the implicit GEMM, 20 iterations per forward).  Direct form for few outputs
// (k <= 4; stride 1, undilated, zero padding): one WAVE per output pixel, lane = 4 input channels (+256 per pass), the
// k x taps weight quads of the lane live in registers across the pixels a wave walks; wave_sum per output channel: 22.6 us.
// (The mirror case - BasicMotionEncoder.convf1, 7x7 on the 2 flow channels -> 128 - was tried as a direct kernel with the
// filter bank in LDS too: 73 us against the implicit GEMM's 22; it stays a GEMM.  tools/diag/thin_time.py)
__device__ __forceinline__ float thin_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + expf(-v));
    if (act == 3) return tanhf(v);
    return v;
}

template <int KOUT, int TAPS_MAX>
__global__ __launch_bounds__(256) void conv_few_outputs_kernel(const float* __restrict__ x, long long ldx,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y, long long ldy, int n, int h, int wd,
                                                               int c, int kh, int kw, int ph, int pw, int act,
                                                               int pixels_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long P = (long long)n * h * wd;
    const int taps = kh * kw;
    const int c4 = lane * 4;
    // (c <= 256 per pass; wider inputs take further passes over the channel range)
    for (int cb = 0; cb < c; cb += 256) {
        const bool cok = cb + c4 < c;
        f32x4 wr[KOUT][TAPS_MAX];
#pragma unroll
        for (int k = 0; k < KOUT; ++k)
#pragma unroll
            for (int t = 0; t < TAPS_MAX; ++t) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                wr[k][t] = (cok && t < taps) ? *reinterpret_cast<const f32x4*>(w + ((size_t)k * taps + t) * c + cb + c4) : z;
            }
        const long long p0 = ((long long)blockIdx.x * 4 + wave) * pixels_per_wave;
        for (int i = 0; i < pixels_per_wave; ++i) {
            const long long pix = p0 + i;
            if (pix >= P) break;  // (wave-uniform)
            const int img = (int)(pix / ((long long)h * wd));
            const int r = (int)(pix - (long long)img * h * wd);
            const int oy = r / wd, ox = r - oy * wd;
            float acc[KOUT];
#pragma unroll
            for (int k = 0; k < KOUT; ++k) acc[k] = 0.f;
#pragma unroll
            for (int t = 0; t < TAPS_MAX; ++t) {
                const int ky = t / kw, kx = t - ky * kw;
                const int sy = oy - ph + ky, sx = ox - pw + kx;  // (wave-uniform)
                if (t < taps && (unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)wd && cok) {
                    const f32x4 xv =
                        *reinterpret_cast<const f32x4*>(x + (((size_t)img * h + sy) * wd + sx) * ldx + cb + c4);
#pragma unroll
                    for (int k = 0; k < KOUT; ++k) {
                        const f32x4 pr = xv * wr[k][t];
                        acc[k] += (pr[0] + pr[1]) + (pr[2] + pr[3]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < KOUT; ++k) acc[k] = wave_sum(acc[k]);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < KOUT; ++k) {
                    float* o = y + (size_t)pix * ldy + k;
                    if (c <= 256)
                        *o = thin_act(acc[k] + (bias ? bias[k] : 0.f), act);
                    else if (cb == 0)
                        *o = acc[k];  // partial: later passes add, the last one finishes
                    else if (cb + 256 < c)
                        *o += acc[k];
                    else
                        *o = thin_act(*o + acc[k] + (bias ? bias[k] : 0.f), act);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- C ABI
extern "C" size_t vspw_instance_norm_workspace(int n, int hw, int c) {
    if (n <= 0 || hw <= 0 || c <= 0) return 0;
    const int nchunk = (hw + INORM_ROWS - 1) / INORM_ROWS;
    return (size_t)n * nchunk * 2 * c * sizeof(float);
}

extern "C" int vspw_instance_norm_coeffs(const float* x, int n, int hw, int c, float eps, float* scale, float* shift,
                                         void* ws, size_t ws_bytes, void* stream) {
    if (!x || !scale || !shift || n <= 0 || hw <= 0 || c <= 0 || c > 256 || (c & 3)) return VSPW_EINVAL;
    const int nchunk = (hw + INORM_ROWS - 1) / INORM_ROWS;
    if (!ws || ws_bytes < (size_t)n * nchunk * 2 * c * sizeof(float)) return VSPW_EINVAL;
    float* part = reinterpret_cast<float*>(ws);
    hipStream_t st = vspw_stream(stream);
    hipLaunchKernelGGL(inorm_stats_kernel, dim3(nchunk, n), dim3(256), 0, st, x, part, hw, c);
    hipLaunchKernelGGL(inorm_finalize_kernel, dim3(n), dim3(256), 0, st, part, scale, shift, nchunk, hw, c, eps);
    return vspw_launch_status();
}

extern "C" int vspw_affine_act(const float* x, const float* scale, const float* shift, int coef_stride,
                               const float* residual, int relu_in, int relu_out, float* y, int n, int hw, int c,
                               void* stream) {
    if (!x || !scale || !shift || !y || n <= 0 || hw <= 0 || c <= 0 || (c & 3) || coef_stride < 0 || (coef_stride & 3))
        return VSPW_EINVAL;
    const long long total4 = (long long)n * hw * (c >> 2);
    hipLaunchKernelGGL(affine_act_kernel, dim3(vspw_stream_grid(total4, 256)), dim3(256), 0, vspw_stream(stream), x, scale,
                       shift, residual, y, total4, hw, c, coef_stride, relu_in, relu_out);
    return vspw_launch_status();
}

extern "C" int vspw_avgpool2x2(const float* in, float* out, long long planes, int h, int w, void* stream) {
    if (!in || !out || planes <= 0 || h < 2 || w < 2) return VSPW_EINVAL;
    const long long total = planes * (h / 2) * (w / 2);
    hipLaunchKernelGGL(avgpool2x2_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream), in, out,
                       planes, h, w);
    return vspw_launch_status();
}

extern "C" int vspw_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3, const float* flow,
                                long long ldf, float* out, long long ldo, int b, int h1, int w1, void* stream) {
    if (!l0 || !l1 || !l2 || !l3 || !flow || !out || b <= 0 || h1 < 16 || w1 < 16 || ldf < 2 || ldo < 324)
        return VSPW_EINVAL;  // h1, w1 >= 16: every pyramid level keeps at least 2 rows/columns (the reference divides by W-1)
    CorrPyr pyr;
    pyr.lvl[0] = l0; pyr.lvl[1] = l1; pyr.lvl[2] = l2; pyr.lvl[3] = l3;
    int h = h1, w = w1;
    for (int l = 0; l < 4; ++l) {
        pyr.h[l] = h;
        pyr.w[l] = w;
        h /= 2;
        w /= 2;
    }
    const long long rows = (long long)b * h1 * w1;
    if (rows > 0x7fffffffLL) return VSPW_EINVAL;
    hipLaunchKernelGGL(corr_lookup_kernel, dim3((unsigned)rows), dim3(128), 0, vspw_stream(stream), pyr, flow, (int)ldf,
                       out, (int)ldo, h1, w1);
    return vspw_launch_status();
}

extern "C" int vspw_gru_rh(const float* zr, long long ldzr, const float* h, long long ldh, float* out, long long ldo,
                           long long rows, int c, void* stream) {
    if (!zr || !h || !out || rows <= 0 || c <= 0 || ldzr < 2 * c || ldh < c || ldo < c) return VSPW_EINVAL;
    hipLaunchKernelGGL(gru_rh_kernel, dim3(vspw_stream_grid(rows * c, 256)), dim3(256), 0, vspw_stream(stream), zr,
                       (int)ldzr, h, (int)ldh, out, (int)ldo, rows, c);
    return vspw_launch_status();
}

extern "C" int vspw_gru_update(const float* zr, long long ldzr, const float* q, long long ldq, float* h, long long ldh,
                               long long rows, int c, void* stream) {
    if (!zr || !q || !h || rows <= 0 || c <= 0 || ldzr < c || ldq < c || ldh < c) return VSPW_EINVAL;
    hipLaunchKernelGGL(gru_update_kernel, dim3(vspw_stream_grid(rows * c, 256)), dim3(256), 0, vspw_stream(stream), zr,
                       (int)ldzr, q, (int)ldq, h, (int)ldh, rows, c);
    return vspw_launch_status();
}

extern "C" int vspw_convex_upsample(const float* flow, long long ldf, const float* mask, long long ldm,
                                    float mask_scale, float* out, int n, int h, int w, void* stream) {
    if (!flow || !mask || !out || n <= 0 || h <= 0 || w <= 0 || ldf < 2 || ldm < 576) return VSPW_EINVAL;
    const long long rows = (long long)n * h * w;
    if (rows > 0x7fffffffLL) return VSPW_EINVAL;
    hipLaunchKernelGGL(convex_upsample_kernel, dim3((unsigned)rows), dim3(64), 0, vspw_stream(stream), flow, (int)ldf,
                       mask, (int)ldm, mask_scale, out, h, w);
    return vspw_launch_status();
}

// 1 when vspw_conv2d_thin runs this convolution (stride 1, undilated; see the kernels above)
extern "C" int vspw_conv2d_thin_supported(const vspw_conv_desc* d, long long ldx, long long ldy) {
    if (!d || d->stride != 1 || d->dil != 1 || d->oh != d->h + 2 * d->pad - d->kh + 1 ||
        d->ow != d->w + 2 * d->pad_w - d->kw + 1 || d->oh != d->h || d->ow != d->w)
        return 0;
    const int taps = d->kh * d->kw;
    if (d->k <= 4 && d->c % 4 == 0 && d->c >= 64 && taps <= 9 && ldx % 4 == 0) return 1;
    (void)ldy;
    return 0;
}

extern "C" int vspw_conv2d_thin(const vspw_conv_desc* d, const float* x, long long ldx, const float* w, const float* bias,
                                int act, float* y, long long ldy, void* stream) {
    const int kind = vspw_conv2d_thin_supported(d, ldx, ldy);
    if (!kind || !x || !w || !y || ldx < d->c || ldy < d->k || act < 0 || act > 3) return VSPW_EINVAL;
    // the kernel reads x and w as float4: a channel slot that starts at an odd multiple of 4 bytes is the caller's cue to
    // take vspw_conv2d_fwd_ex instead (models/raft.py checks the same condition before it calls)
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) return VSPW_EINVAL;
    const long long P = (long long)d->n * d->h * d->w;
    hipStream_t st = vspw_stream(stream);
    const int ppw = 4;  // pixels per wave: the weight registers are loaded once per wave
    const int grid = vspw_cdiv(P, 4 * ppw);
#define FEWOUT(KO) hipLaunchKernelGGL((conv_few_outputs_kernel<KO, 9>), dim3(grid), dim3(256), 0, st, x, ldx, w, bias, y, \
                                      ldy, d->n, d->h, d->w, d->c, d->kh, d->kw, d->pad, d->pad_w, act, ppw)
    if (d->k == 1) FEWOUT(1);
    else if (d->k == 2) FEWOUT(2);
    else if (d->k == 3) FEWOUT(3);
    else FEWOUT(4);
#undef FEWOUT
    return vspw_launch_status();
}
