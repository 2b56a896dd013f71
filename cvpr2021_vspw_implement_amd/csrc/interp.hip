// Bilinear resize (align_corners=False) and channel-slice copies, NHWC fp32, HBM-bound.
// Replaces F.interpolate(mode='bilinear', align_corners=False) + torch.cat(dim=1) of the PPM head
// (models/clip_psp.py:45-53, models/models.py:900-908,960-968) and their autograd adjoints.
// In NHWC the four taps of an output pixel are four contiguous C-vectors; the adjoint is written as a gather over
// the (small) source grid so it needs no atomics and is deterministic.
#include "common.h"

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                           int ih, int iw, int oh, int ow, int c, int ldi, int ci,
                                                           int ldo, int co, float sy, float sx) {
    const int cw = (c + 3) / 4;
    const long long total = (long long)n * oh * ow * cw;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool vec = (c % 4 == 0) && (ldi % 4 == 0) && (ci % 4 == 0) && (ldo % 4 == 0) && (co % 4 == 0);
    for (; i < total; i += stride) {
        const int ch = (int)(i % cw) * 4;
        long long r = i / cw;
        const int ox = (int)(r % ow);
        r /= ow;
        const int oy = (int)(r % oh);
        const int img = (int)(r / oh);
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(oy, sy, ih, y0, y1, ly);
        bilinear_src(ox, sx, iw, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* p00 = x + (((size_t)img * ih + y0) * iw + x0) * ldi + ci + ch;
        const float* p01 = x + (((size_t)img * ih + y0) * iw + x1) * ldi + ci + ch;
        const float* p10 = x + (((size_t)img * ih + y1) * iw + x0) * ldi + ci + ch;
        const float* p11 = x + (((size_t)img * ih + y1) * iw + x1) * ldi + ci + ch;
        float* dst = y + (((size_t)img * oh + oy) * ow + ox) * ldo + co + ch;
        if (vec) {
            f32x4 a = *reinterpret_cast<const f32x4*>(p00);
            f32x4 b = *reinterpret_cast<const f32x4*>(p01);
            f32x4 cc = *reinterpret_cast<const f32x4*>(p10);
            f32x4 d = *reinterpret_cast<const f32x4*>(p11);
            // ATen: h0lambda*(w0lambda*v00 + w1lambda*v01) + h1lambda*(w0lambda*v10 + w1lambda*v11)
            f32x4 o = hy * (hx * a + lx * b) + ly * (hx * cc + lx * d);
            *reinterpret_cast<f32x4*>(dst) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ch + e < c) dst[e] = hy * (hx * p00[e] + lx * p01[e]) + ly * (hx * p10[e] + lx * p11[e]);
        }
    }
}

// dx[img][iy][ix][c] = sum over output pixels whose taps include (iy,ix) of weight * dy
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int n,
                                                           int ih, int iw, int oh, int ow, int c, int ldi, int ci,
                                                           int ldo, int co, float sy, float sx) {
    const long long total = (long long)n * ih * iw * c;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float ry = (float)oh / (float)ih, rx = (float)ow / (float)iw;
    for (; i < total; i += stride) {
        const int ch = (int)(i % c);
        long long r = i / c;
        const int ix = (int)(r % iw);
        r /= iw;
        const int iy = (int)(r % ih);
        const int img = (int)(r / ih);
        // conservative output footprint of source index iy: src in (iy-1, iy+1)
        int oy_lo = (int)floorf(((float)iy - 1.f + 0.5f) * ry - 0.5f) - 1;
        int oy_hi = (int)ceilf(((float)iy + 1.f + 0.5f) * ry - 0.5f) + 1;
        int ox_lo = (int)floorf(((float)ix - 1.f + 0.5f) * rx - 0.5f) - 1;
        int ox_hi = (int)ceilf(((float)ix + 1.f + 0.5f) * rx - 0.5f) + 1;
        if (iy == 0) oy_lo = 0;
        if (iy == ih - 1) oy_hi = oh - 1;
        if (ix == 0) ox_lo = 0;
        if (ix == iw - 1) ox_hi = ow - 1;
        oy_lo = max(oy_lo, 0);
        ox_lo = max(ox_lo, 0);
        oy_hi = min(oy_hi, oh - 1);
        ox_hi = min(ox_hi, ow - 1);
        float g = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1;
            float ly;
            bilinear_src(oy, sy, ih, y0, y1, ly);
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - ly;
            if (y1 == iy) wy += ly;
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1;
                float lx;
                bilinear_src(ox, sx, iw, x0, x1, lx);
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - lx;
                if (x1 == ix) wx += lx;
                if (wx == 0.f) continue;
                g += wy * wx * dy[(((size_t)img * oh + oy) * ow + ox) * ldo + co + ch];
            }
        }
        dx[(((size_t)img * ih + iy) * iw + ix) * ldi + ci + ch] = g;
    }
}

// Same adjoint for small source grids (the PPM branches: 1x1 ... 6x6 sources, 60x60 outputs): one workgroup per
// (image, source pixel, 64-channel chunk); 16 float4 channel lanes x 16 footprint lanes, LDS reduction at the end.
__global__ __launch_bounds__(256) void bilinear_bwd_block_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                 int ih, int iw, int oh, int ow, int c, int ldi, int ci,
                                                                 int ldo, int co, float sy, float sx) {
    __shared__ f32x4 red[16][16];
    const int lc = threadIdx.x & 15, lp = threadIdx.x >> 4;
    const int ch = (blockIdx.x * 16 + lc) * 4;
    const int iy = blockIdx.y / iw, ix = blockIdx.y - iy * iw;
    const int img = blockIdx.z;
    const float ry = (float)oh / (float)ih, rx = (float)ow / (float)iw;
    int oy_lo = (int)floorf(((float)iy - 1.f + 0.5f) * ry - 0.5f) - 1;
    int oy_hi = (int)ceilf(((float)iy + 1.f + 0.5f) * ry - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 1.f + 0.5f) * rx - 0.5f) - 1;
    int ox_hi = (int)ceilf(((float)ix + 1.f + 0.5f) * rx - 0.5f) + 1;
    if (iy == 0) oy_lo = 0;
    if (iy == ih - 1) oy_hi = oh - 1;
    if (ix == 0) ox_lo = 0;
    if (ix == iw - 1) ox_hi = ow - 1;
    oy_lo = max(oy_lo, 0);
    ox_lo = max(ox_lo, 0);
    oy_hi = min(oy_hi, oh - 1);
    ox_hi = min(ox_hi, ow - 1);
    const int fw = ox_hi - ox_lo + 1, fh = oy_hi - oy_lo + 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (ch < c) {
        for (int q = lp; q < fw * fh; q += 16) {
            const int oy = oy_lo + q / fw, ox = ox_lo + q % fw;
            int y0, y1, x0, x1;
            float ly, lx;
            bilinear_src(oy, sy, ih, y0, y1, ly);
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - ly;
            if (y1 == iy) wy += ly;
            if (wy == 0.f) continue;
            bilinear_src(ox, sx, iw, x0, x1, lx);
            float wx = 0.f;
            if (x0 == ix) wx += 1.f - lx;
            if (x1 == ix) wx += lx;
            if (wx == 0.f) continue;
            acc += (wy * wx) * *reinterpret_cast<const f32x4*>(dy + (((size_t)img * oh + oy) * ow + ox) * ldo + co + ch);
        }
    }
    red[lp][lc] = acc;
    __syncthreads();
    if (lp == 0 && ch < c) {
        f32x4 a = red[0][lc];
#pragma unroll
        for (int j = 1; j < 16; ++j) a += red[j][lc];
        *reinterpret_cast<f32x4*>(dx + (((size_t)img * ih + iy) * iw + ix) * ldi + ci + ch) = a;
    }
}

template <bool ADD>
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            long long rows, int c, int lds, int sco, int ldd, int dco) {
    const bool vec = (c % 4 == 0) && (lds % 4 == 0) && (sco % 4 == 0) && (ldd % 4 == 0) && (dco % 4 == 0);
    const int cw = vec ? c / 4 : c;
    const long long total = rows * cw;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long long r = i / cw;
        const int j = (int)(i - r * cw);
        if (vec) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)r * lds + sco + j * 4);
            f32x4* d = reinterpret_cast<f32x4*>(dst + (size_t)r * ldd + dco + j * 4);
            if (ADD) v += *d;
            *d = v;
        } else {
            float v = src[(size_t)r * lds + sco + j];
            float* d = dst + (size_t)r * ldd + dco + j;
            if (ADD) v += *d;
            *d = v;
        }
    }
}

extern "C" int vspw_bilinear_fwd(const float* x, float* y, int n, int ih, int iw, int oh, int ow, int c, int ldi,
                                 int ci, int ldo, int co, void* stream) {
    if (!x || !y || n <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0 || c <= 0) return VSPW_EINVAL;
    if (ci < 0 || co < 0 || ci + c > ldi || co + c > ldo) return VSPW_EINVAL;
    long long total = (long long)n * oh * ow * ((c + 3) / 4);
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream), x, y,
                       n, ih, iw, oh, ow, c, ldi, ci, ldo, co, (float)ih / (float)oh, (float)iw / (float)ow);
    return vspw_launch_status();
}

extern "C" int vspw_bilinear_bwd(const float* dy, float* dx, int n, int ih, int iw, int oh, int ow, int c, int ldi,
                                 int ci, int ldo, int co, void* stream) {
    if (!dy || !dx || n <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0 || c <= 0) return VSPW_EINVAL;
    if (ci < 0 || co < 0 || ci + c > ldi || co + c > ldo) return VSPW_EINVAL;
    const bool vec = (c % 4 == 0) && (ldi % 4 == 0) && (ci % 4 == 0) && (ldo % 4 == 0) && (co % 4 == 0);
    const long long footprint = ((long long)oh * ow) / ((long long)ih * iw);
    if (vec && footprint >= 16 && n <= 65535 && (long long)ih * iw <= 65535) {
        dim3 grid(vspw_cdiv(c, 64), ih * iw, n);
        hipLaunchKernelGGL(bilinear_bwd_block_kernel, grid, dim3(256), 0, vspw_stream(stream), dy, dx, ih, iw, oh, ow,
                           c, ldi, ci, ldo, co, (float)ih / (float)oh, (float)iw / (float)ow);
        return vspw_launch_status();
    }
    long long total = (long long)n * ih * iw * c;
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream), dy,
                       dx, n, ih, iw, oh, ow, c, ldi, ci, ldo, co, (float)ih / (float)oh, (float)iw / (float)ow);
    return vspw_launch_status();
}

extern "C" int vspw_copy_channels(const float* src, float* dst, long long rows, int c, int lds, int sco, int ldd,
                                  int dco, void* stream) {
    if (!src || !dst || rows <= 0 || c <= 0 || sco < 0 || dco < 0 || sco + c > lds || dco + c > ldd)
        return VSPW_EINVAL;
    long long total = rows * c / 4 + 1;
    hipLaunchKernelGGL(copy_channels_kernel<false>, dim3(vspw_stream_grid(total, 256)), dim3(256), 0,
                       vspw_stream(stream), src, dst, rows, c, lds, sco, ldd, dco);
    return vspw_launch_status();
}

extern "C" int vspw_add_channels(const float* src, float* dst, long long rows, int c, int lds, int sco, int ldd,
                                 int dco, void* stream) {
    if (!src || !dst || rows <= 0 || c <= 0 || sco < 0 || dco < 0 || sco + c > lds || dco + c > ldd)
        return VSPW_EINVAL;
    long long total = rows * c / 4 + 1;
    hipLaunchKernelGGL(copy_channels_kernel<true>, dim3(vspw_stream_grid(total, 256)), dim3(256), 0,
                       vspw_stream(stream), src, dst, rows, c, lds, sco, ldd, dco);
    return vspw_launch_status();
}
