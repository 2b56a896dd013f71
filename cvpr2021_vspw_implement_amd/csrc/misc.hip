// Small HBM-bound helpers: batched transpose, column sums (bias gradients), axpby, the NetWarp per-channel blend
// and the optical-flow warp (grid_sample) of models/netwarp.py:12-37 with its adjoints.  NHWC fp32.
#include "common.h"

extern "C" int vspw_abi_version(void) { return VSPW_ABI_VERSION; }

int vspw_hip_error_code = 0;
extern "C" int vspw_last_hip_error(void) { return vspw_hip_error_code; }
extern "C" const char* vspw_last_hip_error_string(void) {
    return hipGetErrorString(static_cast<hipError_t>(vspw_hip_error_code));
}

__global__ void transpose_batched_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && c < C) ? in[base + (size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int c = c0 + i, r = r0 + threadIdx.x;
        if (r < R && c < C) out[base + (size_t)c * R + r] = tile[threadIdx.x][i];
    }
}

#define CS_TX 32
#define CS_TY 8
// part[split][c] = sum over the split's rows of a[row][c] * (b ? b[row][c] : 1)
__global__ __launch_bounds__(CS_TX * CS_TY) void colsum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              long long rows, int c, double* __restrict__ part) {
    __shared__ double red[CS_TY][CS_TX * 4];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c0 = (blockIdx.x * CS_TX + tx) * 4;
    double s[4] = {0, 0, 0, 0};
    if (c0 < c) {
        for (long long r = (long long)blockIdx.y * CS_TY + ty; r < rows; r += (long long)gridDim.y * CS_TY) {
            const size_t off = (size_t)r * c + c0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < c) s[e] += b ? (double)a[off + e] * (double)b[off + e] : (double)a[off + e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ty][tx * 4 + e] = s[e];
    __syncthreads();
    const int t = ty * CS_TX + tx;
    if (t < CS_TX * 4) {
        const int ch = blockIdx.x * CS_TX * 4 + t;
        if (ch < c) {
            double v = 0;
#pragma unroll
            for (int j = 0; j < CS_TY; ++j) v += red[j][t];
            part[(size_t)blockIdx.y * c + ch] = v;
        }
    }
}

// out[i] = sum_z part[z][i]: 64 columns x 16 row-lanes per workgroup
__global__ __launch_bounds__(1024) void colsum_final_kernel(const double* __restrict__ part, float* __restrict__ out,
                                                            int splits, int c) {
    __shared__ double red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + tx;
    double a = 0;
    if (col < c)
        for (int z = ty; z < splits; z += 16) a += part[(size_t)z * c + col];
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && col < c) {
        double v = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += red[j][tx];
        out[col] = (float)v;
    }
}

__global__ void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float a, float b) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}

__global__ void chan_blend_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                      const float* __restrict__ w0, const float* __restrict__ w1,
                                      float* __restrict__ out, long long rows, int c) {
    const long long total = rows * c;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int ch = (int)(i % c);
        out[i] = w0[ch] * a[i] + w1[ch] * b[i];
    }
}

// out[i] = w[ch] * g[i]
__global__ void chan_scale_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ out,
                                  long long rows, int c) {
    const long long total = rows * c;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) out[i] = w[(int)(i % c)] * g[i];
}

// out[row][col] = s[row] * w[row][col]: folds the eval-mode BatchNorm scale into conv weights [Cout][KH*KW*Cin]
__global__ __launch_bounds__(256) void row_scale_kernel(const float* __restrict__ w, const float* __restrict__ s,
                                                        float* __restrict__ out, long long rows, long long cols) {
    const long long total = rows * cols;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) out[i] = s[i / cols] * w[i];
}
// out[c] = a[c] * s[c] + t[c]: the conv bias seen through BatchNorm
__global__ void fold_bias_kernel(const float* __restrict__ a, const float* __restrict__ s, const float* __restrict__ t,
                                 float* __restrict__ out, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) out[i] = a[i] * s[i] + t[i];
}

// ---- flow warp -------------------------------------------------------------------------------------------------
// flowwarp(x, flo): vgrid = meshgrid + flo; g = 2*vgrid/max(dim-1,1) - 1; grid_sample(x, g, bilinear, zeros,
// align_corners=False) i.e. pixel coordinate p = ((g+1)*dim - 1)/2.  Note the (dim-1) normalisation combined with
// align_corners=False is a quirk of the reference that is kept on purpose.
__device__ __forceinline__ void warp_coords(int ox, int oy, float fx, float fy, int w, int h, float& px, float& py) {
    const float gx = 2.0f * ((float)ox + fx) / (float)max(w - 1, 1) - 1.0f;
    const float gy = 2.0f * ((float)oy + fy) / (float)max(h - 1, 1) - 1.0f;
    px = ((gx + 1.f) * (float)w - 1.f) * 0.5f;
    py = ((gy + 1.f) * (float)h - 1.f) * 0.5f;
}

__global__ __launch_bounds__(256) void flowwarp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                           float* __restrict__ y, int n, int h, int w, int c) {
    const int lane = threadIdx.x & 63;
    const long long total = (long long)n * h * w;
    long long pix = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; pix < total; pix += stride) {
        const int ox = (int)(pix % w);
        long long r = pix / w;
        const int oy = (int)(r % h);
        const int img = (int)(r / h);
        float px, py;
        warp_coords(ox, oy, flow[pix * 2], flow[pix * 2 + 1], w, h, px, py);
        const float fx0 = floorf(px), fy0 = floorf(py);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float tx = px - fx0, ty = py - fy0;
        const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
        const bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
        const float* base = x + (size_t)img * h * w * c;
        for (int ch = lane; ch < c; ch += 64) {
            float v = 0.f;
            if (vy0 && vx0) v += w00 * base[((size_t)y0 * w + x0) * c + ch];
            if (vy0 && vx1) v += w01 * base[((size_t)y0 * w + x1) * c + ch];
            if (vy1 && vx0) v += w10 * base[((size_t)y1 * w + x0) * c + ch];
            if (vy1 && vx1) v += w11 * base[((size_t)y1 * w + x1) * c + ch];
            y[(size_t)pix * c + ch] = v;
        }
    }
}

// dx must be zeroed by the caller (scatter with atomics: the flow field is arbitrary, so there is no bounded gather).
__global__ __launch_bounds__(256) void flowwarp_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ flow, float* __restrict__ dx,
                                                           float* __restrict__ dflow, int n, int h, int w, int c) {
    const int lane = threadIdx.x & 63;
    const long long total = (long long)n * h * w;
    long long pix = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    const float dpx = (float)w / (float)max(w - 1, 1);  // d px / d flow_x
    const float dpy = (float)h / (float)max(h - 1, 1);
    for (; pix < total; pix += stride) {
        const int ox = (int)(pix % w);
        long long r = pix / w;
        const int oy = (int)(r % h);
        const int img = (int)(r / h);
        float px, py;
        warp_coords(ox, oy, flow[pix * 2], flow[pix * 2 + 1], w, h, px, py);
        const float fx0 = floorf(px), fy0 = floorf(py);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float tx = px - fx0, ty = py - fy0;
        const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
        const bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
        const size_t ib = (size_t)img * h * w * c;
        float gxs = 0.f, gys = 0.f;
        for (int ch = lane; ch < c; ch += 64) {
            const float g = dy[(size_t)pix * c + ch];
            float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
            if (vy0 && vx0) {
                const size_t o = ib + ((size_t)y0 * w + x0) * c + ch;
                v00 = x[o];
                if (dx) atomicAdd(&dx[o], w00 * g);
            }
            if (vy0 && vx1) {
                const size_t o = ib + ((size_t)y0 * w + x1) * c + ch;
                v01 = x[o];
                if (dx) atomicAdd(&dx[o], w01 * g);
            }
            if (vy1 && vx0) {
                const size_t o = ib + ((size_t)y1 * w + x0) * c + ch;
                v10 = x[o];
                if (dx) atomicAdd(&dx[o], w10 * g);
            }
            if (vy1 && vx1) {
                const size_t o = ib + ((size_t)y1 * w + x1) * c + ch;
                v11 = x[o];
                if (dx) atomicAdd(&dx[o], w11 * g);
            }
            gxs += g * ((v01 - v00) * (1.f - ty) + (v11 - v10) * ty);
            gys += g * ((v10 - v00) * (1.f - tx) + (v11 - v01) * tx);
        }
        if (dflow) {
            gxs = wave_sum(gxs);
            gys = wave_sum(gys);
            if (lane == 0) {
                dflow[pix * 2] = gxs * dpx;
                dflow[pix * 2 + 1] = gys * dpy;
            }
        }
    }
}

static void colsum_plan(long long rows, int c, int& gx, int& gy) {
    gx = vspw_cdiv(c, CS_TX * 4);
    long long want = (2048 + gx - 1) / gx;
    long long maxy = (rows + CS_TY - 1) / CS_TY;
    if (want > maxy) want = maxy;
    if (want < 1) want = 1;
    gy = (int)want;
}

extern "C" size_t vspw_colsum_workspace(long long rows, int c) {
    if (rows <= 0 || c <= 0) return 0;
    int gx, gy;
    colsum_plan(rows, c, gx, gy);
    return (size_t)gy * c * sizeof(double);
}

extern "C" int vspw_colsum_prod(const float* a, const float* b, float* out, long long rows, int c, void* ws,
                                size_t ws_bytes, void* stream) {
    if (!a || !out || rows <= 0 || c <= 0) return VSPW_EINVAL;
    int gx, gy;
    colsum_plan(rows, c, gx, gy);
    if (!ws || ws_bytes < (size_t)gy * c * sizeof(double)) return VSPW_EINVAL;
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(CS_TX, CS_TY), 0, vspw_stream(stream), a, b, rows, c, part);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(vspw_cdiv(c, 64)), dim3(1024), 0, vspw_stream(stream), part, out, gy,
                       c);
    return vspw_launch_status();
}

extern "C" int vspw_colsum(const float* a, float* out, long long rows, int c, void* ws, size_t ws_bytes,
                           void* stream) {
    return vspw_colsum_prod(a, nullptr, out, rows, c, ws, ws_bytes, stream);
}

extern "C" int vspw_transpose_batched(const float* in, float* out, int b, int r, int c, void* stream) {
    if (!in || !out || b <= 0 || r <= 0 || c <= 0 || b > 65535) return VSPW_EINVAL;
    dim3 grid(vspw_cdiv(c, 32), vspw_cdiv(r, 32), b);
    hipLaunchKernelGGL(transpose_batched_kernel, grid, dim3(32, 8), 0, vspw_stream(stream), in, out, r, c);
    return vspw_launch_status();
}

extern "C" int vspw_axpby(const float* x, float* y, long long n, float a, float b, void* stream) {
    if (!x || !y || n <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(axpby_kernel, dim3(vspw_stream_grid(n, 256)), dim3(256), 0, vspw_stream(stream), x, y, n, a, b);
    return vspw_launch_status();
}

extern "C" int vspw_chan_blend_fwd(const float* a, const float* b, const float* w0, const float* w1, float* out,
                                   long long rows, int c, void* stream) {
    if (!a || !b || !w0 || !w1 || !out || rows <= 0 || c <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(chan_blend_fwd_kernel, dim3(vspw_stream_grid(rows * c, 256)), dim3(256), 0, vspw_stream(stream),
                       a, b, w0, w1, out, rows, c);
    return vspw_launch_status();
}

extern "C" int vspw_chan_scale(const float* g, const float* w, float* out, long long rows, int c, void* stream) {
    if (!g || !w || !out || rows <= 0 || c <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(chan_scale_kernel, dim3(vspw_stream_grid(rows * c, 256)), dim3(256), 0, vspw_stream(stream), g,
                       w, out, rows, c);
    return vspw_launch_status();
}

extern "C" int vspw_bn_fold_weights(const float* w, const float* cbias, const float* scale, const float* shift,
                                    float* w_out, float* bias_out, int k, long long cols, void* stream) {
    if (!w || !scale || !shift || !w_out || !bias_out || k <= 0 || cols <= 0) return VSPW_EINVAL;
    hipStream_t st = vspw_stream(stream);
    hipLaunchKernelGGL(row_scale_kernel, dim3(vspw_stream_grid((long long)k * cols, 256)), dim3(256), 0, st, w, scale,
                       w_out, (long long)k, cols);
    if (cbias)
        hipLaunchKernelGGL(fold_bias_kernel, dim3(vspw_cdiv(k, 256)), dim3(256), 0, st, cbias, scale, shift, bias_out, k);
    else
        (void)hipMemcpyAsync(bias_out, shift, (size_t)k * sizeof(float), hipMemcpyDeviceToDevice, st);
    return vspw_launch_status();
}

// grid_sample(mode='nearest', zeros, align_corners=False) on the same grid: the temporal-consistency metric warps the NEXT
// frame's label map onto the current frame (reference TC_cal.py:12-38).  ATen rounds the source coordinate with
// nearbyint (ties to even); a source pixel outside the image gives 0.  One thread per output pixel and channel group.
__global__ __launch_bounds__(256) void flowwarp_nearest_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                               float* __restrict__ y, int n, int h, int w, int c) {
    const long long total = (long long)n * h * w;
    long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; pix < total; pix += stride) {
        const int ox = (int)(pix % w);
        const long long r = pix / w;
        const int oy = (int)(r % h);
        const int img = (int)(r / h);
        float px, py;
        warp_coords(ox, oy, flow[pix * 2], flow[pix * 2 + 1], w, h, px, py);
        const float rx = rintf(px), ry = rintf(py);
        const bool ok = rx >= 0.f && rx <= (float)(w - 1) && ry >= 0.f && ry <= (float)(h - 1);
        const float* src = ok ? x + (((size_t)img * h + (int)ry) * w + (int)rx) * c : nullptr;
        for (int ch = 0; ch < c; ++ch) y[(size_t)pix * c + ch] = ok ? src[ch] : 0.f;
    }
}

static int pixel_wave_grid(long long items) {
    long long g = (items + 3) / 4;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}

extern "C" int vspw_flowwarp_fwd(const float* x, const float* flow, float* y, int n, int h, int w, int c,
                                 void* stream) {
    if (!x || !flow || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(flowwarp_fwd_kernel, dim3(pixel_wave_grid((long long)n * h * w)), dim3(256), 0,
                       vspw_stream(stream), x, flow, y, n, h, w, c);
    return vspw_launch_status();
}

extern "C" int vspw_flowwarp_nearest(const float* x, const float* flow, float* y, int n, int h, int w, int c,
                                     void* stream) {
    if (!x || !flow || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(flowwarp_nearest_kernel, dim3(vspw_stream_grid((long long)n * h * w, 256)), dim3(256), 0,
                       vspw_stream(stream), x, flow, y, n, h, w, c);
    return vspw_launch_status();
}

extern "C" int vspw_flowwarp_bwd(const float* dy, const float* x, const float* flow, float* dx, float* dflow, int n,
                                 int h, int w, int c, void* stream) {
    if (!dy || !x || !flow || n <= 0 || h <= 0 || w <= 0 || c <= 0) return VSPW_EINVAL;
    if (dx) {
        hipError_t e = hipMemsetAsync(dx, 0, (size_t)n * h * w * c * sizeof(float), vspw_stream(stream));
        if (e != hipSuccess) return VSPW_ELAUNCH;
    }
    hipLaunchKernelGGL(flowwarp_bwd_kernel, dim3(pixel_wave_grid((long long)n * h * w)), dim3(256), 0,
                       vspw_stream(stream), dy, x, flow, dx, dflow, n, h, w, c);
    return vspw_launch_status();
}

// ---- SGD with momentum / weight decay, `mult` sequential applications -----------------------------------------
// torch.optim.SGD semantics (train_clip2.py:215-236: momentum 0.9, weight_decay per group, dampening 0, no nesterov)
// applied `mult` times in a row: the reference's get_*_lr_params generators yield a parameter once per enclosing
// module, and torch.optim.SGD of the pinned PyTorch 1.3.1 (README.md:13; a plain Python loop) then updates it once
// per occurrence.  1.3.1 applies the weight decay IN PLACE on p.grad (`d_p = p.grad.data; d_p.add_(wd, p.data)`), so
// the r-th occurrence sees g + wd*(p_0 + ... + p_{r-1}): the decay term accumulates across the occurrences.
//   g += wd*p ; buf = first ? g : momentum*buf + g ; p -= lr*buf        (g lives in a register; p.grad is not rewritten)
__device__ __forceinline__ void sgd_apply(float& pv, float gv, float& bv, bool have, float lr, float wd,
                                          float momentum, int mult) {
    for (int r = 0; r < mult; ++r) {
        gv = gv + wd * pv;
        bv = have ? momentum * bv + gv : gv;
        have = true;
        pv -= lr * bv;
    }
}

__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ buf, long long n, float lr, float wd,
                                                       float momentum, int mult, int first) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float pv = p[i];
        float bv = first ? 0.f : buf[i];
        sgd_apply(pv, g[i], bv, !first, lr, wd, momentum, mult);
        p[i] = pv;
        buf[i] = bv;
    }
}

extern "C" int vspw_sgd_step(float* p, const float* g, float* buf, long long n, float lr, float wd, float momentum,
                             int mult, int first, void* stream) {
    if (!p || !g || !buf || n <= 0 || mult <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(sgd_step_kernel, dim3(vspw_stream_grid(n, 256)), dim3(256), 0, vspw_stream(stream), p, g, buf,
                       n, lr, wd, momentum, mult, first);
    return vspw_launch_status();
}


// ---- multi-tensor SGD: one launch updates every parameter of the model -------------------------------------------
// entries[] (device) is sorted by chunk0; workgroup b finds its tensor by binary search and updates one 16384-element
// chunk of it with the same arithmetic as sgd_step_kernel.  With lr_table != nullptr the learning rate of an entry is
// lr_table[entry.lr_slot] (a small device array the host rewrites per step): the entry table itself can then stay
// constant across steps, which is what a captured hipGraph of the training step needs.
#define SGD_CHUNK 16384
__global__ __launch_bounds__(256) void sgd_multi_kernel(const vspw_sgd_entry* __restrict__ entries, int n_entries,
                                                        float momentum, const float* __restrict__ lr_table) {
    const long long b = blockIdx.x;
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (entries[mid].chunk0 <= b)
            lo = mid;
        else
            hi = mid - 1;
    }
    const vspw_sgd_entry e = entries[lo];
    const long long begin = (b - e.chunk0) * SGD_CHUNK;
    const long long end = min(e.n, begin + SGD_CHUNK);
    float* __restrict__ p = e.p;
    const float* __restrict__ g = e.g;
    float* __restrict__ buf = e.buf;
    const float lr = (lr_table != nullptr && e.lr_slot >= 0) ? lr_table[e.lr_slot] : e.lr;
    for (long long i = begin + threadIdx.x; i < end; i += blockDim.x) {
        float pv = p[i];
        float bv = e.first ? 0.f : buf[i];
        sgd_apply(pv, g[i], bv, !e.first, lr, e.wd, momentum, e.mult);
        p[i] = pv;
        buf[i] = bv;
    }
}

extern "C" long long vspw_sgd_chunk_elems(void) { return SGD_CHUNK; }

extern "C" int vspw_sgd_multi(const vspw_sgd_entry* entries, int n_entries, long long total_chunks, float momentum,
                              const float* lr_table, void* stream) {
    if (!entries || n_entries <= 0 || total_chunks <= 0 || total_chunks > 0x7fffffffLL) return VSPW_EINVAL;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, vspw_stream(stream), entries,
                       n_entries, momentum, lr_table);
    return vspw_launch_status();
}

// ---- plane gathers of the flow plumbing (NCHW planes: [planes][h][w]) ------------------------------------------
// nearest resize  : F.interpolate(flow, size, mode='nearest') (models/netwarp.py:199,214; models/netwarp_ocr.py:252):
//                   src = min(floor(dst * (float)in / out), in - 1), ATen's nearest_idx with a float32 scale
// shift           : out[y][x] = in[y - top][x - left] or 0 outside: F.pad(mode='constant') of RAFT's InputPadder
//                   (RAFT_core/utils/utils.py:7-25) for top, left >= 0 and its unpad crop (:25) for negative offsets
// unnormalize     : (x * std[c] + mean[c]) * 255 - the image un-normalisation in front of the flow network
//                   (models/netwarp.py:186-187), separately rounded products / sums like ATen's elementwise ops
__global__ __launch_bounds__(256) void nearest_resize_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 long long planes, int h, int w, int oh, int ow) {
    const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
    const long long total = planes * oh * ow;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ow);
        long long r = i / ow;
        const int y = (int)(r % oh);
        const long long p = r / oh;
        const int iy = min((int)floorf((float)y * sy), h - 1), ix = min((int)floorf((float)x * sx), w - 1);
        out[i] = in[(p * h + iy) * w + ix];
    }
}

// adjoint as a gather: an input pixel collects the gradient of every output pixel that selected it
__global__ __launch_bounds__(256) void nearest_resize_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                                 long long planes, int h, int w, int oh, int ow) {
    const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
    const long long total = planes * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % w);
        long long r = i / w;
        const int iy = (int)(r % h);
        const long long p = r / h;
        int y0 = (int)floorf((float)iy / sy) - 1, y1 = (int)floorf((float)(iy + 1) / sy) + 1;
        int x0 = (int)floorf((float)ix / sx) - 1, x1 = (int)floorf((float)(ix + 1) / sx) + 1;
        y0 = max(y0, 0); x0 = max(x0, 0); y1 = min(y1, oh - 1); x1 = min(x1, ow - 1);
        float g = 0.f;
        for (int y = y0; y <= y1; ++y) {
            if (min((int)floorf((float)y * sy), h - 1) != iy) continue;
            for (int x = x0; x <= x1; ++x)
                if (min((int)floorf((float)x * sx), w - 1) == ix) g += dout[(p * oh + y) * ow + x];
        }
        din[i] = g;
    }
}

__global__ __launch_bounds__(256) void plane_shift_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          long long planes, int h, int w, int oh, int ow, int top,
                                                          int left) {
    const long long total = planes * oh * ow;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ow);
        long long r = i / ow;
        const int y = (int)(r % oh);
        const long long p = r / oh;
        const int iy = y - top, ix = x - left;
        out[i] = ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? in[(p * h + iy) * w + ix] : 0.f;
    }
}

__global__ __launch_bounds__(256) void unnormalize_kernel(const float* __restrict__ in, float* __restrict__ out, int c,
                                                          long long hw, long long total, float3 stdv, float3 meanv,
                                                          float post) {
#pragma clang fp contract(off)  // x*std, +mean, *post each rounded on its own (hipcc contracts a*b+c into an FMA by default)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)((i / hw) % c);
        const float s = ch == 0 ? stdv.x : (ch == 1 ? stdv.y : stdv.z), m = ch == 0 ? meanv.x : (ch == 1 ? meanv.y : meanv.z);
        const float t = in[i] * s;
        const float u = t + m;
        out[i] = u * post;
    }
}

extern "C" int vspw_nearest_resize_fwd(const float* in, float* out, long long planes, int h, int w, int oh, int ow,
                                       void* stream) {
    if (!in || !out || planes < 1 || h < 1 || w < 1 || oh < 1 || ow < 1) return VSPW_EINVAL;
    hipLaunchKernelGGL(nearest_resize_fwd_kernel, dim3(vspw_stream_grid(planes * oh * ow, 256)), dim3(256), 0,
                       vspw_stream(stream), in, out, planes, h, w, oh, ow);
    return vspw_launch_status();
}

extern "C" int vspw_nearest_resize_bwd(const float* dout, float* din, long long planes, int h, int w, int oh, int ow,
                                       void* stream) {
    if (!dout || !din || planes < 1 || h < 1 || w < 1 || oh < 1 || ow < 1) return VSPW_EINVAL;
    hipLaunchKernelGGL(nearest_resize_bwd_kernel, dim3(vspw_stream_grid(planes * h * w, 256)), dim3(256), 0,
                       vspw_stream(stream), dout, din, planes, h, w, oh, ow);
    return vspw_launch_status();
}

extern "C" int vspw_plane_shift(const float* in, float* out, long long planes, int h, int w, int oh, int ow, int top,
                                int left, void* stream) {
    if (!in || !out || planes < 1 || h < 1 || w < 1 || oh < 1 || ow < 1) return VSPW_EINVAL;
    hipLaunchKernelGGL(plane_shift_kernel, dim3(vspw_stream_grid(planes * oh * ow, 256)), dim3(256), 0,
                       vspw_stream(stream), in, out, planes, h, w, oh, ow, top, left);
    return vspw_launch_status();
}

// dst [n][h][w][c] = src [n][oh][ow][c] at the pixels (y, x) = (s*oy, s*ox), zero elsewhere: the data gradient of a strided
// POINTWISE convolution (the 1x1 stride-2 downsample of layer2.0, models/resnet.py:125-131) is a plain GEMM on the output
// pixels followed by this scatter - through the implicit GEMM's data-gradient gather three quarters of the MFMA work
// multiply zeros (34 TFLOP/s, 275 us; compact GEMM + scatter: ~115 us).
__global__ __launch_bounds__(256) void strided_scatter_kernel(const float* __restrict__ src, float* __restrict__ dst, int n,
                                                              int oh, int ow, int h, int w, int c4, int stride) {
    const long long total = (long long)n * h * w * c4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c4);
        long long r = i / c4;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h);
        const int img = (int)(r / h);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y % stride == 0 && x % stride == 0) {
            const int oy = y / stride, ox = x / stride;
            if (oy < oh && ox < ow) v = reinterpret_cast<const f32x4*>(src)[(((long long)img * oh + oy) * ow + ox) * c4 + ch];
        }
        reinterpret_cast<f32x4*>(dst)[i] = v;
    }
}

extern "C" int vspw_strided_scatter_nhwc(const float* src, float* dst, int n, int oh, int ow, int h, int w, int c, int stride,
                                         void* stream) {
    if (!src || !dst || n < 1 || oh < 1 || ow < 1 || h < 1 || w < 1 || c < 4 || c % 4 || stride < 1) return VSPW_EINVAL;
    const long long total = (long long)n * h * w * (c / 4);
    hipLaunchKernelGGL(strided_scatter_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream), src, dst,
                       n, oh, ow, h, w, c / 4, stride);
    return vspw_launch_status();
}

extern "C" int vspw_unnormalize_rgb(const float* in, float* out, int n, long long hw, float s0, float s1, float s2,
                                    float m0, float m1, float m2, float post, void* stream) {
    if (!in || !out || n < 1 || hw < 1) return VSPW_EINVAL;
    const long long total = (long long)n * 3 * hw;
    hipLaunchKernelGGL(unnormalize_kernel, dim3(vspw_stream_grid(total, 256)), dim3(256), 0, vspw_stream(stream), in, out,
                       3, hw, total, make_float3(s0, s1, s2), make_float3(m0, m1, m2), post);
    return vspw_launch_status();
}
