// BatchNorm (training + eval) fused with ReLU / residual add / Dropout2d mask, NHWC fp32, gfx950.
//
// Replaces SynchronizedBatchNorm2d.forward (models/sync_batchnorm/batchnorm.py:68-98, i.e. F.batch_norm with
// momentum 0.1 / eps 1e-5 on a single device, or the sum/ssum exchange of :110-150 across devices) plus the
// nn.ReLU / `out += residual` / nn.Dropout2d that follow it in models/resnet.py:40-51,75-90,
// models/clip_psp.py:36-39,75-78, models/clip_ocr.py:44-45,59-61, and the autograd backward of all of them.
//
// All kernels are HBM-bound streaming passes over a [rows][C] matrix (rows = N*H*W pixels):
//   stats      : 1 read                       -> per-channel sum, sum of squares (fp64 accumulation)
//   apply      : 1-2 reads, 1 write           -> z = relu(x*scale+shift (+res)) (*mask)
//   bwd_reduce : 3 reads                      -> sum g, sum g*xhat
//   bwd_apply  : 3 reads, 1-2 writes          -> dx (, dres)
// The cross-rank SyncBN exchange happens between stats and finalize (host side, RCCL all-reduce of `sums`).
#include "common.h"

#define RED_TX 32  // threads across channels (each 4 channels)
#define RED_TY 8   // row lanes

// partial[split][2][c] (fp64)
__global__ __launch_bounds__(RED_TX * RED_TY) void bn_stats_kernel(const float* __restrict__ x, long long rows, int c,
                                                                  double* __restrict__ part) {
    __shared__ double red[2][RED_TY][RED_TX * 4];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c0 = (blockIdx.x * RED_TX + tx) * 4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const bool vec = (c % 4 == 0);
    if (c0 < c) {
        for (long long r = (long long)blockIdx.y * RED_TY + ty; r < rows; r += (long long)gridDim.y * RED_TY) {
            const float* px = x + (size_t)r * c + c0;
            if (vec) {
                f32x4 v = *reinterpret_cast<const f32x4*>(px);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    double d = (double)v[e];
                    s[e] += d;
                    q[e] += d * d;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c0 + e < c) {
                        double d = (double)px[e];
                        s[e] += d;
                        q[e] += d * d;
                    }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][ty][tx * 4 + e] = s[e];
        red[1][ty][tx * 4 + e] = q[e];
    }
    __syncthreads();
    const int t = ty * RED_TX + tx;  // 0..255
    if (t < RED_TX * 4) {
        const int ch = blockIdx.x * RED_TX * 4 + t;
        if (ch < c) {
            double a = 0, b = 0;
#pragma unroll
            for (int j = 0; j < RED_TY; ++j) {
                a += red[0][j][t];
                b += red[1][j][t];
            }
            double* out = part + (size_t)blockIdx.y * 2 * c;
            out[ch] = a;
            out[c + ch] = b;
        }
    }
}

// sums[i] = sum_z part[z][i]: 64 columns x 16 row-lanes per workgroup, coalesced rows, LDS tree at the end.
template <typename T>
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const T* __restrict__ part, double* __restrict__ sums,
                                                               int splits, int n) {
    __shared__ double red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + tx;
    double a0 = 0, a1 = 0;
    if (col < n) {
        int z = ty;
        for (; z + 16 < splits; z += 32) {
            a0 += (double)part[(size_t)z * n + col];
            a1 += (double)part[(size_t)(z + 16) * n + col];
        }
        if (z < splits) a0 += (double)part[(size_t)z * n + col];
    }
    red[ty][tx] = a0 + a1;
    __syncthreads();
    if (ty == 0 && col < n) {
        double a = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) a += red[j][tx];
        sums[col] = a;
    }
}

// reduce the [tiles][2][c] fp32 partials of the conv epilogue AND finalise, one workgroup per 32 channels (128-byte
// rows) x 32 tile groups: these launches are pure latency - 375 tiles in 12 trips of 8 loads instead of 24
__global__ __launch_bounds__(1024) void bn_finalize_partials_kernel(
    const float* __restrict__ part, int tiles, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift, int c) {
    __shared__ double red[2][32][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    double s = 0, q = 0;
    if (i < c) {
#pragma unroll 8
        for (int z = ty; z < tiles; z += 32) {  // unrolled: 16 independent loads in flight, fixed addition order
            s += (double)part[(size_t)z * 2 * c + i];
            q += (double)part[(size_t)z * 2 * c + c + i];
        }
    }
    red[0][ty][tx] = s;
    red[1][ty][tx] = q;
    __syncthreads();
    if (ty != 0 || i >= c) return;
    s = q = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        s += red[0][j][tx];
        q += red[1][j][tx];
    }
    double m = s / count;
    double var = q / count - m * m;
    if (var < 0) var = 0;
    float mf = (float)m;
    float is = (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma ? gamma[i] : 1.f;
    float b = beta ? beta[i] : 0.f;
    mean[i] = mf;
    invstd[i] = is;
    float sc = g * is;
    scale[i] = sc;
    shift[i] = b - mf * sc;
    if (rmean) rmean[i] = (1.f - momentum) * rmean[i] + momentum * mf;
    if (rvar) {
        double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[i] = (1.f - momentum) * rvar[i] + momentum * (float)unb;
    }
}

// Small populations (rows <= 1024: the 1x1 ... 6x6 pyramid-pool branches, the 124 object contexts of the OCR head):
// statistics straight from the activations, TWO-PASS in fp64 (mean, then sum (x - mean)^2), and finalise in the same
// launch.  The one-pass E[x^2] - mean^2 over fp32 tile partials of the conv epilogue loses (mean/std)^2 x 6e-8 of the
// variance; with a population of 2 clips whose pooled features nearly coincide that ratio reaches 1e3-1e4 (measured on
// the bench workload: the scale-1 PPM branch came out 4e-3 off in relative L2, 4x the float32 reference arithmetic).
__global__ __launch_bounds__(256) void bn_small_finalize_kernel(
    const float* __restrict__ x, int rows, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift,
    double* __restrict__ sums, int c) {
    __shared__ double red[8][32];
    __shared__ double mu[32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    double s = 0;
    if (i < c)
        for (int r = ty; r < rows; r += 8) s += (double)x[(size_t)r * c + i];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0) {
        s = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += red[j][tx];
        mu[tx] = s;  // the LOCAL sum; with `sums` the caller exchanges it across ranks and finalises separately
    }
    __syncthreads();
    const double lsum = mu[tx];
    const double m = lsum / (double)rows;
    double q = 0;
    if (i < c)
        for (int r = ty; r < rows; r += 8) {
            const double d = (double)x[(size_t)r * c + i] - m;
            q += d * d;
        }
    __syncthreads();
    red[ty][tx] = q;
    __syncthreads();
    if (ty != 0 || i >= c) return;
    q = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) q += red[j][tx];
    if (sums) {  // [sum x, sum x^2] for the cross-rank exchange (sum x^2 reassembled exactly enough in fp64)
        sums[i] = lsum;
        sums[c + i] = q + lsum * m;
        return;
    }
    double var = q / count;
    float mf = (float)m;
    float is = (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma ? gamma[i] : 1.f;
    float b = beta ? beta[i] : 0.f;
    mean[i] = mf;
    invstd[i] = is;
    float sc = g * is;
    scale[i] = sc;
    shift[i] = b - mf * sc;
    if (rmean) rmean[i] = (1.f - momentum) * rmean[i] + momentum * mf;
    if (rvar) {
        double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[i] = (1.f - momentum) * rvar[i] + momentum * (float)unb;
    }
}

// sums[i] = sum_z part[z][i] (fp64) and the fp32 parameter gradients dbeta = sums[0][:], dgamma = sums[1][:]
template <typename T>
__global__ __launch_bounds__(1024) void reduce_partials_pg_kernel(const T* __restrict__ part,
                                                                  double* __restrict__ sums, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, int splits, int c) {
    __shared__ double red[32][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + tx;
    const int n = 2 * c;
    double a = 0;
    if (col < n) {
#pragma unroll 8
        for (int z = ty; z < splits; z += 32) a += (double)part[(size_t)z * n + col];
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && col < n) {
        double v = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) v += red[j][tx];
        sums[col] = v;
        if (col < c) {
            if (dbeta) dbeta[col] = (float)v;
        } else if (dgamma) {
            dgamma[col - c] = (float)v;
        }
    }
}

// reduce_partials_pg_kernel<float> + bn_bwd_affine_coeffs_kernel in one launch (the pointwise nodes whose BatchNorm-backward
// apply is staged by their gradient GEMMs: 30 per TCB-PSP step): a thread column owns channel i and BOTH of its statistics.
// Same partition of the partial rows (z = ty, ty + 32, ...) and the same order of the 32 group sums as the two-launch form:
// bit-identical sums, parameter gradients and coefficients.
__global__ __launch_bounds__(1024) void bn_bwd_reduce_coeffs_kernel(
    const float* __restrict__ part, int splits, int c, double inv_count, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd, int training, double* __restrict__ sums,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double red[2][32][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    const int n = 2 * c;
    double a = 0, b = 0;
    if (i < c) {
#pragma unroll 8
        for (int z = ty; z < splits; z += 32) {
            a += (double)part[(size_t)z * n + i];
            b += (double)part[(size_t)z * n + c + i];
        }
    }
    red[0][ty][tx] = a;
    red[1][ty][tx] = b;
    __syncthreads();
    if (ty != 0 || i >= c) return;
    double sg = 0, sgx = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        sg += red[0][j][tx];
        sgx += red[1][j][tx];
    }
    sums[i] = sg;
    sums[c + i] = sgx;
    if (dbeta) dbeta[i] = (float)sg;
    if (dgamma) dgamma[i] = (float)sgx;
    const double is = invstd[i];
    const double av = (gamma ? (double)gamma[i] : 1.0) * is;
    double bv = 0.0, kv = 0.0;
    if (training) {
        const double mg = sg * inv_count, mgx = sgx * inv_count;
        bv = -av * is * mgx;
        kv = av * ((double)mean[i] * is * mgx - mg);
    }
    coef[i] = (float)av;
    coef[c + i] = (float)bv;
    coef[2 * c + i] = (float)kv;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean,
                                   float* __restrict__ rvar, float momentum, float eps, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift,
                                   int c, int clamp_var) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    double m = sums[i] / count;
    double var = sums[c + i] / count - m * m;
    if (var < 0) var = 0;
    float mf = (float)m;
    // clamp_var: the reference's multi-device path, bias_var.clamp(eps) ** -0.5 (models/sync_batchnorm/batchnorm.py:150)
    float is = clamp_var ? (float)(1.0 / sqrt(var > (double)eps ? var : (double)eps))
                         : (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma ? gamma[i] : 1.f;
    float b = beta ? beta[i] : 0.f;
    mean[i] = mf;
    invstd[i] = is;
    float sc = g * is;
    scale[i] = sc;
    shift[i] = b - mf * sc;
    if (rmean) rmean[i] = (1.f - momentum) * rmean[i] + momentum * mf;
    if (rvar) {
        double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[i] = (1.f - momentum) * rvar[i] + momentum * (float)unb;
    }
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                      float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                      float* __restrict__ shift, int c) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    float mf = rmean[i];
    float is = 1.f / sqrtf(rvar[i] + eps);
    float g = gamma ? gamma[i] : 1.f;
    float b = beta ? beta[i] : 0.f;
    mean[i] = mf;
    invstd[i] = is;
    float sc = g * is;
    scale[i] = sc;
    shift[i] = b - mf * sc;
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ res,
                                                       const float* __restrict__ mask, float* __restrict__ z,
                                                       long long rows, int c, long long rpi, int relu) {
    const int W = VEC ? 4 : 1;
    const long long cw = c / W;
    const long long total = rows * cw;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (VEC && !mask && stride % cw == 0) {
        // the grid stride is a multiple of the row length: this thread stays on one channel group, so the
        // coefficients live in registers and the loop is pure streaming (no integer division per element)
        const int ch = (int)(i % cw) * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ch);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + ch);
        for (; i < total; i += stride) {
            f32x4 o = reinterpret_cast<const f32x4*>(x)[i] * sc + sh;
            if (res) o += reinterpret_cast<const f32x4*>(res)[i];
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
            }
            reinterpret_cast<f32x4*>(z)[i] = o;
        }
        return;
    }
    for (; i < total; i += stride) {
        const long long r = i / cw;
        const int ch = (int)(i - r * cw) * W;
        const size_t off = (size_t)r * c + ch;
        if (VEC) {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
            f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ch);
            f32x4 sh = *reinterpret_cast<const f32x4*>(shift + ch);
            f32x4 o = v * sc + sh;
            if (res) o += *reinterpret_cast<const f32x4*>(res + off);
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
            }
            if (mask) o *= *reinterpret_cast<const f32x4*>(mask + (size_t)(r / rpi) * c + ch);
            *reinterpret_cast<f32x4*>(z + off) = o;
        } else {
            float o = x[off] * scale[ch] + shift[ch];
            if (res) o += res[off];
            if (relu) o = o > 0.f ? o : 0.f;
            if (mask) o *= mask[(size_t)(r / rpi) * c + ch];
            z[off] = o;
        }
    }
}

__global__ __launch_bounds__(RED_TX * RED_TY) void bn_bwd_reduce_kernel(
    const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ mask, long long rows,
    int c, long long rpi, int relu, double* __restrict__ part) {
    __shared__ double red[2][RED_TY][RED_TX * 4];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c0 = (blockIdx.x * RED_TX + tx) * 4;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const bool vec = (c % 4 == 0);
    if (c0 < c) {
        float mu[4], is[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mu[e] = (c0 + e < c) ? mean[c0 + e] : 0.f;
            is[e] = (c0 + e < c) ? invstd[c0 + e] : 0.f;
        }
        for (long long r = (long long)blockIdx.y * RED_TY + ty; r < rows; r += (long long)gridDim.y * RED_TY) {
            const size_t off = (size_t)r * c + c0;
            f32x4 g = {0.f, 0.f, 0.f, 0.f}, zv = {1.f, 1.f, 1.f, 1.f}, xv = {0.f, 0.f, 0.f, 0.f};
            if (vec) {
                g = *reinterpret_cast<const f32x4*>(dz + off);
                xv = *reinterpret_cast<const f32x4*>(x + off);
                if (relu) zv = *reinterpret_cast<const f32x4*>(z + off);
                if (mask) g *= *reinterpret_cast<const f32x4*>(mask + (size_t)(r / rpi) * c + c0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c0 + e < c) {
                        g[e] = dz[off + e];
                        xv[e] = x[off + e];
                        if (relu) zv[e] = z[off + e];
                        if (mask) g[e] *= mask[(size_t)(r / rpi) * c + c0 + e];
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = (relu && !(zv[e] > 0.f)) ? 0.f : g[e];
                const float xh = (xv[e] - mu[e]) * is[e];
                s[e] += (double)ge;
                q[e] += (double)ge * (double)xh;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][ty][tx * 4 + e] = s[e];
        red[1][ty][tx * 4 + e] = q[e];
    }
    __syncthreads();
    const int t = ty * RED_TX + tx;
    if (t < RED_TX * 4) {
        const int ch = blockIdx.x * RED_TX * 4 + t;
        if (ch < c) {
            double a = 0, b = 0;
#pragma unroll
            for (int j = 0; j < RED_TY; ++j) {
                a += red[0][j][t];
                b += red[1][j][t];
            }
            double* out = part + (size_t)blockIdx.y * 2 * c;
            out[ch] = a;
            out[c + ch] = b;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const double* __restrict__ sums, double inv_count, const float* __restrict__ mask, long long rows, int c,
    long long rpi, int relu, int training, float* __restrict__ dx, float* __restrict__ dres) {
    const int W = VEC ? 4 : 1;
    const long long cw = c / W;
    const long long total = rows * cw;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (VEC && !mask && stride % cw == 0) {
        // fixed channel group per thread (see bn_apply_kernel): per-channel terms are hoisted; the per-element
        // arithmetic is the same expression tree as the general path below (bit-identical results)
        const int ch = (int)(i % cw) * 4;
        f32x4 a = {1.f, 1.f, 1.f, 1.f}, is = a, mu = {0.f, 0.f, 0.f, 0.f}, mg = mu, mgx = mu;
        if (dx) {
            is = *reinterpret_cast<const f32x4*>(invstd + ch);
            f32x4 gm = {1.f, 1.f, 1.f, 1.f};
            if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + ch);
            a = gm * is;
            if (training) {
                mu = *reinterpret_cast<const f32x4*>(mean + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mg[e] = (float)(sums[ch + e] * inv_count);
                    mgx[e] = (float)(sums[c + ch + e] * inv_count);
                }
            }
        }
        for (; i < total; i += stride) {
            f32x4 g = reinterpret_cast<const f32x4*>(dz)[i];
            if (relu) {
                const f32x4 zv = reinterpret_cast<const f32x4*>(z)[i];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!(zv[e] > 0.f)) g[e] = 0.f;
            }
            if (dres) reinterpret_cast<f32x4*>(dres)[i] = g;
            if (dx) {
                f32x4 o;
                if (training) {
                    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (xv[e] - mu[e]) * is[e];
                        o[e] = a[e] * (g[e] - mg[e] - xh * mgx[e]);
                    }
                } else {
                    o = a * g;
                }
                reinterpret_cast<f32x4*>(dx)[i] = o;
            }
        }
        return;
    }
    for (; i < total; i += stride) {
        const long long r = i / cw;
        const int ch = (int)(i - r * cw) * W;
        const size_t off = (size_t)r * c + ch;
        if (VEC) {
            f32x4 g = *reinterpret_cast<const f32x4*>(dz + off);
            if (mask) g *= *reinterpret_cast<const f32x4*>(mask + (size_t)(r / rpi) * c + ch);
            if (relu) {
                const f32x4 zv = *reinterpret_cast<const f32x4*>(z + off);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!(zv[e] > 0.f)) g[e] = 0.f;
            }
            if (dres) *reinterpret_cast<f32x4*>(dres + off) = g;
            if (dx) {
                const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + ch);
                f32x4 gm = {1.f, 1.f, 1.f, 1.f};
                if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + ch);
                f32x4 o;
                if (training) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (xv[e] - mu[e]) * is[e];
                        const float mg = (float)(sums[ch + e] * inv_count);
                        const float mgx = (float)(sums[c + ch + e] * inv_count);
                        o[e] = gm[e] * is[e] * (g[e] - mg - xh * mgx);
                    }
                } else {
                    o = gm * is * g;
                }
                *reinterpret_cast<f32x4*>(dx + off) = o;
            }
        } else {
            float g = dz[off];
            if (mask) g *= mask[(size_t)(r / rpi) * c + ch];
            if (relu && !(z[off] > 0.f)) g = 0.f;
            if (dres) dres[off] = g;
            if (dx) {
                const float is = invstd[ch];
                const float gm = gamma ? gamma[ch] : 1.f;
                float o;
                if (training) {
                    const float xh = (x[off] - mean[ch]) * is;
                    const float mg = (float)(sums[ch] * inv_count);
                    const float mgx = (float)(sums[c + ch] * inv_count);
                    o = gm * is * (g - mg - xh * mgx);
                } else {
                    o = gm * is * g;
                }
                dx[off] = o;
            }
        }
    }
}

__global__ void bn_param_grads_kernel(const double* __restrict__ sums, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, int c) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    if (dbeta) dbeta[i] = (float)sums[i];
    if (dgamma) dgamma[i] = (float)sums[c + i];
}

static void reduce_plan(long long rows, int c, int& gx, int& gy) {
    gx = vspw_cdiv(c, RED_TX * 4);
    long long want = (1024 + gx - 1) / gx;
    long long maxy = (rows + RED_TY * 4 - 1) / (RED_TY * 4);
    if (want > maxy) want = maxy;
    if (want < 1) want = 1;
    gy = (int)want;
}

extern "C" size_t vspw_bn_stats_workspace(long long rows, int c) {
    if (rows <= 0 || c <= 0) return 0;
    int gx, gy;
    reduce_plan(rows, c, gx, gy);
    return (size_t)gy * 2 * c * sizeof(double);
}
extern "C" size_t vspw_bn_bwd_workspace(long long rows, int c) { return vspw_bn_stats_workspace(rows, c); }

extern "C" int vspw_bn_stats(const float* x, long long rows, int c, double* sums, void* ws, size_t ws_bytes,
                             void* stream) {
    if (!x || !sums || rows <= 0 || c <= 0) return VSPW_EINVAL;
    int gx, gy;
    reduce_plan(rows, c, gx, gy);
    if (!ws || ws_bytes < (size_t)gy * 2 * c * sizeof(double)) return VSPW_EINVAL;
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(gx, gy), dim3(RED_TX, RED_TY), 0, vspw_stream(stream), x, rows, c, part);
    hipLaunchKernelGGL(reduce_partials_kernel<double>, dim3(vspw_cdiv(2 * c, 64)), dim3(1024), 0, vspw_stream(stream),
                       (const double*)part, sums, gy, 2 * c);
    return vspw_launch_status();
}

extern "C" int vspw_bn_reduce_partials_f32(const float* part, int tiles, int c, double* sums, void* stream) {
    if (!part || !sums || tiles <= 0 || c <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3(vspw_cdiv(2 * c, 64)), dim3(1024), 0, vspw_stream(stream),
                       part, sums, tiles, 2 * c);
    return vspw_launch_status();
}

extern "C" int vspw_bn_finalize_partials_f32(const float* part, int tiles, double count, const float* gamma,
                                             const float* beta, float* running_mean, float* running_var,
                                             float momentum, float eps, float* mean, float* invstd, float* scale,
                                             float* shift, int c, void* stream) {
    if (!part || tiles <= 0 || !mean || !invstd || !scale || !shift || c <= 0 || !(count > 0)) return VSPW_EINVAL;
    hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3(vspw_cdiv(c, 32)), dim3(1024), 0, vspw_stream(stream), part,
                       tiles, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                       c);
    return vspw_launch_status();
}

extern "C" int vspw_bn_small_finalize(const float* x, int rows, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                      float* invstd, float* scale, float* shift, double* sums, int c, void* stream) {
    if (!x || rows <= 0 || rows > 1024 || c <= 0) return VSPW_EINVAL;
    if (!sums && (!mean || !invstd || !scale || !shift)) return VSPW_EINVAL;
    hipLaunchKernelGGL(bn_small_finalize_kernel, dim3(vspw_cdiv(c, 32)), dim3(256), 0, vspw_stream(stream), x, rows,
                       (double)rows, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                       sums, c);
    return vspw_launch_status();
}

extern "C" int vspw_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                float* invstd, float* scale, float* shift, int c, void* stream) {
    if (!sums || !mean || !invstd || !scale || !shift || c <= 0 || !(count > 0)) return VSPW_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(vspw_cdiv(c, 256)), dim3(256), 0, vspw_stream(stream), sums, count,
                       gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, c, 0);
    return vspw_launch_status();
}

extern "C" int vspw_bn_finalize_clamped(const double* sums, double count, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                        float* invstd, float* scale, float* shift, int c, void* stream) {
    if (!sums || !mean || !invstd || !scale || !shift || c <= 0 || !(count > 0)) return VSPW_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(vspw_cdiv(c, 256)), dim3(256), 0, vspw_stream(stream), sums, count,
                       gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, c, 1);
    return vspw_launch_status();
}

extern "C" int vspw_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* mean, float* invstd, float* scale,
                                   float* shift, int c, void* stream) {
    if (!running_mean || !running_var || !mean || !invstd || !scale || !shift || c <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(vspw_cdiv(c, 256)), dim3(256), 0, vspw_stream(stream), gamma, beta,
                       running_mean, running_var, eps, mean, invstd, scale, shift, c);
    return vspw_launch_status();
}

extern "C" int vspw_bn_apply(const float* x, const float* scale, const float* shift, const float* residual,
                             const float* chan_mask, float* z, long long rows, int c, long long rows_per_image,
                             int relu, void* stream) {
    if (!x || !scale || !shift || !z || rows <= 0 || c <= 0) return VSPW_EINVAL;
    if (chan_mask && rows_per_image <= 0) return VSPW_EINVAL;
    if (rows_per_image <= 0) rows_per_image = rows;
    if (c % 4 == 0) {
        long long total = rows * (c / 4);
        hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(vspw_stream_grid(total, 256)), dim3(256), 0,
                           vspw_stream(stream), x, scale, shift, residual, chan_mask, z, rows, c, rows_per_image, relu);
    } else {
        long long total = rows * c;
        hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(vspw_stream_grid(total, 256)), dim3(256), 0,
                           vspw_stream(stream), x, scale, shift, residual, chan_mask, z, rows, c, rows_per_image, relu);
    }
    return vspw_launch_status();
}

extern "C" int vspw_bn_bwd_reduce_pg(const float* dz, const float* z, const float* x, const float* mean,
                                     const float* invstd, const float* chan_mask, long long rows, int c,
                                     long long rows_per_image, int relu, double* sums, float* dgamma, float* dbeta,
                                     void* ws, size_t ws_bytes, void* stream) {
    if (!dz || !x || !mean || !invstd || !sums || rows <= 0 || c <= 0) return VSPW_EINVAL;
    if (relu && !z) return VSPW_EINVAL;
    if (chan_mask && rows_per_image <= 0) return VSPW_EINVAL;
    if (rows_per_image <= 0) rows_per_image = rows;
    int gx, gy;
    reduce_plan(rows, c, gx, gy);
    if (!ws || ws_bytes < (size_t)gy * 2 * c * sizeof(double)) return VSPW_EINVAL;
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(gx, gy), dim3(RED_TX, RED_TY), 0, vspw_stream(stream), dz, z, x, mean,
                       invstd, chan_mask, rows, c, rows_per_image, relu, part);
    hipLaunchKernelGGL(reduce_partials_pg_kernel<double>, dim3(vspw_cdiv(2 * c, 32)), dim3(1024), 0,
                       vspw_stream(stream), (const double*)part, sums, dgamma, dbeta, gy, c);
    return vspw_launch_status();
}

extern "C" int vspw_bn_bwd_reduce_partials_f32(const float* part, int tiles, int c, double* sums, float* dgamma,
                                               float* dbeta, void* stream) {
    if (!part || !sums || tiles <= 0 || c <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(reduce_partials_pg_kernel<float>, dim3(vspw_cdiv(2 * c, 32)), dim3(1024), 0, vspw_stream(stream),
                       part, sums, dgamma, dbeta, tiles, c);
    return vspw_launch_status();
}

extern "C" int vspw_bn_bwd_reduce_partials_coeffs_f32(const float* part, int tiles, int c, double count, const float* gamma,
                                                      const float* mean, const float* invstd, int training, double* sums,
                                                      float* dgamma, float* dbeta, float* coef, void* stream) {
    if (!part || !sums || !mean || !invstd || !coef || tiles <= 0 || c <= 0 || count <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(bn_bwd_reduce_coeffs_kernel, dim3(vspw_cdiv(c, 32)), dim3(1024), 0, vspw_stream(stream), part, tiles, c,
                       1.0 / count, gamma, mean, invstd, training, sums, dgamma, dbeta, coef);
    return vspw_launch_status();
}

extern "C" int vspw_bn_bwd_reduce(const float* dz, const float* z, const float* x, const float* mean,
                                  const float* invstd, const float* chan_mask, long long rows, int c,
                                  long long rows_per_image, int relu, double* sums, void* ws, size_t ws_bytes,
                                  void* stream) {
    if (!dz || !x || !mean || !invstd || !sums || rows <= 0 || c <= 0) return VSPW_EINVAL;
    if (relu && !z) return VSPW_EINVAL;
    if (chan_mask && rows_per_image <= 0) return VSPW_EINVAL;
    if (rows_per_image <= 0) rows_per_image = rows;
    int gx, gy;
    reduce_plan(rows, c, gx, gy);
    if (!ws || ws_bytes < (size_t)gy * 2 * c * sizeof(double)) return VSPW_EINVAL;
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(gx, gy), dim3(RED_TX, RED_TY), 0, vspw_stream(stream), dz, z, x, mean,
                       invstd, chan_mask, rows, c, rows_per_image, relu, part);
    hipLaunchKernelGGL(reduce_partials_kernel<double>, dim3(vspw_cdiv(2 * c, 64)), dim3(1024), 0, vspw_stream(stream),
                       (const double*)part, sums, gy, 2 * c);
    return vspw_launch_status();
}

extern "C" int vspw_bn_bwd_apply(const float* dz, const float* z, const float* x, const float* mean,
                                 const float* invstd, const float* gamma, const double* sums, double count,
                                 const float* chan_mask, long long rows, int c, long long rows_per_image, int relu,
                                 int training, float* dx, float* dres, float* dgamma, float* dbeta, void* stream) {
    if (!dz || !x || !mean || !invstd || !sums || rows <= 0 || c <= 0 || !(count > 0)) return VSPW_EINVAL;
    if (relu && !z) return VSPW_EINVAL;
    if (chan_mask && rows_per_image <= 0) return VSPW_EINVAL;
    if (rows_per_image <= 0) rows_per_image = rows;
    if (dx || dres) {
        if (c % 4 == 0) {
            long long total = rows * (c / 4);
            hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(vspw_stream_grid(total, 256)), dim3(256), 0,
                               vspw_stream(stream), dz, z, x, mean, invstd, gamma, sums, 1.0 / count, chan_mask, rows,
                               c, rows_per_image, relu, training, dx, dres);
        } else {
            long long total = rows * c;
            hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(vspw_stream_grid(total, 256)), dim3(256), 0,
                               vspw_stream(stream), dz, z, x, mean, invstd, gamma, sums, 1.0 / count, chan_mask, rows,
                               c, rows_per_image, relu, training, dx, dres);
        }
    }
    if (dgamma || dbeta)
        hipLaunchKernelGGL(bn_param_grads_kernel, dim3(vspw_cdiv(c, 256)), dim3(256), 0, vspw_stream(stream), sums,
                           dgamma, dbeta, c);
    return vspw_launch_status();
}

// Coefficients of BatchNorm's backward "apply" as an affine map of (g, y) per channel:
//   dy = a*(g - mg - xhat*mgx),  xhat = (y - mu)*is,  a = gamma*is,  mg = sum(g)/count,  mgx = sum(g*xhat)/count
//      = coef[0]*g + coef[1]*y + coef[2]      with coef[0] = a, coef[1] = -a*is*mgx, coef[2] = a*(mu*is*mgx - mg)
// evaluated in fp64 and rounded once.  Consumed by the affine-operand GEMMs (vspw_conv2d_bwd_data_aff /
// vspw_conv2d_bwd_weight_aff), which therefore never need dy in memory.  training == 0: dy = a*g.
__global__ void bn_bwd_affine_coeffs_kernel(const double* __restrict__ sums, double inv_count,
                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                            const float* __restrict__ invstd, float* __restrict__ coef, int c,
                                            int training) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const double is = invstd[i];
    const double a = (gamma ? (double)gamma[i] : 1.0) * is;
    double b = 0.0, k = 0.0;
    if (training) {
        const double mg = sums[i] * inv_count, mgx = sums[c + i] * inv_count;
        b = -a * is * mgx;
        k = a * ((double)mean[i] * is * mgx - mg);
    }
    coef[i] = (float)a;
    coef[c + i] = (float)b;
    coef[2 * c + i] = (float)k;
}

extern "C" int vspw_bn_bwd_affine_coeffs(const double* sums, double count, const float* gamma, const float* mean,
                                         const float* invstd, float* coef, int c, int training, void* stream) {
    if (!sums || !mean || !invstd || !coef || c <= 0 || count <= 0) return VSPW_EINVAL;
    hipLaunchKernelGGL(bn_bwd_affine_coeffs_kernel, dim3(vspw_cdiv(c, 256)), dim3(256), 0, vspw_stream(stream), sums,
                       1.0 / count, gamma, mean, invstd, coef, c, training);
    return vspw_launch_status();
}
