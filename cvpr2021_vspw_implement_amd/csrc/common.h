// Shared device/host helpers for the VSPW hot-path HIP library (gfx950 only).
// All tensors are fp32, activations are NHWC ("pixel rows x channel columns").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/vspw_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VSPW_WAVE 64

extern "C" int vspw_hip_error_code;  // last non-success hipError_t seen by vspw_launch_status (misc.hip)
static inline int vspw_launch_status() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) vspw_hip_error_code = (int)e;
    return e == hipSuccess ? VSPW_OK : VSPW_ELAUNCH;
}

static inline hipStream_t vspw_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int vspw_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Grid size for HBM-bound grid-stride kernels: 256 CUs x 4 workgroups of 256 threads.  (Measured on the BatchNorm apply at the
// layer-3 shapes, tools/diag/bn_bw.py, rotating buffers: 512 blocks 4.1-4.9 TB/s, 1024 4.9-5.8, 2048 - the value of rounds
// 1-4 - 4.6-5.7, 16384 4.7-5.8; the training step: 82.1 -> 81.75 ms.  Two float4 per stream in flight per thread was
// tried too and LOSES 8-12 %.)
static inline int vspw_stream_grid(long long work_items, int block) {
    static const long long cap = getenv("VSPW_STREAM_BLOCKS") ? atoll(getenv("VSPW_STREAM_BLOCKS")) : 256 * 4;  // (env: experiments)
    long long g = (work_items + block - 1) / block;
    if (g > cap && cap >= 1) g = cap;  // (a zero / negative / unparsable VSPW_STREAM_BLOCKS is ignored)
    if (g < 1) g = 1;
    return (int)g;
}

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch, used for speed only); give every XCD a
// contiguous range of tiles so that the N-tiles of one M-tile (same gathered pixels) and neighbouring M-tiles (shared
// 3x3 halo rows) hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblocks >> 3, r = nblocks & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// PyTorch area_pixel_compute_source_index for align_corners=False (bilinear):
// src = max(0, scale*(dst+0.5)-0.5); i0=floor(src); i1=i0+(i0<in-1); l1=src-i0.
__device__ __forceinline__ void bilinear_src(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = src - (float)i0;
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
}
