// Peer exchange of BatchNorm statistics between the ranks of one node (one process per GPU, xGMI).
//
// SynchronizedBatchNorm semantics (reference models/sync_batchnorm/batchnorm.py:110-131, comm.py: the master thread
// collects [sum x, sum x^2, count] from every replica and hands the totals back) need one small all-reduce per BatchNorm
// layer and direction - 224 per TCB-PSP step, each 2*C doubles (<= 32 KB), each on the critical path.  A general
// collective library pays a launch + a protocol per call; here every rank owns an ARENA in its own HBM that all peers
// have mapped (hipIpc), and one single-workgroup kernel per exchange does
//   1. push: write my values into slot (seq % SLOTS), source = my rank, of EVERY rank's arena (posted writes over
//            xGMI).  Each double travels as two 8-byte words {32 data bits | 32-bit sequence tag}: an 8-byte store is
//            atomic, so a word whose tag equals the current sequence number IS valid data - no separate flag, no fence,
//            no acknowledgement round trip (the "LL" idea of collective libraries);
//   2. pull: spin on the words of MY OWN arena (local memory) until every source's words carry the tag, and add the W
//            contributions in rank order (same order on every rank: bit-identical totals everywhere).
// seq lives in device memory and is advanced by the kernel itself, so a captured hipGraph replays correctly.
// Slot reuse: a rank can finish exchange k only after every peer has pushed k, i.e. has finished k-1 entirely; a writer
// is therefore less than two exchanges ahead of any reader and 4 slots are more than enough (tags are compared for
// EQUALITY, so whatever an older exchange left in a slot never matches).
// A peer that never arrives (crashed process) ends the wait after `timeout_ticks` of the 100 MHz wall clock: the
// status word is set, the result is poisoned with NaN and the kernel returns - the GPU never hangs on a dead peer.
// Every arena access is a system-scope relaxed atomic (write-through stores, cache-missing loads): nothing of an
// arena ever sits in an L2, so no cache maintenance (and no write-back of a GEMM's dirty output lines) is involved.
//
// Arena layout: [SLOTS][world][2 * slot_doubles] 8-byte words.
#include <string.h>

#include "common.h"

#define XCHG_SLOTS 4
#define XCHG_MAX_WORLD 16
#define XCHG_THREADS 1024
#define XCHG_PER_LANE 8  // slot_doubles <= XCHG_THREADS * XCHG_PER_LANE
#define XCHG_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define XCHG_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)

typedef unsigned long long u64;

struct XchgPeers {
    char* arena[XCHG_MAX_WORLD];
};

__device__ __forceinline__ u64* xchg_box(char* arena, int slot, int world, int src, int slot_doubles) {
    return reinterpret_cast<u64*>(arena) + ((size_t)slot * world + src) * (2 * (size_t)slot_doubles);
}

__device__ __forceinline__ void xchg_allreduce_body(double* __restrict__ data, int n, const XchgPeers& peers, int world,
                                                    int rank, u64* __restrict__ counter, int slot_doubles,
                                                    long long timeout_ticks, int* __restrict__ status) {
    __shared__ int timed_out;
    const int tid = threadIdx.x;
    const u64 seq = *counter + 1;
    const int slot = (int)(seq % XCHG_SLOTS);
    const u64 tag = (seq & 0xffffffffULL) << 32;  // arenas are zeroed and seq starts at 1: a fresh word never matches
    if (tid == 0) timed_out = 0;
    __syncthreads();
    // 1. push
    for (int i = tid; i < n; i += XCHG_THREADS) {
        const u64 bits = (u64)__double_as_longlong(data[i]);
        const u64 w0 = tag | (bits & 0xffffffffULL), w1 = tag | (bits >> 32);
        for (int w = 0; w < world; ++w) {
            u64* dst = xchg_box(peers.arena[w], slot, world, rank, slot_doubles) + 2 * (size_t)i;
            XCHG_ST(dst, w0);
            XCHG_ST(dst + 1, w1);
        }
    }
    // 2. pull, source by source in rank order
    double acc[XCHG_PER_LANE];
#pragma unroll
    for (int j = 0; j < XCHG_PER_LANE; ++j) acc[j] = 0.0;
    const long long t0 = wall_clock64();
    bool gave_up = false;
    for (int w = 0; w < world && !gave_up; ++w) {
        const u64* src = xchg_box(peers.arena[rank], slot, world, w, slot_doubles);
        unsigned pending = 0;
#pragma unroll
        for (int j = 0; j < XCHG_PER_LANE; ++j)
            if (tid + j * XCHG_THREADS < n) pending |= 1u << j;
        while (pending) {
            u64 a[XCHG_PER_LANE], b[XCHG_PER_LANE];
#pragma unroll
            for (int j = 0; j < XCHG_PER_LANE; ++j) {  // every outstanding word pair in flight together
                if (pending & (1u << j)) {
                    const size_t i = (size_t)(tid + j * XCHG_THREADS);
                    a[j] = XCHG_LD(src + 2 * i);
                    b[j] = XCHG_LD(src + 2 * i + 1);
                }
            }
#pragma unroll
            for (int j = 0; j < XCHG_PER_LANE; ++j) {
                if ((pending & (1u << j)) && (a[j] >> 32) == (tag >> 32) && (b[j] >> 32) == (tag >> 32)) {
                    acc[j] += __longlong_as_double((long long)((a[j] & 0xffffffffULL) | (b[j] << 32)));
                    pending &= ~(1u << j);
                }
            }
            if (pending) {
                if (wall_clock64() - t0 > timeout_ticks) {
                    timed_out = 1;
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
    if (timed_out) {
        if (tid == 0) *status = 1;
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        for (int i = tid; i < n; i += XCHG_THREADS) data[i] = nan;
    } else {
#pragma unroll
        for (int j = 0; j < XCHG_PER_LANE; ++j) {
            const int i = tid + j * XCHG_THREADS;
            if (i < n) data[i] = acc[j];
        }
    }
    if (tid == 0) *counter = seq;
}

__global__ __launch_bounds__(XCHG_THREADS) void xchg_allreduce_kernel(double* __restrict__ data, int n, XchgPeers peers, int world,
                                                             int rank, u64* __restrict__ counter,
                                                             int slot_doubles, long long timeout_ticks,
                                                             int* __restrict__ status) {
    xchg_allreduce_body(data, n, peers, world, rank, counter, slot_doubles, timeout_ticks, status);
}

// Exchange + the training-mode BatchNorm finalisation that consumes the totals (same arithmetic as bn.hip's
// bn_finalize_kernel: batchnorm.py:133-150) in ONE launch: the forward pass of a synchronised BatchNorm costs no more
// launches than an unsynchronised one.
struct XchgFinalize {
    double count;
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    float momentum, eps;
    float* mean;
    float* invstd;
    float* scale;
    float* shift;
    int c, clamp_var;
};

__global__ __launch_bounds__(XCHG_THREADS) void xchg_bn_finalize_kernel(double* __restrict__ sums, XchgPeers peers, int world,
                                                               int rank, u64* __restrict__ counter,
                                                               int slot_doubles, long long timeout_ticks,
                                                               int* __restrict__ status, XchgFinalize f) {
    xchg_allreduce_body(sums, 2 * f.c, peers, world, rank, counter, slot_doubles, timeout_ticks, status);
    __syncthreads();
    for (int i = threadIdx.x; i < f.c; i += blockDim.x) {
        const double m = sums[i] / f.count;
        double var = sums[f.c + i] / f.count - m * m;
        if (var < 0) var = 0;
        const float mf = (float)m;
        const float is = f.clamp_var ? (float)(1.0 / sqrt(var > (double)f.eps ? var : (double)f.eps))
                                     : (float)(1.0 / sqrt(var + (double)f.eps));
        const float g = f.gamma ? f.gamma[i] : 1.f;
        const float b = f.beta ? f.beta[i] : 0.f;
        f.mean[i] = mf;
        f.invstd[i] = is;
        const float sc = g * is;
        f.scale[i] = sc;
        f.shift[i] = b - mf * sc;
        if (f.rmean) f.rmean[i] = (1.f - f.momentum) * f.rmean[i] + f.momentum * mf;
        if (f.rvar) {
            const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
            f.rvar[i] = (1.f - f.momentum) * f.rvar[i] + f.momentum * (float)unb;
        }
    }
}

extern "C" size_t vspw_xchg_arena_bytes(int world, int slot_doubles) {
    if (world < 1 || world > XCHG_MAX_WORLD || slot_doubles < 1 || slot_doubles > XCHG_THREADS * XCHG_PER_LANE) return 0;
    return (size_t)XCHG_SLOTS * world * 2 * slot_doubles * sizeof(u64);
}

extern "C" int vspw_xchg_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

// Allocate this rank's arena (uncached device memory: peers' writes are visible to a running kernel), zero it and
// export its IPC handle (vspw_xchg_handle_bytes() bytes, host memory).  Synchronous; called once at start-up.
extern "C" int vspw_xchg_alloc(size_t bytes, void** arena, void* handle_out) {
    if (!arena || !handle_out || bytes == 0) return VSPW_EINVAL;
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
        vspw_hip_error_code = (int)e;
        (void)hipGetLastError();
        return VSPW_ELAUNCH;
    }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out), p);
    if (e != hipSuccess) {
        vspw_hip_error_code = (int)e;
        (void)hipGetLastError();
        (void)hipFree(p);
        return VSPW_ELAUNCH;
    }
    *arena = p;
    return VSPW_OK;
}

extern "C" int vspw_xchg_open(const void* handle, void** arena) {
    if (!handle || !arena) return VSPW_EINVAL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        vspw_hip_error_code = (int)e;
        (void)hipGetLastError();
        return VSPW_ELAUNCH;
    }
    *arena = p;
    return VSPW_OK;
}

extern "C" int vspw_xchg_close(void* arena) {
    if (!arena) return VSPW_EINVAL;
    return hipIpcCloseMemHandle(arena) == hipSuccess ? VSPW_OK : VSPW_ELAUNCH;
}

extern "C" int vspw_xchg_free(void* arena) {
    if (!arena) return VSPW_EINVAL;
    return hipFree(arena) == hipSuccess ? VSPW_OK : VSPW_ELAUNCH;
}

// In-place sum of data[0..n) (doubles, device) over the `world` ranks whose arenas are arenas[0..world) (HOST array of
// device pointers: arenas[rank] is this rank's own, the others are vspw_xchg_open'ed).  counter: device uint64, zero at
// start-up, owned by the exchange (advanced by every call); status: device int, set to 1 on a timeout.
extern "C" int vspw_xchg_allreduce_f64(double* data, int n, void* const* arenas, int world, int rank,
                                       unsigned long long* counter, int slot_doubles, double timeout_s, int* status,
                                       void* stream) {
    if (!data || !arenas || !counter || !status || n < 1 || n > slot_doubles || world < 1 || world > XCHG_MAX_WORLD ||
        rank < 0 || rank >= world || vspw_xchg_arena_bytes(world, slot_doubles) == 0)
        return VSPW_EINVAL;
    XchgPeers peers;
    for (int w = 0; w < XCHG_MAX_WORLD; ++w) peers.arena[w] = w < world ? reinterpret_cast<char*>(arenas[w]) : nullptr;
    for (int w = 0; w < world; ++w)
        if (!peers.arena[w]) return VSPW_EINVAL;
    const long long ticks = (long long)(timeout_s * 1e8);  // wall_clock64: 100 MHz
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(1), dim3(XCHG_THREADS), 0, vspw_stream(stream), data, n, peers, world, rank,
                       counter, slot_doubles, ticks, status);
    return vspw_launch_status();
}

// vspw_xchg_allreduce_f64 on sums [2][c] followed by vspw_bn_finalize / vspw_bn_finalize_clamped (clamp_var) in one
// launch; count = rows behind the totals over ALL ranks.
extern "C" int vspw_xchg_bn_finalize(double* sums, int c, void* const* arenas, int world, int rank,
                                     unsigned long long* counter, int slot_doubles, double timeout_s, int* status,
                                     double count, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float momentum, float eps, float* mean, float* invstd,
                                     float* scale, float* shift, int clamp_var, void* stream) {
    if (!sums || !arenas || !counter || !status || c < 1 || 2 * c > slot_doubles || world < 1 ||
        world > XCHG_MAX_WORLD || rank < 0 || rank >= world || vspw_xchg_arena_bytes(world, slot_doubles) == 0 || !mean ||
        !invstd || !scale || !shift)
        return VSPW_EINVAL;
    XchgPeers peers;
    for (int w = 0; w < XCHG_MAX_WORLD; ++w) peers.arena[w] = w < world ? reinterpret_cast<char*>(arenas[w]) : nullptr;
    for (int w = 0; w < world; ++w)
        if (!peers.arena[w]) return VSPW_EINVAL;
    XchgFinalize f = {count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, c, clamp_var};
    hipLaunchKernelGGL(xchg_bn_finalize_kernel, dim3(1), dim3(XCHG_THREADS), 0, vspw_stream(stream), sums, peers, world, rank,
                       counter, slot_doubles, (long long)(timeout_s * 1e8), status, f);
    return vspw_launch_status();
}
