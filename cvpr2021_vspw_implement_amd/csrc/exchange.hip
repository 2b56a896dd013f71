// Peer exchange of BatchNorm statistics between the ranks of one node (one process per GPU, xGMI).
//
// SynchronizedBatchNorm semantics (reference models/sync_batchnorm/batchnorm.py:110-131, comm.py: the master thread
// collects [sum x, sum x^2, count] from every replica and hands the totals back) need one small all-reduce per BatchNorm
// layer and direction - 224 per TCB-PSP step, each 2*C doubles (<= 32 KB), each on the critical path.  A general
// collective library pays a launch + a ring protocol per call (measured: 17 us per RCCL all-reduce with ONE rank);
// here every rank owns an ARENA in its own HBM that all peers have mapped (hipIpc), and one single-workgroup kernel does
//   1. push   : write my 2*C doubles into slot (seq % SLOTS), source = my rank, of EVERY rank's arena (posted writes);
//   2. publish: system-scope fence, then store seq into the matching flag word of every arena;
//   3. wait   : spin on the W flag words of MY OWN arena (local memory) until all sources have published seq;
//   4. sum    : add the W contributions in rank order (same order on every rank: bit-identical totals everywhere).
// seq lives in device memory and is advanced by the kernel itself, so a captured hipGraph replays correctly.
// Slot reuse: a rank can finish exchange k+1 only after every peer has PUBLISHED k+1, i.e. has finished reading k; so a
// writer is never more than two exchanges ahead of a reader and 4 slots are enough.
// A peer that never arrives (crashed process) ends the wait after `timeout_ticks` of the 100 MHz wall clock: the
// status word is set, the result is poisoned with NaN and the kernel returns - the GPU never hangs on a dead peer.
//
// Arena layout: [SLOTS][world] uint64 flags (rounded up to 4 KB) | [SLOTS][world][slot_doubles] doubles.
#include <string.h>

#include "common.h"

#define XCHG_SLOTS 4
#define XCHG_MAX_WORLD 16
#define XCHG_FLAG_BYTES 4096

struct XchgPeers {
    char* arena[XCHG_MAX_WORLD];
};

__device__ __forceinline__ unsigned long long* xchg_flag(char* arena, int slot, int world, int src) {
    return reinterpret_cast<unsigned long long*>(arena) + (size_t)slot * world + src;
}
__device__ __forceinline__ double* xchg_box(char* arena, int slot, int world, int src, int slot_doubles) {
    return reinterpret_cast<double*>(arena + XCHG_FLAG_BYTES) + ((size_t)slot * world + src) * slot_doubles;
}

__device__ __forceinline__ void xchg_allreduce_body(double* __restrict__ data, int n, const XchgPeers& peers, int world,
                                                    int rank, unsigned long long* __restrict__ counter,
                                                    int slot_doubles, long long timeout_ticks,
                                                    int* __restrict__ status) {
    __shared__ int timed_out;
    const int tid = threadIdx.x;
    const unsigned long long seq = *counter + 1;  // flags start at 0: the first exchange publishes 1
    const int slot = (int)(seq % XCHG_SLOTS);
    if (tid == 0) timed_out = 0;
    // 1. push
    for (int w = 0; w < world; ++w) {
        double* dst = xchg_box(peers.arena[w], slot, world, rank, slot_doubles);
        for (int i = tid; i < n; i += blockDim.x)
            __hip_atomic_store(dst + i, data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    // 2. publish
    if (tid < world)
        __hip_atomic_store(xchg_flag(peers.arena[tid], slot, world, rank), seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    // 3. wait
    if (tid < world) {
        const unsigned long long* f = xchg_flag(peers.arena[rank], slot, world, tid);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if (wall_clock64() - t0 > timeout_ticks) {
                timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __threadfence_system();
    // 4. sum in rank order
    if (timed_out) {
        if (tid == 0) *status = 1;
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        for (int i = tid; i < n; i += blockDim.x) data[i] = nan;
    } else {
        char* mine = peers.arena[rank];
        for (int i = tid; i < n; i += blockDim.x) {
            double s = 0.0;
            for (int w = 0; w < world; ++w)
                s += __hip_atomic_load(xchg_box(mine, slot, world, w, slot_doubles) + i, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            data[i] = s;
        }
    }
    if (tid == 0) *counter = seq;
}

__global__ __launch_bounds__(512) void xchg_allreduce_kernel(double* __restrict__ data, int n, XchgPeers peers, int world,
                                                             int rank, unsigned long long* __restrict__ counter,
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

__global__ __launch_bounds__(512) void xchg_bn_finalize_kernel(double* __restrict__ sums, XchgPeers peers, int world,
                                                               int rank, unsigned long long* __restrict__ counter,
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
    if (world < 1 || world > XCHG_MAX_WORLD || slot_doubles < 1) return 0;
    if ((size_t)XCHG_SLOTS * world * sizeof(unsigned long long) > XCHG_FLAG_BYTES) return 0;
    return XCHG_FLAG_BYTES + (size_t)XCHG_SLOTS * world * slot_doubles * sizeof(double);
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
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(1), dim3(512), 0, vspw_stream(stream), data, n, peers, world, rank,
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
    hipLaunchKernelGGL(xchg_bn_finalize_kernel, dim3(1), dim3(512), 0, vspw_stream(stream), sums, peers, world, rank,
                       counter, slot_doubles, (long long)(timeout_s * 1e8), status, f);
    return vspw_launch_status();
}
