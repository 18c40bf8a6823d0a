// common.h — shared device/host helpers for the deepcv_amd HIP kernels (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/deepcv_hip.h"

#define DCV_WAVE 64
#define DCV_NUM_CU 256            // MI355X: 8 XCDs x 32 CUs; used for grid sizing only (never for correctness)

#define DCV_CHECK_ARG(cond) do { if (!(cond)) return DCV_ERR_INVALID_ARG; } while (0)
#define DCV_RETURN_LAUNCH() do { return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH; } while (0)

static inline hipStream_t dcv_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t dcv_ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }
static inline size_t dcv_align_up(size_t a, size_t b) { return dcv_ceil_div(a, b) * b; }

// Host-side launch counters per convolution kernel family (dcv_launch_counts): which kernel served a layer is decided by plans inside this library, and the
// parity tests at the bench sizes assert that the family the bench times is the one they compared (relaxed increments on the launching host thread).
extern unsigned long long dcv_launch_counter_[DCV_KERNEL_FAMILIES];
static inline void dcv_count_launch(int family) { __atomic_fetch_add(&dcv_launch_counter_[family], 1ull, __ATOMIC_RELAXED); }

// Ordered sums of partial slabs (slab_reduce in elementwise.hip): the second stage of every weight-gradient kernel, immediate or deferred.
int dcv_launch_slab_reduce(int n, const dcv_slab_segment* segments, hipStream_t st);

// Grid size for a bandwidth-bound grid-stride kernel: enough blocks to fill 256 CUs x 8 blocks, never more than the work.
static inline unsigned dcv_stream_grid(size_t work_items, unsigned block) {
    size_t blocks = dcv_ceil_div(work_items, block);
    size_t cap = (size_t)DCV_NUM_CU * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

__device__ __forceinline__ float dcv_act_apply(float v, int act, float slope) {
    switch (act) {
        case DCV_ACT_RELU: return v > 0.f ? v : 0.f;
        case DCV_ACT_LEAKY_RELU: return v > 0.f ? v : v * slope;
        case DCV_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case DCV_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// derivative of the activation expressed from its OUTPUT y (matches torch's threshold/leaky_relu/sigmoid/tanh backward formulas)
__device__ __forceinline__ float dcv_act_grad_from_out(float y, int act, float slope) {
    switch (act) {
        case DCV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case DCV_ACT_LEAKY_RELU: return y > 0.f ? 1.f : slope;
        case DCV_ACT_SIGMOID: return y * (1.f - y);
        case DCV_ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}

// A scalar for the host's early read-back (dcv_publish_scalars and the loss kernels which publish on their own).  Packed record — host_seq is the word
// right behind host_vals[0], 8-byte aligned —: ONE system-scope 64-bit store {value, sequence number}; the host polls the sequence word and then reads the
// value of the same write, no fence on the device (a release fence at system scope writes the XCD's L2 back: ~1 us on the publishing thread's block,
// which the whole grid waits for at its next synchronisation).  Otherwise: the value into every slot, fence, the sequence number, as before.
__device__ __forceinline__ void dcv_publish_value(float* host_vals, int* host_seq, int slots, float value, int seq) {
    if (host_seq == reinterpret_cast<int*>(host_vals) + 1 && (reinterpret_cast<unsigned long long>(host_vals) & 7ull) == 0ull) {
        const unsigned long long rec = ((unsigned long long)(unsigned)seq << 32) | (unsigned long long)__float_as_uint(value);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_vals), rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    for (int i = 0; i < slots; ++i) host_vals[i] = value;
    __threadfence_system();                                              // the values are in host memory before the sequence number is
    *reinterpret_cast<volatile int*>(host_seq) = seq;
    __threadfence_system();
}

// full-wave (64 lanes) butterfly sum; every lane ends with the total
template <typename T>
__device__ __forceinline__ T dcv_wave_sum(T v) {
#pragma unroll
    for (int off = DCV_WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, DCV_WAVE);
    return v;
}

__device__ __forceinline__ float dcv_wave_max(float v) {
#pragma unroll
    for (int off = DCV_WAVE / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, DCV_WAVE));
    return v;
}

// Block-wide sum of one value per thread through LDS (`red` must hold blockDim.x / 64 entries of T); fixed order, result on all threads.
template <typename T>
__device__ __forceinline__ T dcv_block_sum(T v, T* red) {
    const int lane = threadIdx.x & (DCV_WAVE - 1), wid = threadIdx.x >> 6, nw = (blockDim.x + DCV_WAVE - 1) >> 6;
    v = dcv_wave_sum(v);
    __syncthreads();            // protect `red` against a previous use
    if (lane == 0) red[wid] = v;
    __syncthreads();
    T t = T(0);
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
