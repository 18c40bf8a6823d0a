// resident.hip — the LDS-resident network: the training forward of a small-channel DeepcvModule (reference src/deepcv/meta/base_module.py:113-155 over
// `layer()` stacks, src/deepcv/meta/nn.py:519-554) as ONE persistent launch.
//
// Why: the stock CIFAR-10 classifier is five Conv -> act -> BatchNorm -> GroupNorm layers of 4 / 16 channels on 32x32 / 16x16 maps.  A sample's
// activations are 16 KB per layer, so the per-layer kernels spend their time on everything except arithmetic: a launch boundary per kernel (1.6 us), a fill
// and drain of the chip per kernel, an 8 MB write + read of every intermediate tensor.  Here a block of 1024 threads owns TWO samples and keeps them in
// LDS from the input image to the logits (256 blocks = 512 samples, one block per CU); HBM only sees what the backward pass needs later (y, the layer
// outputs, the saved statistics: write-only streams) and the filters (scalar cache).  The one thing a sample needs from the rest of the batch is the
// BatchNorm statistics: per layer each block publishes its 2 * C sums as 8-byte {epoch, value} granules (one agent-scope store each: the tag travels
// with the data, no fence) and gathers all blocks' granules with agent-scope loads until every tag shows the epoch (bounded spin); the sums are then
// reduced in a fixed order in fp64 by every block, so all blocks derive bit-identical statistics.  tools/probes/resident_probe.hip measured a 4 -> 4 5x5
// layer at 12.9 us this way (conv 6.4 + gather 5.1 + HBM stores 1.4) against 21.7 us for its two per-layer launches.
//
// Convolution variants (all read the zero-padded input tile from LDS, filters never touch LDS):
//   c4 -> 4   (stock layers 1-3): two vertically adjacent output pixels per thread, input rows rotate through registers (5 ds_read_b128 per 80
//             v_pk_fma_f32), one filter row (80 scalars) at a time from the scalar cache.
//   c4 -> 8/16 (layer 4): one pixel x 8 output channels per thread, one tap (32 scalars) at a time.
//   16 -> 16  (layer 5): v_mfma_f32_16x16x4_f32, M = the 16 pixels of an image row, N = 16 output channels, K = 4 input channels; the planar LDS layout
//             [channel vector][row][col] makes ONE ds_read_b128 per tap the A operand of four MFMAs; the B fragments are 36 registers loaded straight
//             from the OIHW weight (each lane's 36 values are contiguous there).
// Roofline: VALU / MFMA fp32 (157.3 TF) for the convolutions; the gathers are latency (about 5 us each).
#include "common.h"
#include "conv2d_internal.h"

namespace {

constexpr int RS_T = 1024;          // threads per block: 16 waves, 4 per SIMD (<= 128 VGPRs)
constexpr int RS_SPB = 2;           // samples per block
constexpr int RS_XIN = 1296;        // float4 per sample of the padded input tile: (32 + 4)^2, or 4 planes of (16 + 2)^2
constexpr int RS_Y = 4096;          // floats per sample of a layer's activated convolution output
constexpr int RS_MAXC = 16;         // channels of a resident layer
constexpr int RS_MAXQ = 2 * RS_MAXC;
constexpr int RS_WARM_LINES = 3;   // 64-B filter lines a lane touches in the forward prologue without looking at them (3 x 64 lines = 12 KiB: a 3x3 16 -> 16 filter)
constexpr int RS_MAXG = 256;        // blocks
constexpr int RS_GRAN_OP = RS_MAXQ * RS_MAXG + 2 * RS_MAXQ;   // granules of one synchronisation: hop-1 partials [RS_MAXQ][RS_MAXG], hop-2 totals [RS_MAXQ][2]
constexpr int RS_SPINS = 1 << 16;   // gather polls before giving up (each at least a memory round trip: tens of milliseconds)
constexpr size_t RS_CTL_BYTES = 1024;  // [0] launch generation, [1] error word, [2] test hook (1 + index of a block which never publishes), [3] the block whose time stamps are recorded, [4..): its time stamps (64-bit, 100 MHz): tools/resident_stamps.py
constexpr int RS_MAX_STAMPS = 120;

typedef unsigned long long rs_u64;
typedef __attribute__((address_space(1))) rs_u64 rs_gu64;
typedef __attribute__((address_space(1))) float rs_gf32;
typedef __attribute__((address_space(1))) unsigned rs_gu32;
typedef __attribute__((address_space(4))) const float rs_cfloat;       // constant address space: wave-uniform loads go through the scalar cache
#define RS_CONST(p) ((rs_cfloat*)(unsigned long long)(p))

// LDS map (dynamic): xin [RS_SPB][RS_XIN] float4 | yt [RS_SPB][RS_Y] float.
constexpr size_t RS_XIN_BYTES = (size_t)RS_SPB * RS_XIN * 16, RS_Y_BYTES = (size_t)RS_SPB * RS_Y * 4, RS_SMEM = RS_XIN_BYTES + RS_Y_BYTES;

// Workgroup barrier which only drains the LDS / scalar counter.  `__syncthreads()` also waits for every outstanding global STORE of the wave (vmcnt(0)):
// with 32 KB of y / layer-output stores in flight per phase that wait was 2 us per barrier, although nothing in this launch reads those bytes back
// (the one exception, the link's referenced tensor, sits behind a full __syncthreads()).  Loads are tracked by the compiler per register as usual.
__device__ __forceinline__ void rs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The thread index as a value the optimiser cannot see through.  Every phase of the persistent kernels derives lane / wave / tile addresses from the thread
// index; derived from `threadIdx.x` itself those are loop invariants of the op loop, LLVM hoists dozens of them in front of it and — the kernels sit at the
// 128-VGPR cap of four waves per SIMD — spills them (33 VGPRs = 35 MB of scratch stores per launch, every reload an exposed L2 round trip in the middle of a
// phase).  A phase that starts from rs_tid() recomputes its few index instructions instead.
__device__ __forceinline__ int rs_tid() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
__device__ __forceinline__ int rs_wave_of(int t) { return __builtin_amdgcn_readfirstlane(t >> 6); }        // the wave index as a scalar (t: a thread index of this wave)

typedef float rs_f4 __attribute__((ext_vector_type(4)));

// 1: cin <= 4 -> 4 (VALU, row pairs)   4: cin <= 4 -> 16 on rows of 16 pixels (MFMA, K = 4)   2: cin <= 4 -> 8 / 16 otherwise (VALU)   3: 16 -> 16 (MFMA)
__host__ __device__ __forceinline__ int rs_variant(int cin, int cout, int k, int w) {
    if (cin <= 4 && cout == 4 && (k == 3 || k == 5)) return 1;
    if (cin <= 4 && cout == 16 && (k == 1 || k == 3) && w == 16) return 4;
    if (cin <= 4 && (cout == 8 || cout == 16) && (k == 1 || k == 3)) return 2;
    if (cin == 16 && cout == 16 && k == 3 && w == 16) return 3;
    return 0;
}

// Sum over the 64 lanes without the LDS crossbar (`__shfl_xor` is ds_bpermute_b32: sixteen waves reducing eight values each kept the LDS pipe busy for
// 1.5 us per layer): four DPP steps inside each row of 16 lanes, then the four row sums through v_readlane.  Result on every lane (wave-uniform).
template <int CTRL> __device__ __forceinline__ float rs_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float rs_wave_sum(float v) {
    v += rs_dpp<0xB1>(v);                       // quad_perm [1, 0, 3, 2]
    v += rs_dpp<0x4E>(v);                       // quad_perm [2, 3, 0, 1]
    v += rs_dpp<0x141>(v);                      // row_half_mirror
    v += rs_dpp<0x140>(v);                      // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ double rs_rsqrt(double v) {                   // as norm_chain.hip: fp32 estimate + two Newton steps in fp64
    double r = (double)__frsqrt_rn((float)v);
    r = r * (1.5 - 0.5 * v * r * r);
    return r * (1.5 - 0.5 * v * r * r);
}

// Exact x / d for 0 <= x < 2^20, 0 < d < 2^12 with inv = 1.f / d: the index decodes of the short phases (a 32-bit integer division is ~40 VALU
// instructions, and with 16 waves on 4 SIMDs every instruction of a phase costs 16 cycles of the block's time)
__device__ __forceinline__ int rs_fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
// The program structs are kilobytes of kernel arguments which the prologues read field by field through scalar loads, each behind the branch the previous one
// decided: a chain of scalar-cache misses.  One load per 64-B line of the argument segment (BYTES: all of it, the two pointers behind the struct included),
// all in flight together (one asm statement per group: left to the compiler they go out eight at a time), and the chain hits the scalar cache.  Lines 0 - 30
// for the backward program, 0 - 39 for the forward one; no line past the segment is touched.
template <int BYTES>
__device__ __forceinline__ void rs_kernarg_warm() {
    static_assert((BYTES > 30 * 64 && BYTES <= 31 * 64) || (BYTES > 39 * 64 && BYTES <= 40 * 64), "rs_kernarg_warm: list the lines of this argument segment");
    const __attribute__((address_space(4))) unsigned* ka = (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned d;
    asm volatile("s_load_dword %0, %1, 0x0\n\t" "s_load_dword %0, %1, 0x40\n\t" "s_load_dword %0, %1, 0x80\n\t" "s_load_dword %0, %1, 0xc0\n\t" "s_load_dword %0, %1, 0x100\n\t" "s_load_dword %0, %1, 0x140\n\t" "s_load_dword %0, %1, 0x180\n\t" "s_load_dword %0, %1, 0x1c0\n\t" "s_load_dword %0, %1, 0x200\n\t" "s_load_dword %0, %1, 0x240\n\t" "s_load_dword %0, %1, 0x280\n\t" "s_load_dword %0, %1, 0x2c0\n\t" "s_load_dword %0, %1, 0x300\n\t" "s_load_dword %0, %1, 0x340\n\t" "s_load_dword %0, %1, 0x380\n\t" "s_load_dword %0, %1, 0x3c0\n\t" "s_load_dword %0, %1, 0x400\n\t" "s_load_dword %0, %1, 0x440\n\t" "s_load_dword %0, %1, 0x480\n\t" "s_load_dword %0, %1, 0x4c0\n\t" "s_load_dword %0, %1, 0x500\n\t" "s_load_dword %0, %1, 0x540\n\t" "s_load_dword %0, %1, 0x580\n\t" "s_load_dword %0, %1, 0x5c0\n\t" "s_load_dword %0, %1, 0x600\n\t" "s_load_dword %0, %1, 0x640\n\t" "s_load_dword %0, %1, 0x680\n\t" "s_load_dword %0, %1, 0x6c0\n\t" "s_load_dword %0, %1, 0x700\n\t" "s_load_dword %0, %1, 0x740\n\t" "s_load_dword %0, %1, 0x780\n\t"
                 "s_waitcnt lgkmcnt(0)" : "=&s"(d) : "s"(ka) : "memory");
    if constexpr (BYTES > 39 * 64)
        asm volatile("s_load_dword %0, %1, 0x7c0\n\t" "s_load_dword %0, %1, 0x800\n\t" "s_load_dword %0, %1, 0x840\n\t" "s_load_dword %0, %1, 0x880\n\t" "s_load_dword %0, %1, 0x8c0\n\t" "s_load_dword %0, %1, 0x900\n\t" "s_load_dword %0, %1, 0x940\n\t" "s_load_dword %0, %1, 0x980\n\t" "s_load_dword %0, %1, 0x9c0\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(d) : "s"(ka) : "memory");
}

// Sums of the 8 values of every lane over the 64 lanes in ~30 instructions instead of 8 butterflies: each exchange step also halves what a lane keeps
// (xor 1: half of the values, xor 2: half again), two rotations inside the rows of 16 lanes, two ds_bpermute across the rows.  On return lane l holds,
// in r[0] and r[1], the totals of values (l & 1) * 4 + ((l >> 1) & 1) * 2 + {0, 1}; lanes 0 - 3 together hold all eight.
template <int CTRL> __device__ __forceinline__ float rs_dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ void rs_reduce8(const float (&v)[8], float (&r)[2]) {
    const int lane = rs_tid() & 63;
    const bool b0 = lane & 1, b1 = lane & 2;
    float a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float mine = b0 ? v[4 + i] : v[i], send = b0 ? v[i] : v[4 + i]; a[i] = mine + rs_dpp_mov<0xB1>(send); }     // quad_perm [1, 0, 3, 2]
#pragma unroll
    for (int i = 0; i < 2; ++i) { const float mine = b1 ? a[2 + i] : a[i], send = b1 ? a[i] : a[2 + i]; b[i] = mine + rs_dpp_mov<0x4E>(send); }     // quad_perm [2, 3, 0, 1]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        b[i] += rs_dpp_mov<0x124>(b[i]);        // row_ror:4
        b[i] += rs_dpp_mov<0x128>(b[i]);        // row_ror:8
        b[i] += __shfl_xor(b[i], 16, 64);
        b[i] += __shfl_xor(b[i], 32, 64);
        r[i] = b[i];
    }
}

// Sum of a double over the 64 lanes: four DPP steps inside the rows of 16 lanes (each moves the two 32-bit halves), the four row sums through
// v_readlane, combined in a fixed order.  Wave-uniform result.
template <int CTRL> __device__ __forceinline__ double rs_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rs_wave_sum_f64(double v) {
    v += rs_dpp_f64<0xB1>(v); v += rs_dpp_f64<0x4E>(v); v += rs_dpp_f64<0x141>(v); v += rs_dpp_f64<0x140>(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * k), __builtin_amdgcn_readlane(__double2loint(v), 16 * k));
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// Sums over all blocks of the grid, in two hops of 8-byte {epoch, value} granules (agent-scope stores and loads: the tag travels with the data, no
// fence; every spin is bounded).  Hop 1: thread q < Q of every block publishes its block's partial of quantity q (fp32) to A[q][block].  Quantity q
// belongs to block q % G: that block's first four waves poll A[q][0 .. G), sum the G values in fp64 in a fixed order and publish the total as two
// granules (its low and high word) to B[q].  Hop 2: threads 0 .. 2Q of every block poll B and leave the totals in `totals` (LDS, doubles) — the same
// bits in every block.  A one-hop version (every block gathering all Q * G partials: 64 KB per block at 16 channels, two million polling loads per
// sweep on the chip) took 2.5 - 6.5 us per layer.  Returns false when a poll gave up (a block of the grid is not resident).  Block-uniform control flow.
// hop 1 alone: a caller with independent work (the weight gradients of the backward launch) does it between this and rs_allreduce(..., published = true)
__device__ __forceinline__ void rs_allreduce_publish(rs_u64* ga, unsigned epoch, int Q, float mine, bool mute = false) {
    const int t = rs_tid();
    if (t < Q && !mute) __hip_atomic_store((rs_gu64*)(ga + (size_t)t * RS_MAXG + blockIdx.x), ((rs_u64)epoch << 32) | (rs_u64)__float_as_uint(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the block's share of the reduction: for every quantity it owns, poll the G partials, sum, publish the total (hop 2).  A caller with independent work
// does it between this and rs_allreduce_collect: the totals are then on their way while every block computes.
__device__ __forceinline__ void rs_allreduce_reduce_owned(rs_u64* ga, rs_u64* gb, unsigned epoch, int Q, int G, double* red, int* s_fail) {
    const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t);
    for (int q = (int)blockIdx.x; q < Q; q += G) {                                   // the quantities this block owns

        double d = 0.0;
        if (t < 256) {
            bool got = true;
            float v = 0.f;
            for (int spins = 0; spins < RS_SPINS; ++spins) {
                got = true;
                if (t < G) {
                    const rs_u64 g = __hip_atomic_load((rs_gu64*)(ga + (size_t)q * RS_MAXG + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = __uint_as_float((unsigned)g);
                    got = (unsigned)(g >> 32) == epoch;
                }
                if (__all(got)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!__all(got) && lane == 0) *s_fail = 1;
            d = rs_wave_sum_f64(t < G ? (double)v : 0.0);
            if (lane == 0) red[wave] = d;
        }
        rs_barrier();
        if (t == 0) {
            const double tot = (red[0] + red[1]) + (red[2] + red[3]);
            __hip_atomic_store((rs_gu64*)(gb + 2 * q), ((rs_u64)epoch << 32) | (rs_u64)(unsigned)__double2loint(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((rs_gu64*)(gb + 2 * q + 1), ((rs_u64)epoch << 32) | (rs_u64)(unsigned)__double2hiint(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        rs_barrier();
    }
}

__device__ __forceinline__ bool rs_allreduce_collect(rs_u64* gb, unsigned epoch, int Q, double* totals, unsigned* w32, int* s_fail) {
    const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t);
    if (wave == 0) {
        bool got = true;
        unsigned v = 0;
        for (int spins = 0; spins < RS_SPINS; ++spins) {
            got = true;
            if (t < 2 * Q) {
                const rs_u64 g = __hip_atomic_load((rs_gu64*)(gb + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = (unsigned)g;
                got = (unsigned)(g >> 32) == epoch;
            }
            if (__all(got)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!__all(got) && lane == 0) *s_fail = 1;
        if (t < 2 * Q) w32[t] = v;
    }
    rs_barrier();
    if (t < Q) totals[t] = __hiloint2double((int)w32[2 * t + 1], (int)w32[2 * t]);
    rs_barrier();
    return *s_fail == 0;
}

__device__ __forceinline__ bool rs_allreduce(rs_u64* ga, rs_u64* gb, unsigned epoch, int Q, int G, float mine, double* totals, double* red, unsigned* w32, int* s_fail,
                                             bool published = false, bool mute = false) {
    if (!published) rs_allreduce_publish(ga, epoch, Q, mine, mute);
    rs_allreduce_reduce_owned(ga, gb, epoch, Q, G, red, s_fail);
    return rs_allreduce_collect(gb, epoch, Q, totals, w32, s_fail);
}

// ------------------------------------------------------------------------------------------------ convolutions (LDS -> LDS)
// Input tile of sample s: plane cv of [XH][XW] float4 at xin + s * RS_XIN + cv * XH * XW.  Output: yt[s * RS_Y + (r * W + c) * Cout + co].
// Waves 0 - 7 work on the block's first sample, waves 8 - 15 on the second.  Every wave leaves its sums of y and y^2 per output channel in
// wsum[wave][2][RS_MAXC] (zeros where it had no pixel), except the scalar wide variant (returns false: the caller runs a pass over yt).

// cin <= 4 -> 4 on the matrix cores: v_mfma_f32_4x4x1_16b_f32 = 16 independent 4 x 4 outer products per wave instruction (K = 1), D_b[i][j] += A_b[i] B_b[j]
// with A from lane 4 b + i, B from lane 4 b + j and D_b[i][j] in register i of lane 4 b + j (tools/probes/mfma4x4_probe.hip: 8.4 clocks per instruction
// and SIMD, 97 - 107 MAC / clk / CU with the LDS reads below — the rate of v_pk_fma_f32, but 256 MACs per instruction instead of 128 and none of the
// instructions which a VALU convolution spends around its FMAs: the VALU version (two pixels per thread, 80 v_pk_fma_f32 per filter row with 5 scalar
// loads, 5 ds_read_b128 and 15 register moves) took 7.4 - 7.7 us per 5x5 layer, 46 % of the SIMD cycles in VALU issue with every wave in line for it).
// A wave takes GROUPS of 64 consecutive pixels of its sample: lane l = pixel l of the group, block b = pixels 4 b .. 4 b + 3, j = output channel.  Per
// (tap, input channel) one MFMA: A = x[pixel + tap][ci] — the pixel's float4 is ONE ds_read_b128 per tap for its four MFMAs —, B = w[tap][ci][co = j],
// from the filter copy the prologue left in LDS as [tap][co][ci] (one ds_read_b128 per tap, shared by the wave's two groups).  CI = 3 (the RGB model
// input: the fourth lane of the pixel vectors is zero padding): that MFMA is skipped.  XWC: the tile's row stride when it is known at compile time
// (32-pixel rows: every tap offset is an immediate of its ds_read), 0: run-time geometry.  Output: lane (b, j) holds channel j of pixels 4 b .. 4 b + 3.
constexpr int RS_FILT_OPS = 6;                                                       // forward launch: ops [0, RS_FILT_OPS) may be cin <= 4 -> 4 layers (their filters live in LDS)
constexpr int RB_FILT_SLOTS = 3;                                                     // backward launch (LDS is scarcer there): at most that many 4 <- 4 data gradients
constexpr int RS_FILT_FLOATS = 5 * 5 * 16;

// the prologue's part: thread t < taps * 16 fetches element t of the [tap][co][ci] copy from the prepared [tap][ci][co] filter
__device__ __forceinline__ float rs_filt_fetch(const float* filt, int k, int t) {
    return t < k * k * 16 ? filt[((t >> 4) * 4 + (t & 3)) * 4 + ((t >> 2) & 3)] : 0.f;
}

template <int K, int CI, int XWC>
__device__ __forceinline__ bool rs_conv_c4_co4(const dcv_resident_op& o, const float4* xin, float* yt, const float* wl, const float* bias_lds, float* wsum) {
    constexpr int pad = K / 2;
    const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t), s = wave >> 3, w8 = wave & 7;
    const int H = o.h, W = XWC > 0 ? XWC - 2 * pad : o.w, XW = XWC > 0 ? XWC : W + 2 * pad, HW = H * W, GS = (HW + 63) >> 6;
    const int j = lane & 3;
    const float bias = bias_lds ? bias_lds[j] : 0.f;                                 // (from LDS: see rs_conv_c4_co4_32 (c))
    const int act = o.act; const float slope = o.slope;                              // (requested before the sched_barriers of the tap loop)
    const rs_f4* wl4 = reinterpret_cast<const rs_f4*>(wl) + j;
    const rs_f4* xs = reinterpret_cast<const rs_f4*>(xin) + s * RS_XIN;
    float sum = 0.f, sq = 0.f;
    for (int g0 = w8; g0 < GS; g0 += 16) {                                           // the wave's groups, two at a time: g0 and g0 + 8 (one weight read serves both)
        const bool two = g0 + 8 < GS;
        const int p0 = g0 * 64 + lane, p1 = p0 + 512;
        const int q0 = p0 < HW ? p0 : 0, q1 = (two && p1 < HW) ? p1 : q0;
        int r0, c0, r1, c1;
        if (XWC > 0) { r0 = q0 / W; c0 = q0 - r0 * W; r1 = q1 / W; c1 = q1 - r1 * W; }
        else { const float inv_w = 1.f / (float)W; r0 = rs_fdiv(q0, inv_w); c0 = q0 - r0 * W; r1 = rs_fdiv(q1, inv_w); c1 = q1 - r1 * W; }
        const rs_f4* x0 = xs + r0 * XW + c0;
        const rs_f4* x1 = xs + r1 * XW + c1;
        rs_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // Software pipeline over the taps: the three operand vectors of tap n + 1 are requested BEFORE the eight MFMAs of tap n.  All 16 waves leave the
        // barrier in front of this phase together, so a loop of "request a filter row's operands, wait, multiply" runs the LDS and the matrix pipe one
        // after the other for the whole block (5 x (960 + 1340) clocks: 5.5 us measured for 3.2 us of MFMAs).  sched_barrier keeps the compiler from
        // hoisting every request to the top (75 live vectors: spills).
        rs_f4 wv[2], a0[2], a1[2];
        wv[0] = wl4[0]; a0[0] = x0[0]; a1[0] = x1[0];
#pragma unroll
        for (int tap = 0; tap < K * K; ++tap) {
            const int cur = tap & 1, nxt = cur ^ 1;
            if (tap + 1 < K * K) {
                const int ky = (tap + 1) / K, kx = (tap + 1) - ky * K;
                wv[nxt] = wl4[(tap + 1) * 4]; a0[nxt] = x0[ky * XW + kx]; a1[nxt] = x1[ky * XW + kx];
            }
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[cur].x, wv[cur].x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[cur].x, wv[cur].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[cur].y, wv[cur].y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[cur].y, wv[cur].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[cur].z, wv[cur].z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[cur].z, wv[cur].z, acc1, 0, 0, 0);
            if constexpr (CI > 3) { acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[cur].w, wv[cur].w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[cur].w, wv[cur].w, acc1, 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // lane (b, j): channel j of pixels 4 b + i of each group
        float* y0 = yt + s * RS_Y + (g0 * 64 + (lane & ~3)) * 4 + j;
        const int pb = g0 * 64 + (lane & ~3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (pb + i < HW) { const float v = dcv_act_apply(acc0[i] + bias, act, slope); y0[i * 4] = v; sum += v; sq += v * v; }
            if (two && pb + 512 + i < HW) { const float v = dcv_act_apply(acc1[i] + bias, act, slope); y0[(512 + i) * 4] = v; sum += v; sq += v * v; }
        }
    }
    // the wave's sums of y, y^2 per channel: lanes j, j + 4, ... hold channel j
    sum += rs_dpp_mov<0x124>(sum); sq += rs_dpp_mov<0x124>(sq);                      // row_ror:4
    sum += rs_dpp_mov<0x128>(sum); sq += rs_dpp_mov<0x128>(sq);                      // row_ror:8
    sum += __shfl_xor(sum, 16, 64); sq += __shfl_xor(sq, 16, 64);
    sum += __shfl_xor(sum, 32, 64); sq += __shfl_xor(sq, 32, 64);
    if (lane < 4) { wsum[(wave * 2) * RS_MAXC + lane] = sum; wsum[(wave * 2 + 1) * RS_MAXC + lane] = sq; }
    return true;
}

// The same convolution for 32 x 32 maps (the stock 5x5 layers), tuned to what tools/probes/conv4x4_probe.hip measured on the version above: 1.8 us of MFMAs,
// 2.1 us of x reads (LDS bandwidth: 50 ds_read_b128 per wave), 0.3 us of filter reads, 1.0 us of epilogue (four-way conflicting 4-byte stores), all of it
// back to back.  Here (a) the wave's two groups are vertically adjacent rows — lanes 0 - 31 rows r0 / r0 + 1, lanes 32 - 63 rows r0 + 2 / r0 + 3 — so the x vector
// of tile row ry and column kx is group 0's operand for filter row ry and group 1's for filter row ry - 1: 6 x K reads instead of 2 x K x K, and group 1 reuses
// the K filter vectors group 0 used one row earlier; (b) the roles of the operands are swapped — A = filter (row i = output channel), B = x (column j = pixel) —
// so register i of a lane is channel i of ITS pixel: the output is one conflict-free ds_write_b128 per pixel and the sums are the VALU kernel's rs_reduce8.
// (c) the bias comes from LDS (`bias_lds`, four floats left there by the prologue; NULL: none): a scalar load from memory inside this phase shares its
// counter with the LDS reads (lgkmcnt, and scalar loads return out of order), so the first wait for a tile vector also waited 1 - 1.5 us for the bias; (d) four
// accumulation chains — per group one for the even and one for the odd input channels — so that no MFMA waits for its predecessor's result (one chain per
// group: every wave spent 2.6 us after its last MFMA was issued waiting for the chain to drain).
// MOM (the backward launch's data gradients): the output is the NEXT layer's dout, and the epilogue leaves that layer's two batch moments sum d, sum d u
// (u = (y - mean) rstd of ITS y, `ynext`: the LDS region the output's pixels index too) per wave as [wave][4][RS_MAXC] where the moments pass would
// (resident_bwd_kernel) — the pass itself (2.2 us of index arithmetic, a reduction and two barriers for two pixels per lane) is then skipped.
template <int K, int CI, bool MOM = false>
__device__ __forceinline__ bool rs_conv_c4_co4_32(const dcv_resident_op& o, const float4* xin, float* yt, const float* wl, const float* bias_lds, float* wsum,
                                                  const float* ynext = nullptr, rs_f4 mom_mean = rs_f4{0.f, 0.f, 0.f, 0.f}, rs_f4 mom_rstd = rs_f4{0.f, 0.f, 0.f, 0.f}) {
    constexpr int XW = 32 + 2 * (K / 2), NS = (K + 1) * K, DEPTH = 2;
    const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t), s = wave >> 3, w8 = wave & 7;
    const int r0 = w8 * 4 + (lane >> 5) * 2, c = lane & 31;
    const rs_f4* wl4 = reinterpret_cast<const rs_f4*>(wl) + (lane & 3);
    const rs_f4* x0 = reinterpret_cast<const rs_f4*>(xin) + s * RS_XIN + r0 * XW + c;
    // what the epilogue needs from memory is requested here: behind the sched_barriers of the tap loop its scalar loads (kernel argument, then the bias
    // through the pointer found there) would only be issued when the MFMAs are done — two exposed round trips
    const int act = o.act; const float slope = o.slope;
    rs_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc0b = {0.f, 0.f, 0.f, 0.f}, acc1b = {0.f, 0.f, 0.f, 0.f};
    rs_f4 a[DEPTH + 1], wq[DEPTH + 1], wprev[K];
    auto request = [&](int n, int slot) {
        const int ry = n / K, kx = n - ry * K;
        a[slot] = x0[ry * XW + kx];
        if (ry < K) wq[slot] = wl4[(ry * K + kx) * 4];
    };
#pragma unroll
    for (int n = 0; n < DEPTH; ++n) request(n, n % (DEPTH + 1));
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int slot = n % (DEPTH + 1), ry = n / K, kx = n - ry * K;
        if (n + DEPTH < NS) request(n + DEPTH, (n + DEPTH) % (DEPTH + 1));
        const rs_f4 av = a[slot];
        const bool g1 = ry >= 1, g0 = ry < K;                                        // group 1 (rows r0 + 1): filter row ry - 1; group 0 (rows r0): filter row ry
        const rs_f4 w1 = wprev[kx], w0 = g0 ? wq[slot] : wprev[kx];
        if (g1) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w1.x, av.x, acc1, 0, 0, 0);
        if (g0) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w0.x, av.x, acc0, 0, 0, 0);
        if (g1) acc1b = __builtin_amdgcn_mfma_f32_4x4x1f32(w1.y, av.y, acc1b, 0, 0, 0);
        if (g0) acc0b = __builtin_amdgcn_mfma_f32_4x4x1f32(w0.y, av.y, acc0b, 0, 0, 0);
        if (g1) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w1.z, av.z, acc1, 0, 0, 0);
        if (g0) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w0.z, av.z, acc0, 0, 0, 0);
        if constexpr (CI > 3) {
            if (g1) acc1b = __builtin_amdgcn_mfma_f32_4x4x1f32(w1.w, av.w, acc1b, 0, 0, 0);
            if (g0) acc0b = __builtin_amdgcn_mfma_f32_4x4x1f32(w0.w, av.w, acc0b, 0, 0, 0);
        }
        if (g0) wprev[kx] = w0;
        __builtin_amdgcn_sched_barrier(0);
    }
    float q8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4* yo = reinterpret_cast<float4*>(yt + s * RS_Y) + r0 * 32 + c;
    if constexpr (MOM) rs_barrier();                                                 // ynext was committed by all threads right before this call: by now a formality
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = dcv_act_apply((g ? acc1[i] + acc1b[i] : acc0[i] + acc0b[i]) + (bias_lds ? bias_lds[i] : 0.f), act, slope);
            if constexpr (!MOM) { q8[i] += v[i]; q8[4 + i] += v[i] * v[i]; }
        }
        if constexpr (MOM) {
            const rs_f4 y4 = *reinterpret_cast<const rs_f4*>(reinterpret_cast<const float4*>(ynext + s * RS_Y) + r0 * 32 + c + g * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float u = (y4[i] - mom_mean[i]) * mom_rstd[i]; q8[i] += v[i]; q8[4 + i] += v[i] * u; }
        }
        yo[g * 32] = make_float4(v[0], v[1], v[2], v[3]);
    }
    float r[2];
    rs_reduce8(q8, r);                                                               // value index = quantity * 4 + channel
    if (lane < 4) {
        constexpr int QS = MOM ? 4 : 2;                                              // quantities per wave in the caller's table
        const int vi = (lane & 1) * 4 + ((lane >> 1) & 1) * 2;                       // r[0], r[1] = values vi, vi + 1
        wsum[(wave * QS + (vi >> 2)) * RS_MAXC + (vi & 3)] = r[0];
        wsum[(wave * QS + (vi >> 2)) * RS_MAXC + (vi & 3) + 1] = r[1];
    }
    return true;
}

// cin <= 4 -> 8 / 16 channels, any map: one pixel x 4 output channels per item, one filter row (K taps x 4 x 4 scalars) at a time; the items of a wave
// share (sample, channel quarter): H * W is a multiple of 64.  (The stock 4 -> 16 layer on 16-pixel rows runs on the matrix cores instead.)
template <int K>
__device__ __forceinline__ bool rs_conv_c4_wide(const dcv_resident_op& o, const float4* xin, float* yt, float* wsum) {
    const int H = o.h, W = o.w, pad = K / 2, XW = W + 2 * pad, HW = H * W, CO = o.cout, nq = CO >> 2;
    const int per = HW * nq;
    for (int it = rs_tid(); it < RS_SPB * per; it += RS_T) {
        const int s = __builtin_amdgcn_readfirstlane(it >= per ? 1 : 0), rem = it - s * per;
        const int qt = __builtin_amdgcn_readfirstlane(rem / HW), px = rem - qt * HW, r = px / W, col = px - r * W;
        const float4* xs = xin + s * RS_XIN + r * XW + col;
        rs_cfloat* wl = RS_CONST(o.filt) + qt * 4;                                   // [K][K][4][CO]
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            float4 xv[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) xv[kx] = xs[ky * XW + kx];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                rs_cfloat* wt = wl + (ky * K + kx) * 4 * CO;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = fmaf(xv[kx].x, wt[c], acc[c]);
                    acc[c] = fmaf(xv[kx].y, wt[CO + c], acc[c]);
                    acc[c] = fmaf(xv[kx].z, wt[2 * CO + c], acc[c]);
                    acc[c] = fmaf(xv[kx].w, wt[3 * CO + c], acc[c]);
                }
            }
        }
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = dcv_act_apply(acc[c] + (o.bias ? RS_CONST(o.bias)[qt * 4 + c] : 0.f), o.act, o.slope);
        reinterpret_cast<float4*>(yt + s * RS_Y + px * CO + qt * 4)[0] = make_float4(v[0], v[1], v[2], v[3]);
    }
    (void)wsum;
    return false;
}

// MFMA epilogue shared by the two matrix-core variants: D holds pixels 4 * (lane / 16) + i of the row, output channel lane % 16
__device__ __forceinline__ void rs_mfma_store(const dcv_resident_op& o, const rs_f4& acc, float bias, float* yp, float& sum, float& sq) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float v = dcv_act_apply(acc[i] + bias, o.act, o.slope); yp[i * 16] = v; sum += v; sq += v * v; }
}
// the same tile when only the first `co` (< 16) output channels exist (the data gradient of a 4 -> 16 layer): row base `yrow`, pixel stride `co`
__device__ __forceinline__ void rs_mfma_store_narrow(const rs_f4& acc, float* yrow, int co) {
    const int lane = rs_tid() & 63, m = lane & 15, kq = lane >> 4;
    if (m < co) {
#pragma unroll
        for (int i = 0; i < 4; ++i) yrow[(4 * kq + i) * co + m] = acc[i];
    }
}
__device__ __forceinline__ void rs_mfma_sums(float sum, float sq, float* wsum) {
    const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t);
    sum += __shfl_xor(sum, 16, 64); sq += __shfl_xor(sq, 16, 64);
    sum += __shfl_xor(sum, 32, 64); sq += __shfl_xor(sq, 32, 64);
    if (lane < 16) { wsum[(wave * 2) * RS_MAXC + lane] = sum; wsum[(wave * 2 + 1) * RS_MAXC + lane] = sq; }
}

// cin <= 4 -> 16 channels on rows of 16 pixels: ONE v_mfma_f32_16x16x4_f32 per tap (K = the 4 input channels of the pixel's float4).  A: lane (pixel
// m = lane % 16, channel k = lane / 16) reads one float of the tile; B: w[co = lane % 16][ci = lane / 16][tap] straight from the OIHW filter (zero for
// ci >= cin).  Wave w of a sample owns the rows w, w + 8, ...: 18 MFMAs on the stock 16x16 maps.
template <int K>
__device__ __forceinline__ bool rs_conv_c4_mfma(const dcv_resident_op& o, const float4* xin, float* yt, float* wsum, const float* frag = nullptr, const float* bias_lds = nullptr) {
    const int H = o.h, pad = K / 2, XW = 16 + 2 * pad;
    const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t), m = lane & 15, kq = lane >> 4, s = wave >> 3;
    float breg[K * K];
    if (frag != nullptr) {                                                           // (parked in LDS two ops ago: resident_fwd_kernel's s_c4frag)
#pragma unroll
        for (int q = 0; q < K * K; ++q) breg[q] = frag[q * 64 + lane];
    } else {
        const float* wp = o.filt + ((size_t)m * o.cin + (kq < o.cin ? kq : 0)) * (K * K);
#pragma unroll
        for (int q = 0; q < K * K; ++q) breg[q] = wp[q];
        if (kq >= o.cin) {
#pragma unroll
            for (int q = 0; q < K * K; ++q) breg[q] = 0.f;
        }
    }
    const float bias = bias_lds ? bias_lds[m] : (o.bias ? o.bias[m] : 0.f);          // (the prologue left every layer's bias in LDS)
    const float* xf = reinterpret_cast<const float*>(xin);
    float sum = 0.f, sq = 0.f;
    for (int r = wave & 7; r < H; r += 8) {
        const float* xs = xf + ((size_t)s * RS_XIN + r * XW + m) * 4 + kq;
        rs_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[(ky * XW + kx) * 4], breg[ky * K + kx], acc, 0, 0, 0);
        rs_mfma_store(o, acc, bias, yt + s * RS_Y + (r * 16 + 4 * kq) * 16 + m, sum, sq);
    }
    rs_mfma_sums(sum, sq, wsum);
    return true;
}

// B fragments: lane (n = co = lane % 16, k = lane / 16) needs w[co][4k + j][tap], j < 4, tap < 9: 36 contiguous floats of the OIHW filter.  Requested by
// the PREVIOUS layer's apply step (a global round trip, 1 - 2 us, which nothing there waits for).
__device__ __forceinline__ void rs_load_bfrag(const float* w, float4 (&b4)[9]) {
    const int lane = rs_tid() & 63, m = lane & 15, kq = lane >> 4;
    const float4* wp = reinterpret_cast<const float4*>(w + ((size_t)m * 16 + 4 * kq) * 9);
#pragma unroll
    for (int q = 0; q < 9; ++q) b4[q] = wp[q];
}

// the same fragments from the prepared data-gradient layout [n 16][kq 4][j 4][tap 9] (dcv_conv2d_prepare_filters op 4)
__device__ __forceinline__ void rs_load_bfrag_lds(const float4* frag, float4 (&b4)[9]) {
    const int lane = rs_tid() & 63;
#pragma unroll
    for (int q = 0; q < 9; ++q) b4[q] = frag[q * 64 + lane];
}
__device__ __forceinline__ void rs_load_bfrag_prepared(const float* f, float4 (&b4)[9]) {
    const int lane = rs_tid() & 63, m = lane & 15, kq = lane >> 4;
    const float4* wp = reinterpret_cast<const float4*>(f + ((size_t)m * 4 + kq) * 36);
#pragma unroll
    for (int q = 0; q < 9; ++q) b4[q] = wp[q];
}

// 16 -> 16 channels, 3x3, rows of 16 pixels: per tap one ds_read_b128 (plane k = lane / 16: channels 4k .. 4k + 3 of pixel lane % 16) feeds four
// v_mfma_f32_16x16x4_f32 (MFMA j contracts channels {j, 4 + j, 8 + j, 12 + j}).
// `o.cout` < 16 (data gradient towards a 4-channel tensor): the B fragments of the missing channels are zero, only `o.cout` channels are stored.
__device__ __forceinline__ bool rs_conv_c16_mfma(const dcv_resident_op& o, const float4* xin, float* yt, const float4 (&b4)[9], float* wsum) {
    const int H = o.h, XW = 18, XH = H + 2;
    const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t), m = lane & 15, kq = lane >> 4, s = wave >> 3;
    float breg[36];
#pragma unroll
    for (int q = 0; q < 9; ++q) { breg[4 * q] = b4[q].x; breg[4 * q + 1] = b4[q].y; breg[4 * q + 2] = b4[q].z; breg[4 * q + 3] = b4[q].w; }
    const float bias = o.bias ? o.bias[m] : 0.f;
    float sum = 0.f, sq = 0.f;
    for (int r = wave & 7; r < H; r += 8) {
        const float4* xs = xin + s * RS_XIN + (kq * XH + r) * XW + m;
        rs_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 a = xs[ky * XW + kx];
                const int tap = ky * 3 + kx;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, breg[tap], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, breg[9 + tap], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, breg[18 + tap], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, breg[27 + tap], acc, 0, 0, 0);
            }
        if (o.cout == 16) rs_mfma_store(o, acc, bias, yt + s * RS_Y + (r * 16 + 4 * kq) * 16 + m, sum, sq);
        else rs_mfma_store_narrow(acc, yt + s * RS_Y + (size_t)r * 16 * o.cout, o.cout);
    }
    if (o.cout == 16) rs_mfma_sums(sum, sq, wsum);
    return true;
}

// ------------------------------------------------------------------------------------------------ the kernel

__global__ __launch_bounds__(RS_T) void resident_fwd_kernel(dcv_resident_program p, unsigned* ctl, rs_u64* gran) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* xin = reinterpret_cast<float4*>(smem);
    float* xinf = reinterpret_cast<float*>(smem);
    float* yt = reinterpret_cast<float*>(smem + RS_XIN_BYTES);
    __shared__ float s_wsum[16 * 2 * RS_MAXC];                                       // per-wave sums of y, y^2 left by the convolution epilogues
    __shared__ float s_samp[RS_SPB][2][RS_MAXC];                                     // per-sample sums of y, y^2
    __shared__ __attribute__((aligned(16))) float s_S[RS_SPB][RS_MAXC];              // layer output = S y + T
    __shared__ __attribute__((aligned(16))) float s_T[RS_SPB][RS_MAXC];
    __shared__ __attribute__((aligned(16))) float s_par[DCV_RESIDENT_MAX_OPS][8][RS_MAXC];   // per-channel parameters of every layer: bn gamma, beta, running mean, var, gn gamma, beta, convolution bias
    __shared__ __attribute__((aligned(16))) float s_filt[RS_FILT_OPS][RS_FILT_FLOATS];  // filters of the cin <= 4 -> 4 layers as [tap][co][ci] (rs_conv_c4_co4)
    __shared__ float s_c4frag[9 * 64];                                               // [q][lane]: rs_conv_c4_mfma's filter registers
    __shared__ float4 s_bfrag[9 * 64];                                               // MFMA B fragments of the first 16 -> 16 layer, [q][lane] (rs_load_bfrag's registers): fetched in the prologue's round trip — at the layer itself the filter has left the L2 and the fetch was 2 us
    __shared__ double s_tot[RS_MAXQ], s_red[4];                                      // batch totals of the current synchronisation
    __shared__ unsigned s_w32[2 * RS_MAXQ];
    __shared__ unsigned s_gen;
    __shared__ int s_fail;
    const int t = threadIdx.x;
    const int s_of_t = t >> 9, tt = t & 511;                                         // sample-major split of the block for the per-sample passes
    const int N = p.n, G = (int)gridDim.x;
    const bool valid0 = (int)blockIdx.x * RS_SPB < N, valid1 = (int)blockIdx.x * RS_SPB + 1 < N;
    // (wave 0 alone, the others behind a barrier: the scalar cache is the CU's, and in a hipGraph replay the argument segment is HOST memory — sixteen waves
    // requesting 40 lines each at once were 160 k PCIe reads per launch: 0.2 -> 0.9 ms per step of the hipGraph fallback)
    if (rs_wave_of(rs_tid()) == 0) rs_kernarg_warm<(int)(sizeof(dcv_resident_program) + 2 * sizeof(void*))>();
    rs_barrier();
    // the control words: requested here, first needed at the end of the prologue (the first stamp is taken unconditionally and written with the second one:
    // a stamp which compared block indices here would put the words' round trip in front of every load below)
    int lz = 0; asm volatile("" : "+v"(lz));                                         // (a zero the compiler takes for a per-lane value: uniform words would be moved to scalar registers here, i.e. waited for here)
    const unsigned ctl_mute = ctl[2 + lz], ctl_stamp = ctl[3 + lz];                  // ctl[2]: test hook (tests/test_resident_gpu.py): block ctl[2] - 1 never publishes -> every block's bounded spin gives up
    const unsigned gen0 = __hip_atomic_load((rs_gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const rs_u64 t_start = wall_clock64();
    int n_stamp = 1;
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(ctl + 4);
    // diagnostics: block ctl[3] records its phase times (tools/resident_stamps.py), block 0 by default
    auto stamp = [&]() {
        if (blockIdx.x == ctl_stamp && t == 0 && n_stamp < RS_MAX_STAMPS) { if (n_stamp == 1) stamps[1] = t_start; stamps[1 + n_stamp] = wall_clock64(); }
        ++n_stamp;
    };
    // MFMA B fragments of a 16 -> 16 layer travel two ops ahead: requested behind op oi's publish for op oi + 2 (the gather wait hides the trip), parked
    // in s_bfrag when op oi + 1 starts; bfrag_op: the op whose fragments are there
    int bfrag_op = -1, bfrag_pending = -1;
    // ... and the same for the per-lane filter registers of a 4 -> 16 layer on the matrix cores (rs_conv_c4_mfma: K * K floats per lane, read from memory at
    // the head of its convolution until then)
    int c4frag_op = -1, c4frag_pending = -1;
    float c4v = 0.f;
    float4 bfv = make_float4(0.f, 0.f, 0.f, 0.f);
    {   // ONE global round trip: every layer's per-channel parameters -> LDS (the coefficient phases never wait for memory), the two samples -> the first
        // layer's padded tile (loads issued before the tile is zeroed)
        // a replayed step's batch: read through the address the host left in the mailbox for THIS replay (pinned host memory: the longest round trip of this
        // prologue, requested first and — a vector load, counted in order — not waited for until the sample loads need it), and kept in the static tensor
        // for the backward pass
        const dcv_resident_op& o = p.op[0];
        const float* batch = o.x;
        rs_u64 batch_word = 0;
        const bool from_mailbox = p.mailbox != nullptr && p.batch == nullptr;
        // (a plain load through a per-lane-looking address: a vector load, counted in order.  As a SYSTEM-scope atomic load this word took 700 us per launch
        // when the step replayed as a hipGraph — and 4 us in the launch-list replay; pinned host memory is not cached on the device either way)
        if (from_mailbox) batch_word = *((const rs_gu64*)&p.mailbox->src[p.x_job] + lz);
        else if (p.mailbox != nullptr) batch = p.batch;
        // a wave's slots belong to ONE op (128 slots each): its descriptor is read through scalar loads (rs_kernarg_warm above) and the row's pointer picked
        // by selects — indexed per lane, the descriptor words, then the pointer, then the value were three dependent round trips ahead of everything below
        // The loads are unconditional (a lane without a parameter reads a control word) and their results are not touched before the commit behind the
        // barrier: under branches the loaded value meets its default in a copy right behind the load, i.e. a wait for everything in flight here.
        float pl[2], pd[2]; int pslot[2]; bool pok[2];                               // (up to 10 ops x 8 rows x 16 channels = 1280 slots: two per thread)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = t + k * RS_T;
            const int oi = __builtin_amdgcn_readfirstlane(e >> 7), oc = oi < p.count ? oi : 0;
            const int r = e & 127, f = r >> 4, c = r & 15;
            const dcv_resident_op& w = p.op[oc];
            // (the pointers made opaque: the compiler turns a select among struct fields back into a load indexed by f)
            const float* q0 = w.bn_gamma; const float* q1 = w.bn_beta; const float* q2 = w.running_mean; const float* q3 = w.running_var;
            const float* q4 = w.gn_gamma; const float* q5 = w.gn_beta; const float* q6 = w.bias;
            asm volatile("" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3), "+s"(q4), "+s"(q5), "+s"(q6));
            const float* src = f == 0 ? q0 : f == 1 ? q1 : f == 2 ? q2 : f == 3 ? q3 : f == 4 ? q4 : f == 5 ? q5 : q6;
            const bool slot = oi < p.count && w.kind == DCV_RES_LAYER && c < w.cout && f < 7;
            pslot[k] = slot ? e : -1;
            pd[k] = (f == 0 || f == 4) ? 1.f : 0.f;
            pok[k] = slot && src != nullptr;
            pl[k] = *(const rs_gf32*)(pok[k] ? src + c : reinterpret_cast<const float*>(ctl));
        }
        // ... the filters of the cin <= 4 -> 4 layers, element t of their [tap][co][ci] copy in LDS
        float fv[RS_FILT_OPS];
#pragma unroll
        for (int k = 0; k < RS_FILT_OPS; ++k) {
            fv[k] = 0.f;
            if (k < p.count && p.op[k].kind == DCV_RES_LAYER && rs_variant(p.op[k].cin, p.op[k].cout, p.op[k].k, p.op[k].w) == 1) fv[k] = rs_filt_fetch(p.op[k].filt, p.op[k].k, t);
        }
        // ... and one float of every 64-B line of every filter: the scalar loads of the convolutions then find them in L2 instead of HBM
        float warm[RS_WARM_LINES] = {};
        {
            // (nobody needs the values: they are looked at once, behind the barrier below — looked at here, each load waited for everything in flight)
            const int oi = __builtin_amdgcn_readfirstlane(t >> 6), line = t & 63;
            if (oi < p.count && p.op[oi].kind == DCV_RES_LAYER) {
                const dcv_resident_op& w = p.op[oi];
                const int floats = w.k * w.k * ((w.cin + 3) & ~3) * w.cout;
#pragma unroll
                for (int i = 0; i < RS_WARM_LINES; ++i) { const int f = (line + 64 * i) * 16; warm[i] = *(const rs_gf32*)(w.filt + (f < floats ? f : 0)); }
                for (int f = (line + 64 * RS_WARM_LINES) * 16; f < floats; f += 64 * 16) { const float v = w.filt[f]; asm volatile("" :: "v"(v)); }
            }
        }
        const int pad = o.k / 2, XW = o.w + 2 * pad, XH = o.h + 2 * pad, HW = o.h * o.w, C = o.cin;
        const float inv_hw = 1.f / (float)HW, inv_w = 1.f / (float)o.w, inv_c = 1.f / (float)C, inv_cv = 1.f / (float)((C >> 2) > 0 ? (C >> 2) : 1);
        const int n = (int)blockIdx.x * RS_SPB + s_of_t;
        constexpr int SB = 8;                                                        // elements per thread in flight
        const bool vec = !o.x_nchw && (C & 3) == 0;
        const bool planes = o.x_nchw && C <= 4 && HW <= 2 * 512;                     // the RGB model input: a thread gathers the channels of its two pixels
        const int total = planes ? 0 : (vec ? HW * (C >> 2) : HW * C);
        float4 v4[SB]; float v1[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) { v4[k] = make_float4(0.f, 0.f, 0.f, 0.f); v1[k] = 0.f; }
        if (from_mailbox) batch = reinterpret_cast<const float*>(batch_word);
        const bool keep = p.mailbox != nullptr && batch != o.x && n < N;
        const float* src = batch + (size_t)(n < N ? n : 0) * HW * C;
        float* stat = const_cast<float*>(o.x) + (size_t)(n < N ? n : 0) * HW * C;
        if (planes) {
            // v1[4 k + c] = channel c of pixel tt + 512 k: coalesced per plane, and ONE conflict-free float4 per pixel into the tile below (element-wise, the
            // 4-byte LDS stores of a channel-interleaved tile are 8-way bank conflicts)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) { const int px = tt + k * 512; v1[4 * k + c] = (c < C && px < HW) ? src[c * HW + px] : 0.f; }
        } else if (vec) {
#pragma unroll
            for (int k = 0; k < SB; ++k) { const int e = tt + k * 512; v4[k] = reinterpret_cast<const float4*>(src)[e < total ? e : 0]; }
        } else {
#pragma unroll
            for (int k = 0; k < SB; ++k) { const int e = tt + k * 512; v1[k] = src[e < total ? e : 0]; }
        }
        for (int i = t; i < RS_SPB * RS_XIN; i += RS_T) xin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t == 0) { s_gen = gen0; s_fail = 0; }
        rs_barrier();
#pragma unroll
        for (int k = 0; k < 2; ++k) if (pslot[k] >= 0) (&s_par[0][0][0])[pslot[k]] = pok[k] ? pl[k] : pd[k];
#pragma unroll
        for (int i = 0; i < RS_WARM_LINES; ++i) asm volatile("" :: "v"(warm[i]));
        if (t < RS_FILT_FLOATS) {
#pragma unroll
            for (int k = 0; k < RS_FILT_OPS; ++k) s_filt[k][t] = fv[k];
        }
        float* dst = xinf + (size_t)s_of_t * RS_XIN * 4;
        if (planes) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int px = tt + k * 512;
                if (px < HW && n < N) {
                    const int r = rs_fdiv(px, inv_w), col = px - r * o.w;
                    reinterpret_cast<float4*>(dst)[(r + pad) * XW + col + pad] = make_float4(v1[4 * k], v1[4 * k + 1], v1[4 * k + 2], v1[4 * k + 3]);
                    if (keep) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (c < C) stat[c * HW + px] = v1[4 * k + c];
                    }
                }
            }
        }
        // (the first SB * 512 elements of the sample were requested before the tile was zeroed; larger samples: further batches through arrays of their own —
        // one pair of arrays refilled inside the loop lived in scratch memory)
        auto commit = [&](int e0, const float4 (&c4)[SB], const float (&c1)[SB]) {
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int e = e0 + tt + k * 512;
                if (keep && e < total) { if (vec) reinterpret_cast<float4*>(stat)[e] = c4[k]; else stat[e] = c1[k]; }
                if (e < total && n < N) {
                    if (vec) {
                        const int px = rs_fdiv(e, inv_cv), cv = e - px * (C >> 2), r = rs_fdiv(px, inv_w), col = px - r * o.w;
                        reinterpret_cast<float4*>(dst)[(cv * XH + r + pad) * XW + col + pad] = c4[k];
                    } else {
                        int c, px;
                        if (o.x_nchw) { c = rs_fdiv(e, inv_hw); px = e - c * HW; } else { px = rs_fdiv(e, inv_c); c = e - px * C; }
                        const int r = rs_fdiv(px, inv_w), col = px - r * o.w;
                        dst[(((c >> 2) * XH + r + pad) * XW + col + pad) * 4 + (c & 3)] = c1[k];
                    }
                }
            }
        };
        if (total > 0) commit(0, v4, v1);
        for (int e0 = SB * 512; e0 < total; e0 += SB * 512) {
            float4 n4[SB]; float n1[SB];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int e = e0 + tt + k * 512;
                n4[k] = make_float4(0.f, 0.f, 0.f, 0.f); n1[k] = 0.f;
                if (vec) n4[k] = reinterpret_cast<const float4*>(src)[e < total ? e : 0]; else n1[k] = src[e < total ? e : 0];
            }
            commit(e0, n4, n1);
        }
    }
    rs_barrier();
    const unsigned gen = s_gen;
    stamp();
    bool conv_done = false;                                                          // this layer's convolution already ran (behind the previous apply step)
    bool sums_done = false;
    int fc_from_y = 0;                                                               // the FC input is the wide tensor a LINK left in yt
    bool fc_done = false;                                                            // the FC behind a LINK ran in the LINK's branch (its weights travelled during the LINK)
    for (int oi = 0; oi < p.count; ++oi) {
        const dcv_resident_op& o = p.op[oi];
        const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t), s_of_t = __builtin_amdgcn_readfirstlane(t >> 9), tt = t & 511;     // (per op: see rs_tid)
        if (c4frag_pending == oi + 1) { if (t < 9 * 64) s_c4frag[t] = c4v; c4frag_op = c4frag_pending; c4frag_pending = -1; }
        if (bfrag_pending == oi + 1) { if (t < 9 * 64) s_bfrag[t] = bfv; bfrag_op = bfrag_pending; bfrag_pending = -1; }     // (this op's barriers separate the stores from their readers)
        if (o.kind == DCV_RES_LAYER) {
            const int C = o.cout, CV = C >> 2, H = o.h, W = o.w, HW = H * W;
            // ---- convolution + bias + activation: xin -> yt, per-wave sums of y, y^2 -> s_wsum
            if (!conv_done) {
                const int var = rs_variant(o.cin, o.cout, o.k, o.w);
                if (var == 1) {
                    const float* wl = s_filt[oi < RS_FILT_OPS ? oi : 0];
                    const float* bl = o.bias ? s_par[oi][6] : nullptr;
                    if (o.k == 3) sums_done = rs_conv_c4_co4<3, 4, 0>(o, xin, yt, wl, bl, s_wsum);
                    else if (o.w != 32 || o.h != 32) sums_done = rs_conv_c4_co4<5, 4, 0>(o, xin, yt, wl, bl, s_wsum);
                    else sums_done = o.cin == 3 ? rs_conv_c4_co4_32<5, 3>(o, xin, yt, wl, bl, s_wsum) : rs_conv_c4_co4_32<5, 4>(o, xin, yt, wl, bl, s_wsum);
                }
                else if (var == 2) sums_done = o.k == 3 ? rs_conv_c4_wide<3>(o, xin, yt, s_wsum) : rs_conv_c4_wide<1>(o, xin, yt, s_wsum);
                else if (var == 4) { const float* fr = oi == c4frag_op ? s_c4frag : nullptr; const float* bl = o.bias ? s_par[oi][6] : nullptr; sums_done = o.k == 3 ? rs_conv_c4_mfma<3>(o, xin, yt, s_wsum, fr, bl) : rs_conv_c4_mfma<1>(o, xin, yt, s_wsum, fr, bl); }
                else { float4 b4[9]; if (oi == bfrag_op) rs_load_bfrag_lds(s_bfrag, b4); else rs_load_bfrag(o.filt, b4); sums_done = rs_conv_c16_mfma(o, xin, yt, b4, s_wsum); }
                rs_barrier();
            }
            conv_done = false;
            stamp();
            const bool norm = o.bn || o.groups;
            if (norm && !sums_done) {
                // (scalar wide variant) a pass over yt: wave w8 of the sample owns channel vector w8 % CV and every (8 / CV)-th stripe of 64 pixels
                const int w8 = wave & 7, cv = w8 & (CV - 1), stripes = 8 / CV, st0 = w8 / CV;
                const float4* ys = reinterpret_cast<const float4*>(yt + s_of_t * RS_Y);
                float sum[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
                for (int px = st0 * 64 + lane; px < HW; px += stripes * 64) {
                    const float4 v = ys[px * CV + cv];
                    sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
                    sq[0] += v.x * v.x; sq[1] += v.y * v.y; sq[2] += v.z * v.z; sq[3] += v.w * v.w;
                }
                for (int i = lane; i < 2 * RS_MAXC; i += 64) s_wsum[wave * 2 * RS_MAXC + i] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { sum[j] = rs_wave_sum(sum[j]); sq[j] = rs_wave_sum(sq[j]); }
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s_wsum[(wave * 2) * RS_MAXC + cv * 4 + j] = sum[j]; s_wsum[(wave * 2 + 1) * RS_MAXC + cv * 4 + j] = sq[j]; }
                }
                rs_barrier();
            }
            // ---- per-sample sums (8 waves each); BatchNorm: publish this block's 2C totals, gather every block's
            const unsigned epoch = gen * DCV_RESIDENT_MAX_OPS + (unsigned)oi + 1u;
            rs_u64* go = gran + (size_t)oi * RS_GRAN_OP;
            float mine = 0.f;
            if (norm && t < 2 * C) {
                const int q = t >= C ? 1 : 0, c = t - q * C;
                float v0 = 0.f, v1 = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) { v0 += s_wsum[(w8 * 2 + q) * RS_MAXC + c]; v1 += s_wsum[((8 + w8) * 2 + q) * RS_MAXC + c]; }
                s_samp[0][q][c] = v0; s_samp[1][q][c] = v1;
                mine = (valid0 ? v0 : 0.f) + (valid1 ? v1 : 0.f);
                if (o.sample_sums != nullptr) {                                      // for the backward launch's GroupNorm algebra (sum u, sum u^2 without a pass over y)
                    float* dst = o.sample_sums + ((size_t)blockIdx.x * RS_SPB * 2 + q) * C + c;
                    if (valid0) dst[0] = v0;
                    if (valid1) dst[2 * C] = v1;
                }
            }
            stamp();
            auto store_y = [&]() {                                                   // the sample's [HW][C] floats: the same order in LDS and in HBM
                const int n = (int)blockIdx.x * RS_SPB + s_of_t, v4 = HW * CV;
                if (n >= N) return;
                const float4* ys = reinterpret_cast<const float4*>(yt + s_of_t * RS_Y);
                float4* yg = reinterpret_cast<float4*>(o.y) + (size_t)n * v4;
                for (int e = tt; e < v4; e += 512) yg[e] = ys[e];
            };
            if (o.bn) {
                rs_allreduce_publish(go, epoch, 2 * C, mine, ctl_mute == blockIdx.x + 1u);
                if (oi + 2 < p.count && p.op[oi + 2].kind == DCV_RES_LAYER && rs_variant(p.op[oi + 2].cin, p.op[oi + 2].cout, p.op[oi + 2].k, p.op[oi + 2].w) == 3) {
                    const int l = t & 63, q = t >> 6;
                    if (t < 9 * 64) { const rs_f4 v = *((const __attribute__((address_space(1))) rs_f4*)p.op[oi + 2].filt + 9 * (4 * (l & 15) + (l >> 4)) + q); bfv = make_float4(v[0], v[1], v[2], v[3]); }
                    bfrag_pending = oi + 2;
                }
                if (oi + 2 < p.count && p.op[oi + 2].kind == DCV_RES_LAYER && rs_variant(p.op[oi + 2].cin, p.op[oi + 2].cout, p.op[oi + 2].k, p.op[oi + 2].w) == 4) {
                    const dcv_resident_op& f = p.op[oi + 2];
                    const int l = t & 63, q = t >> 6, fm = l & 15, fkq = l >> 4, kk = f.k * f.k;
                    c4v = 0.f;
                    if (t < 9 * 64 && q < kk && fkq < f.cin) c4v = *((const rs_gf32*)f.filt + ((size_t)fm * f.cin + fkq) * kk + q);
                    c4frag_pending = oi + 2;
                }
                if (oi == 0 && p.mailbox != nullptr) {
                    // the mailbox's other jobs (labels, optimizer scalars: a few KB) travel while the first layer's partial sums do: job j by a block which
                    // owns no quantity of the all-reduce (blocks 32 ..: nobody waits for them), or by block j when the grid is that small.  The reads of
                    // the pinned host block (PCIe) are on nobody's critical path; the next launch of the step is their first consumer.
                    const int count = p.mailbox->count < DCV_MAX_COPY_JOBS ? p.mailbox->count : DCV_MAX_COPY_JOBS;
                    for (int j = 0; j < count; ++j) {
                        if (j == p.x_job || (G >= RS_MAXQ + DCV_MAX_COPY_JOBS ? RS_MAXQ + j : j % G) != (int)blockIdx.x) continue;
                        const char* js = reinterpret_cast<const char*>(p.mailbox->src[j]);
                        char* jd = reinterpret_cast<char*>(p.mailbox->dst[j]);
                        const size_t bytes = (size_t)p.mailbox->bytes[j];
                        if (!js || !jd || js == jd) continue;
                        const bool v16 = (((reinterpret_cast<uintptr_t>(js) | reinterpret_cast<uintptr_t>(jd)) & 15) == 0);
                        const size_t n16 = v16 ? bytes / 16 : 0;
                        for (size_t i = t; i < n16; i += RS_T) reinterpret_cast<uint4*>(jd)[i] = reinterpret_cast<const uint4*>(js)[i];
                        for (size_t i = n16 * 16 + t; i < bytes; i += RS_T) jd[i] = js[i];
                    }
                }
                // y -> HBM (saved for the backward pass) needs no statistics: it travels while the partial sums do.  Blocks which own a quantity reduce first
                // (everybody waits for their totals), the others store first.
                const bool owner = (int)blockIdx.x < 2 * C;
                if (owner) rs_allreduce_reduce_owned(go, go + RS_MAXQ * RS_MAXG, epoch, 2 * C, G, s_red, &s_fail);
                store_y();
                if (!rs_allreduce_collect(go + RS_MAXQ * RS_MAXG, epoch, 2 * C, s_tot, s_w32, &s_fail)) {
                    // a block of this grid is not resident: report, poison the logits, leave
                    if (t == 0) { __hip_atomic_store((rs_gu32*)(ctl + 1), (unsigned)oi + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    for (int k = 0; k < p.count; ++k)
                        if (p.op[k].kind == DCV_RES_FC)
                            for (int e = t; e < RS_SPB * p.op[k].cout; e += RS_T) {
                                const int n = (int)blockIdx.x * RS_SPB + e / p.op[k].cout;
                                if (n < N) p.op[k].out[(size_t)n * p.op[k].cout + e % p.op[k].cout] = __builtin_nanf("");
                            }
                    if (blockIdx.x == 0 && t == 0) __hip_atomic_store((rs_gu32*)ctl, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (a later launch must not take this launch's granules for its own)
                    return;
                }
            } else { store_y(); rs_barrier(); }
            stamp();
            // ---- coefficients, all in wave 0: thread (sample, channel) sums the 16 partials of its channel, derives the BatchNorm affine v = a y + b, the
            // GroupNorm statistics of v from the sample's sums (norm_chain.hip, nc_apply_kernel's prologue; group totals by DPP exchanges between the
            // cpg adjacent lanes) and the layer's affine S y + T
            if (t < RS_SPB * C) {
                const int ss = t >= C ? 1 : 0, c = t - ss * C;
                float a = 1.f, b = 0.f;
                if (o.bn) {
                    const double s1 = s_tot[c], s2 = s_tot[C + c];
                    const double mean = s1 * o.inv_count;
                    double varc = s2 * o.inv_count - mean * mean; if (varc < 0.0) varc = 0.0;
                    const float rstd = (float)rs_rsqrt(varc + (double)o.bn_eps);
                    a = s_par[oi][0][c] * rstd; b = s_par[oi][1][c] - (float)mean * a;
                    if (blockIdx.x == 0 && ss == 0) {                                // saved / running statistics: written once
                        o.bn_mean[c] = (float)mean; o.bn_rstd[c] = rstd;
                        if (o.running_mean) o.running_mean[c] = (1.f - o.momentum) * s_par[oi][2][c] + o.momentum * (float)mean;
                        if (o.running_var) o.running_var[c] = (1.f - o.momentum) * s_par[oi][3][c] + o.momentum * (float)(varc * o.unbias);
                        if (c == 0 && o.num_batches_tracked) o.num_batches_tracked[0] += 1;
                    }
                }
                float S = a, T = b;
                if (o.groups > 0) {
                    const int cpg = (o.groups & (o.groups - 1)) == 0 ? C >> (31 - __builtin_clz(o.groups)) : C / o.groups;
                    const double hw = (double)HW, ad = (double)a, bd = (double)b, s1 = (double)s_samp[ss][0][c], s2 = (double)s_samp[ss][1][c];
                    double sv = ad * s1 + hw * bd, svv = ad * ad * s2 + 2.0 * ad * bd * s1 + hw * bd * bd;
                    for (int off = 1; off < cpg; off <<= 1) { sv += __shfl_xor(sv, off, 64); svv += __shfl_xor(svv, off, 64); }   // cpg: a power of two of adjacent lanes
                    const double mean = sv * o.inv_group;
                    double varc = svv * o.inv_group - mean * mean; if (varc < 0.0) varc = 0.0;
                    const float rstd = (float)rs_rsqrt(varc + (double)o.gn_eps);
                    const int n = (int)blockIdx.x * RS_SPB + ss, g = (cpg & (cpg - 1)) == 0 ? c >> (31 - __builtin_clz(cpg)) : rs_fdiv(c, 1.f / (float)cpg);
                    if (c == g * cpg && n < N) { o.gn_mean[(size_t)n * o.groups + g] = (float)mean; o.gn_rstd[(size_t)n * o.groups + g] = rstd; }
                    const float gg = s_par[oi][4][c], gb = s_par[oi][5][c];
                    S = a * rstd * gg; T = (b - (float)mean) * rstd * gg + gb;
                }
                s_S[ss][c] = S; s_T[ss][c] = T;
            }
            rs_barrier();
            stamp();
            // ---- apply: out = S y + T [averaged over pool windows] -> HBM, y -> HBM, the next op's input tile (interior here, its halo below)
            auto apply_step = [&]() {
                const int pk = o.pool > 0 ? o.pool : 1, Ho = H / pk, Wo = W / pk;
                const bool nxt_layer = oi + 1 < p.count && p.op[oi + 1].kind == DCV_RES_LAYER;
                if (pk == 1 && C == 4 && H == 32 && W == 32 && nxt_layer && p.op[oi + 1].k == 5) {
                    // the stock model's layers 1 and 2 (4 channels, 32 x 32, no pooling, a 5x5 layer next) with every extent a constant: two pixels per thread
                    // and 544 halo vectors per block; the general path below spends ~200 instructions per thread on six IEEE divisions, reciprocal index
                    // decodes and pooling loops of one iteration (2.0 us, issue-bound like every short phase of this launch)
                    constexpr int XWn = 36;
                    const int ss = s_of_t, n = (int)blockIdx.x * RS_SPB + ss;
                    const float4 S = *reinterpret_cast<const float4*>(s_S[ss]), T = *reinterpret_cast<const float4*>(s_T[ss]);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int e = tt + k * 512, r = e >> 5, c = e & 31;
                        const float4 v = reinterpret_cast<const float4*>(yt + ss * RS_Y)[e];
                        float4 r4 = make_float4(0.f + (v.x * S.x + T.x), 0.f + (v.y * S.y + T.y), 0.f + (v.z * S.z + T.z), 0.f + (v.w * S.w + T.w));      // (0 + ...: the general path's accumulator)
                        if (n < N) reinterpret_cast<float4*>(o.out)[(size_t)n * 1024 + e] = r4;
                        else r4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        xin[ss * RS_XIN + (r + 2) * XWn + c + 2] = r4;
                    }
                    if (t < RS_SPB * 272) {
                        const int hs = t >= 272 ? 1 : 0, e = t - hs * 272;
                        int r, c;
                        if (e < 144) { r = (e * 1821) >> 16; c = e - r * 36; if (r >= 2) r += 32; }             // (e / 36 for e < 144)
                        else { const int e2 = e - 144; r = 2 + (e2 >> 2); c = e2 & 3; if (c >= 2) c += 32; }
                        xin[hs * RS_XIN + r * XWn + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    return;
                }
                if (pk == 1 && C == 16 && H == 16 && W == 16 && nxt_layer && p.op[oi + 1].k == 3) {
                    // the stock model's layer 4 (16 channels, 16 x 16, a 3x3 layer next: four channel-vector planes of 18 x 18), same idea
                    const int ss = s_of_t, n = (int)blockIdx.x * RS_SPB + ss, cv = tt & 3;
                    const float4 S = *reinterpret_cast<const float4*>(s_S[ss] + cv * 4), T = *reinterpret_cast<const float4*>(s_T[ss] + cv * 4);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int e = tt + k * 512, px = e >> 2, r = px >> 4, c = px & 15;
                        const float4 v = reinterpret_cast<const float4*>(yt + ss * RS_Y)[e];
                        float4 r4 = make_float4(0.f + (v.x * S.x + T.x), 0.f + (v.y * S.y + T.y), 0.f + (v.z * S.z + T.z), 0.f + (v.w * S.w + T.w));
                        if (n < N) reinterpret_cast<float4*>(o.out)[(size_t)n * 1024 + e] = r4;
                        else r4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        xin[ss * RS_XIN + (cv * 18 + r + 1) * 18 + c + 1] = r4;
                    }
                    if (t < 8 * 68) {                                                // 8 planes x (2 x 18 + 16 x 2) halo vectors
                        const int pl = (t * 964) >> 16, e = t - pl * 68;             // (t / 68 for t < 544)
                        int r, c;
                        if (e < 36) { r = e >= 18 ? 17 : 0; c = e - (e >= 18 ? 18 : 0); }
                        else { const int e2 = e - 36; r = 1 + (e2 >> 1); c = (e2 & 1) ? 17 : 0; }
                        const int hs = pl >> 2, cvp = pl & 3;
                        xin[hs * RS_XIN + (cvp * 18 + r) * 18 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    return;
                }
                if (pk == 2 && ((C == 4 && H == 32 && W == 32 && nxt_layer && p.op[oi + 1].k == 3) || (C == 16 && H == 16 && W == 16 && !nxt_layer))) {
                    // the stock model's pooled layers: 3 (4 channels, 32 x 32 -> 16 x 16, one 18 x 18 plane for the 3x3 layer next) and 5 (16 channels,
                    // 16 x 16 -> 8 x 8, flat NHWC for the LINK): 256 outputs per sample, window elements accumulated in the general path's order
                    const int ss = s_of_t, n = (int)blockIdx.x * RS_SPB + ss;
                    const bool c4 = C == 4;
                    if (tt < 256) {
                        const int cv = c4 ? 0 : tt & 3, px = c4 ? tt : tt >> 2, ho = c4 ? px >> 4 : px >> 3, wo = c4 ? px & 15 : px & 7;
                        const float4 S = *reinterpret_cast<const float4*>(s_S[ss] + cv * 4), T = *reinterpret_cast<const float4*>(s_T[ss] + cv * 4);
                        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                            for (int kw = 0; kw < 2; ++kw) {
                                const int yi = c4 ? (ho * 2 + kh) * 32 + wo * 2 + kw : ((ho * 2 + kh) * 16 + wo * 2 + kw) * 4 + cv;
                                const float4 v = reinterpret_cast<const float4*>(yt + ss * RS_Y)[yi];
                                acc[0] += v.x * S.x + T.x; acc[1] += v.y * S.y + T.y; acc[2] += v.z * S.z + T.z; acc[3] += v.w * S.w + T.w;
                            }
                        float4 r4 = make_float4(acc[0] * 0.25f, acc[1] * 0.25f, acc[2] * 0.25f, acc[3] * 0.25f);
                        if (n < N) reinterpret_cast<float4*>(o.out)[(size_t)n * 256 + tt] = r4;
                        else r4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (c4) xin[ss * RS_XIN + (ho + 1) * 18 + wo + 1] = r4;
                        else reinterpret_cast<float4*>(xinf + (size_t)ss * RS_XIN * 4)[tt] = r4;
                    }
                    if (c4 && t < RS_SPB * 68) {
                        const int hs = t >= 68 ? 1 : 0, e = t - hs * 68;
                        int r, c;
                        if (e < 36) { r = e >= 18 ? 17 : 0; c = e - (e >= 18 ? 18 : 0); }
                        else { const int e2 = e - 36; r = 1 + (e2 >> 1); c = (e2 & 1) ? 17 : 0; }
                        xin[hs * RS_XIN + r * 18 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    return;
                }
                const int npad = nxt_layer ? p.op[oi + 1].k / 2 : 0, XWn = Wo + 2 * npad, XHn = Ho + 2 * npad;
                const float inv = 1.f / (float)(pk * pk), inv_cv = 1.f / (float)CV, inv_wo = 1.f / (float)Wo;
                const int per = Ho * Wo * CV;
                // waves 0 - 7: the first sample, 8 - 15: the second; a thread's channel vector is fixed (512 is a multiple of CV): S, T in registers
                const int ss = s_of_t, cv = tt & (CV - 1), n = (int)blockIdx.x * RS_SPB + ss;
                const float4 S = *reinterpret_cast<const float4*>(s_S[ss] + cv * 4), T = *reinterpret_cast<const float4*>(s_T[ss] + cv * 4);
                for (int e = tt; e < per; e += 512) {
                    const int px = rs_fdiv(e, inv_cv), ho = rs_fdiv(px, inv_wo), wo = px - ho * Wo;
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int kh = 0; kh < pk; ++kh)
                        for (int kw = 0; kw < pk; ++kw) {
                            const int yi = ((ho * pk + kh) * W + wo * pk + kw) * CV + cv;
                            const float4 v = reinterpret_cast<const float4*>(yt + ss * RS_Y)[yi];
                            acc[0] += v.x * S.x + T.x; acc[1] += v.y * S.y + T.y; acc[2] += v.z * S.z + T.z; acc[3] += v.w * S.w + T.w;
                        }
                    float4 r4;
                    if (o.pool > 0) r4 = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv); else r4 = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    if (n < N) reinterpret_cast<float4*>(o.out)[((size_t)n * Ho * Wo + px) * CV + cv] = r4;
                    else r4 = make_float4(0.f, 0.f, 0.f, 0.f);                       // (a block's missing second sample: keep its tile finite)
                    if (nxt_layer) xin[ss * RS_XIN + (cv * XHn + ho + npad) * XWn + wo + npad] = r4;
                    else reinterpret_cast<float4*>(xinf + (size_t)ss * RS_XIN * 4)[px * CV + cv] = r4;           // flat NHWC for a LINK / FC
                }
                if (nxt_layer && npad > 0) {
                    // zero the halo of the next layer's tile (disjoint from the interior written above): its geometry may differ from this layer's
                    const int rows_part = 2 * npad * XWn, halo = rows_part + Ho * 2 * npad, planes = RS_SPB * CV;
                    const float inv_halo = 1.f / (float)halo, inv_xw = 1.f / (float)XWn, inv_2p = 1.f / (float)(2 * npad);
                    for (int i = t; i < planes * halo; i += RS_T) {
                        const int pl = rs_fdiv(i, inv_halo), e = i - pl * halo;
                        int r, c;
                        if (e < rows_part) { r = rs_fdiv(e, inv_xw); c = e - r * XWn; if (r >= npad) r += Ho; }
                        else { const int e2 = e - rows_part; r = rs_fdiv(e2, inv_2p); c = e2 - r * 2 * npad; r += npad; if (c >= npad) c += Wo; }
                        const int ss = pl >= CV ? 1 : 0, cvp = pl - ss * CV;
                        xin[ss * RS_XIN + (cvp * XHn + r) * XWn + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            };
            if (oi + 1 < p.count && p.op[oi + 1].kind == DCV_RES_LAYER && rs_variant(p.op[oi + 1].cin, p.op[oi + 1].cout, p.op[oi + 1].k, p.op[oi + 1].w) == 3) {
                // the next layer runs on the matrix cores: its B fragments are requested behind this step, and its convolution runs right here so that the
                // 36 registers do not live across the op loop
                apply_step();
                float4 b4[9];
                if (oi + 1 == bfrag_op) rs_load_bfrag_lds(s_bfrag, b4);
                else rs_load_bfrag(p.op[oi + 1].filt, b4);  // (requested in FRONT of the apply step, the 36 registers slowed the general step's loops from 2.2 to 5.3 us and spill in the constant-extent one)
                rs_barrier();
                stamp();
                sums_done = rs_conv_c16_mfma(p.op[oi + 1], xin, yt, b4, s_wsum);
                rs_barrier();
                conv_done = true;
            } else {
                apply_step();
                rs_barrier();
                stamp();
            }
            fc_from_y = 0;
        } else if (o.kind == DCV_RES_LINK) {
            // wide[px][0 .. cin) = current (flat NHWC in xin), wide[px][cin ..) = bilinear resize of ref to (h, w): -> HBM and yt (flat NHWC)
            const int Cc = o.cin, CVc = Cc >> 2, Cr = o.ref_c, Cw = Cc + Cr, HW = o.h * o.w, perw = HW * Cw;
            const float sh = o.align_corners ? (o.h > 1 ? (float)(o.ref_h - 1) / (float)(o.h - 1) : 0.f) : (float)o.ref_h / (float)o.h;
            const float sw = o.align_corners ? (o.w > 1 ? (float)(o.ref_w - 1) / (float)(o.w - 1) : 0.f) : (float)o.ref_w / (float)o.w;
            // a fully connected op behind this one: wave j owns output j for BOTH samples, and the first 5 x 64 float4 of its weight row and its bias are
            // requested here — they travel while the link is computed (in the op's own branch below, the round trip was the first half of its 2.4 us)
            constexpr int FCB = 5;
            const bool fc_next = oi + 1 < p.count && p.op[oi + 1].kind == DCV_RES_FC;
            float4 fcw[FCB]; float fcb = 0.f;
#pragma unroll
            for (int u = 0; u < FCB; ++u) fcw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fc_next && wave < p.op[oi + 1].cout) {
                const dcv_resident_op& f = p.op[oi + 1];
                const int K4 = (f.h * f.w * f.cin) >> 2;
                const float4* wr = reinterpret_cast<const float4*>(f.filt) + (size_t)wave * K4;
#pragma unroll
                for (int u = 0; u < FCB; ++u) { const int k = lane + 64 * u; fcw[u] = wr[k < K4 ? k : 0]; }
                if (f.bias) fcb = f.bias[wave];
            }
            __syncthreads();                                                         // full: every wave's earlier global stores (the referenced tensor) are done
            {   // referenced channels: one item per (sample, pixel, channel); loads first
                const int per = HW * Cr;
                const float inv_cr = 1.f / (float)Cr, inv_w = 1.f / (float)o.w;
                for (int it = t; it < RS_SPB * per; it += RS_T) {
                    const int ss = it >= per ? 1 : 0, e = it - ss * per, px = rs_fdiv(e, inv_cr), cr = e - px * Cr, ho = rs_fdiv(px, inv_w), wo = px - ho * o.w;
                    const int n = (int)blockIdx.x * RS_SPB + ss;
                    if (n >= N) continue;
                    float fs = o.align_corners ? sh * ho : sh * (ho + 0.5f) - 0.5f; if (fs < 0.f) fs = 0.f;
                    int h0 = (int)fs; if (h0 > o.ref_h - 1) h0 = o.ref_h - 1;
                    const int h1 = h0 + (h0 < o.ref_h - 1 ? 1 : 0); const float lh = fs - (float)h0;
                    float fw = o.align_corners ? sw * wo : sw * (wo + 0.5f) - 0.5f; if (fw < 0.f) fw = 0.f;
                    int w0 = (int)fw; if (w0 > o.ref_w - 1) w0 = o.ref_w - 1;
                    const int w1 = w0 + (w0 < o.ref_w - 1 ? 1 : 0); const float lw = fw - (float)w0;
                    // `ref` was stored by this block earlier in this launch (its own samples): agent-scope loads, never the CU's L1
                    const float* rs = o.ref + (size_t)n * o.ref_h * o.ref_w * Cr + cr;
                    auto ld = [&](int hh, int ww) { return __uint_as_float(__hip_atomic_load((rs_gu32*)(rs + ((size_t)hh * o.ref_w + ww) * Cr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
                    const float x00 = ld(h0, w0), x01 = ld(h0, w1), x10 = ld(h1, w0), x11 = ld(h1, w1);
                    const float c00 = (1.f - lh) * (1.f - lw), c01 = (1.f - lh) * lw, c10 = lh * (1.f - lw), c11 = lh * lw;
                    const float v = c00 * x00 + c01 * x01 + c10 * x10 + c11 * x11;
                    yt[ss * RS_Y + px * Cw + Cc + cr] = v;
                    o.out[(size_t)n * perw + px * Cw + Cc + cr] = v;
                }
            }
            {   // current channels: float4 copies
                const int per = HW * CVc;
                const float inv_cv = 1.f / (float)CVc;
                for (int it = t; it < RS_SPB * per; it += RS_T) {
                    const int ss = it >= per ? 1 : 0, e = it - ss * per, px = rs_fdiv(e, inv_cv), cv = e - px * CVc;
                    const int n = (int)blockIdx.x * RS_SPB + ss;
                    if (n >= N) continue;
                    const float4 v = reinterpret_cast<const float4*>(xinf + (size_t)ss * RS_XIN * 4)[e];
                    *reinterpret_cast<float4*>(yt + ss * RS_Y + px * Cw + cv * 4) = v;
                    *reinterpret_cast<float4*>(o.out + (size_t)n * perw + px * Cw + cv * 4) = v;
                }
            }
            rs_barrier();
            stamp();
            fc_from_y = 1;
            if (fc_next) {
                const dcv_resident_op& f = p.op[oi + 1];
                const int K4 = (f.h * f.w * f.cin) >> 2, O = f.cout;
                if (wave < O) {
                    const float4* x0 = reinterpret_cast<const float4*>(yt), * x1 = reinterpret_cast<const float4*>(yt + RS_Y);
                    const float4* wr = reinterpret_cast<const float4*>(f.filt) + (size_t)wave * K4;
                    float a0 = 0.f, a1 = 0.f;
                    for (int k0 = lane; k0 < K4; k0 += 64 * FCB) {
#pragma unroll
                        for (int u = 0; u < FCB; ++u) {
                            const int k = k0 + 64 * u;
                            if (k < K4) {
                                const float4 w4 = fcw[u], p0 = x0[k], p1 = x1[k];
                                a0 += p0.x * w4.x + p0.y * w4.y + p0.z * w4.z + p0.w * w4.w;
                                a1 += p1.x * w4.x + p1.y * w4.y + p1.z * w4.z + p1.w * w4.w;
                            }
                        }
                        if (k0 + 64 * FCB < K4) {                                    // (rows longer than 5 x 64 vectors: the next batch, now with its round trip exposed)
#pragma unroll
                            for (int u = 0; u < FCB; ++u) { const int k = k0 + 64 * (FCB + u); fcw[u] = wr[k < K4 ? k : 0]; }
                        }
                    }
                    a0 = rs_wave_sum(a0); a1 = rs_wave_sum(a1);
                    const int n0 = (int)blockIdx.x * RS_SPB;
                    if (lane == 0 && n0 < N) f.out[(size_t)n0 * O + wave] = dcv_act_apply(a0 + fcb, f.act, f.slope);
                    if (lane == 1 && n0 + 1 < N) f.out[(size_t)(n0 + 1) * O + wave] = dcv_act_apply(a1 + fcb, f.act, f.slope);
                }
                rs_barrier();
                stamp();
                fc_done = true;
            }
        } else if (o.kind == DCV_RES_FC && fc_done) {
            // (computed in the LINK's branch)
        } else if (o.kind == DCV_RES_FC) {
            // logits[n][j] = act(bias[j] + sum_k x[n][k] w[j][k]), K = h * w * cin floats of the sample's flat NHWC activations; 32 lanes per (sample, j)
            const int K4 = (o.h * o.w * o.cin) >> 2, O = o.cout;
            const float4* xs = fc_from_y ? reinterpret_cast<const float4*>(yt + s_of_t * RS_Y) : reinterpret_cast<const float4*>(xinf + (size_t)s_of_t * RS_XIN * 4);
            const int j = tt >> 5, l32 = tt & 31, n = (int)blockIdx.x * RS_SPB + s_of_t;
            float acc = 0.f;
            if (j < O) {
                const float4* wr = reinterpret_cast<const float4*>(o.filt) + (size_t)j * K4;
                constexpr int FB = 10;                                               // weight vectors in flight per lane (the stock head: all of them)
                for (int k0 = l32; k0 < K4; k0 += 32 * FB) {
                    float4 b[FB];
#pragma unroll
                    for (int u = 0; u < FB; ++u) { const int k = k0 + 32 * u; b[u] = wr[k < K4 ? k : 0]; }
#pragma unroll
                    for (int u = 0; u < FB; ++u) {
                        const int k = k0 + 32 * u;
                        if (k < K4) { const float4 a = xs[k]; acc += a.x * b[u].x + a.y * b[u].y + a.z * b[u].z + a.w * b[u].w; }
                    }
                }
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (j < O && l32 == 0 && n < N) o.out[(size_t)n * O + j] = dcv_act_apply(acc + (o.bias ? o.bias[j] : 0.f), o.act, o.slope);
            rs_barrier();
            stamp();
        }
    }
    // every block read `gen` before its first publication and block 0 only gets here after it has gathered everybody's: the next launch is stream-ordered
    if (blockIdx.x == ctl_stamp && t == 0) stamps[0] = (unsigned long long)n_stamp;
    if (blockIdx.x == 0 && t == 0) __hip_atomic_store((rs_gu32*)ctl, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------ the backward chain
// LDS map (dynamic): DZ [RS_SPB][RS_XIN] float4 (the zero-padded dz tile the data gradient reads; between a data gradient and the next dz it holds the
// gather scratch) | Y [RS_SPB][RS_Y] (the layer's saved y) | DO [RS_SPB][RS_Y] (dout of the layer; then its data gradient = the next layer's dout)
// | XT [RS_SPB][RS_XIN] float4 (the layer's zero-padded input x, for its weight gradient).
constexpr size_t RB_SMEM = 2 * RS_XIN_BYTES + 2 * RS_Y_BYTES;          // + XT: the zero-padded input tile of the layer whose weight gradient is pending

// data-gradient variant of a layer (forward cin -> cout, k): the forward kernels on the swapped channel counts
__host__ __device__ __forceinline__ int rb_dgrad_variant(int cin, int cout, int k, int w) {
    if (cout <= 4 && cin == 4 && (k == 3 || k == 5)) return 1;                       // 4 <- 4: VALU row pairs
    if (cout == 16 && cin == 4 && k == 3 && w == 16) return 4;                       // 4 <- 16: VALU, scalar-cache filter (on the matrix cores 3/4 of the MACs multiply zeros)
    if (cout == 16 && (cin == 16 || cin == 4) && k == 3 && w == 16) return 3;        // 16 / 4 <- 16: matrix cores
    return 0;
}

// 4 <- 16 channels, 3x3, 16 x 16 pixels (the data gradient of a 4 -> 16 layer): thread = (half of the 16 dz channels, sample, pixel); per (tap, channel
// vector) one ds_read_b128 of dz and 16 wave-uniform filter scalars ([tap][dz channel 16][x channel 4]: dcv_conv2d_prepare_filters op 4, direct
// layout) feed 16 FMAs; the two halves' partial sums meet in the DO region.  295 k FMAs per block instead of the 1.2 M MACs of the 16 <- 16 matrix-core tile.
__device__ __forceinline__ void rb_dgrad_c16_co4(const float* filt, const float4* dzt, float* dor) {
    constexpr int XH = 18, XW = 18;
    const int t = rs_tid(), hv = __builtin_amdgcn_readfirstlane(t >> 9), tt = t & 511, ss = __builtin_amdgcn_readfirstlane(tt >> 8), px = tt & 255, r = px >> 4, c = px & 15;
    rs_cfloat* wl = RS_CONST(filt);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
        float4 dv[2][3];
#pragma unroll
        for (int cvl = 0; cvl < 2; ++cvl)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) dv[cvl][kx] = dzt[ss * RS_XIN + ((2 * hv + cvl) * XH + r + ky) * XW + c + kx];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int cvl = 0; cvl < 2; ++cvl) {
                rs_cfloat* wt = wl + ((ky * 3 + kx) * 16 + (2 * hv + cvl) * 4) * 4;
#pragma unroll
                for (int co = 0; co < 4; ++co) {
                    acc[co] = fmaf(dv[cvl][kx].x, wt[co], acc[co]);
                    acc[co] = fmaf(dv[cvl][kx].y, wt[4 + co], acc[co]);
                    acc[co] = fmaf(dv[cvl][kx].z, wt[8 + co], acc[co]);
                    acc[co] = fmaf(dv[cvl][kx].w, wt[12 + co], acc[co]);
                }
            }
    }
    float4* out = reinterpret_cast<float4*>(dor + ss * RS_Y + hv * 1024);          // half 1: parked behind the sample's 256 x 4 results
    out[px] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    rs_barrier();
    if (t < 512) {
        float4* o4 = reinterpret_cast<float4*>(dor + ss * RS_Y);
        const float4 a = o4[px], b = o4[256 + px];
        o4[px] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// ---- weight gradients inside the backward launch.  A wave takes whole UNITS = (filter tap or filter row, sample slot) and walks all pixels of the
// sample for it: nothing is reduced across waves, a unit's result goes straight to its chunk (2 * block + sample slot) of the layer's partial slab, and the
// step's one dcv_slab_reduce sums the chunks.  xt: the layer's zero-padded input tile (same layout as the forward's), dzt: its zero-padded dz tile.
__host__ __device__ __forceinline__ int rb_wgrad_variant(int cin, int cout, int k, int w) {
    if (cin <= 4 && cout == 4 && (k == 3 || k == 5) && w <= 32) return 1;            // VALU, lanes = image columns (rb_wgrad_c4_rows_ok: the rows)
    if (cin <= 4 && cout == 16 && k == 3 && w == 16) return 2;                       // MFMA, M = (filter column, input channel)
    if (cin == 16 && cout == 16 && k == 3 && w == 16) return 3;                      // MFMA, M = input channel
    return 0;
}

// rows of the maps rb_wgrad_c4 covers: equal row groups, and for K = 5 row groups which split into four quarters
__host__ __device__ __forceinline__ bool rb_wgrad_c4_rows_ok(int h, int w, int k) {
    const int rg = (w > 0 && (w & (w - 1)) == 0) ? 32 / w : 1;
    return w > 0 && w <= 32 && rg > 0 && h % rg == 0 && (k != 5 || (h / rg) % 4 == 0);
}

// cin <= 4 -> 4, K x K on maps of at most 32 columns.  UNIT = (filter column kx, output channel co) for BOTH samples of the block: lanes 0 - 31 work on
// the first sample, lanes 32 - 63 on the second; a lane owns one image column (maps narrower than 32: 32 / W row groups side by side) and walks DOWN its
// rows.  Walking down, the K input vectors x[r + ky][c + kx] of a step are a sliding window: one new ds_read_b128 of x and one of dz per step feed K * 4
// FMAs into the lane's [ky][ci] accumulators — a fifth of the LDS traffic of a unit per (filter row, channel pair) (6 ds_read_b128 per 40 FMAs: with 16
// waves that is 24 LDS-array cycles against 20 VALU cycles per step and SIMD, the phase was LDS-bound), no index arithmetic inside the loop (rows are a
// constant stride apart), and the window rotates through register NAMES (the loop is unrolled by K), not through moves.  K = 5: the 16 units of the filter
// columns 0 - 3 go one to each wave; the four units of column 4 are cut into four row ranges each (16 quarter units, one per wave) whose partial sums meet
// in LDS (`scratch`, 2 * 16 * K * 4 floats) in a fixed order — every wave does 1.25 units, none idles.  K = 3: 12 units, one round.  A unit's totals over
// its 32 lanes come from rs_reduce8 without the last (cross-sample) exchange.  The units of filter column 0 also sum dz (bias gradient).
typedef float rs_f2 __attribute__((ext_vector_type(2)));
template <int CO> __device__ __forceinline__ float rb_comp(const rs_f4& v) { return v[CO]; }

// one unit over rows [0, nrows) relative to xs / ds (xs: x tile at (first row, column + kx); ds: dz tile interior at (first row, column)): whole groups
// of K rows with the window rotating through register names, the remaining rows one at a time with a shifting window.  XW: the tiles' row stride in
// float4 — a compile-time constant makes every row offset of a group an immediate of its ds_read (one address register per tile, advanced once per group).
// Accumulators are explicit pairs (v_pk_fma_f32 with the dz value broadcast): left to the SLP vectoriser, the unrolled group turned into packed
// multiplies across steps plus separate adds, and 269 spilled registers.
template <int K, int CI, int CO>
__device__ __forceinline__ void rb_wg_unit(const rs_f4* xs, const rs_f4* ds, const int XW, int nrows, bool live, rs_f2 (&acc)[K][2], float& bsum) {
    // window slots: the K rows of the step + the row requested for the next one; K = 5: eight, so that groups of eight steps tile the stock maps' 32 rows and the
    // 8-row quarter units exactly (the shifting-window tail below costs 20 moves per step)
    constexpr int S = K == 5 ? 8 : K + 1;
    rs_f4 w[S];
#pragma unroll
    for (int j = 0; j < K; ++j) w[j] = xs[j * XW];
    rs_f4 dn = ds[0];
    auto fma_rows = [&](const rs_f4 (&win)[S], int j0, float d) {
        const rs_f2 d2 = {d, d};
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const rs_f4 xv = win[(j0 + ky) % S];
            acc[ky][0] = __builtin_elementwise_fma(xv.xy, d2, acc[ky][0]);
            if constexpr (CI > 3) acc[ky][1] = __builtin_elementwise_fma(xv.zw, d2, acc[ky][1]);
            else acc[ky][1].x = fmaf(xv.z, d, acc[ky][1].x);
        }
    };
    int r0 = 0;
    for (; r0 + S <= nrows; r0 += S) {
#pragma unroll
        for (int j = 0; j < S; ++j) {
            // both operands of the NEXT step are requested first (its x row goes into the slot of the row which left the window): an LDS round trip later,
            // when this step's FMAs are done, they have landed
            const bool more = j + 1 < S || r0 + S < nrows;
            w[(j + K) % S] = xs[(more ? j + K : j + K - 1) * XW];
            const rs_f4 dc = dn;
            dn = ds[(more ? j + 1 : j) * XW];
            float d = rb_comp<CO>(dc);
            if (!live) d = 0.f;
            bsum += d;
            fma_rows(w, j, d);
        }
        xs += S * XW; ds += S * XW;
    }
    for (; r0 < nrows; ++r0) {                                                       // at most K rows: rows r0 .. r0 + K - 1 sit in slots 0 .. K - 1
        const bool more = r0 + 1 < nrows;
        w[K] = xs[(more ? K : K - 1) * XW];
        const rs_f4 dc = dn;
        dn = ds[(more ? 1 : 0) * XW];
        float d = rb_comp<CO>(dc);
        if (!live) d = 0.f;
        bsum += d;
        fma_rows(w, 0, d);
#pragma unroll
        for (int j = 0; j < K; ++j) w[j] = w[j + 1];
        xs += XW; ds += XW;
    }
}

// rs_reduce8 over the 32 lanes of each half of the wave: lanes 0 - 3 hold the first half's eight totals (as rs_reduce8), lanes 32 - 35 the second half's
__device__ __forceinline__ void rs_reduce8_half(const float (&v)[8], float (&r)[2]) {
    const int lane = rs_tid() & 63;
    const bool b0 = lane & 1, b1 = lane & 2;
    float a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float mine = b0 ? v[4 + i] : v[i], send = b0 ? v[i] : v[4 + i]; a[i] = mine + rs_dpp_mov<0xB1>(send); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const float mine = b1 ? a[2 + i] : a[i], send = b1 ? a[i] : a[2 + i]; b[i] = mine + rs_dpp_mov<0x4E>(send); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        b[i] += rs_dpp_mov<0x124>(b[i]);        // row_ror:4
        b[i] += rs_dpp_mov<0x128>(b[i]);        // row_ror:8
        b[i] += __shfl_xor(b[i], 16, 64);
        r[i] = b[i];
    }
}

// `pass` 0: the wave's own unit (filter column wave / 4, all rows); pass 1 (K = 5): its quarter of the last filter column's unit -> scratch
template <int K, int CI, int CO, int XWC>
__device__ __forceinline__ void rb_wgrad_c4_units(const dcv_resident_bwd_op& o, const float4* xt, const float4* dzt, int wave, float* scratch) {
    constexpr int pad = K / 2, FULL = 4 * K <= 16 ? 4 * K : 16;
    const int W = XWC > 0 ? XWC - 2 * pad : o.w, XW = XWC > 0 ? XWC : W + 2 * pad, H = o.h;
    const bool pow2 = (W & (W - 1)) == 0;
    const int RG = pow2 ? 32 / W : 1, RPG = H / RG;                                  // row groups side by side in a half-wave; rows per group (H % RG == 0)
    const int chunk0 = (int)blockIdx.x * RS_SPB, wsize = o.cout * o.cin * K * K;
#pragma unroll 1
    for (int pass = (wave < FULL ? 0 : 1); pass < (4 * K > 16 ? 2 : 1); ++pass) {
        const int kx = pass == 0 ? wave >> 2 : K - 1, q = wave >> 2;
        const int nrows = pass == 0 ? RPG : RPG >> 2, row0 = pass == 0 ? 0 : q * nrows;
        const int lane = rs_tid() & 63, s = lane >> 5, l32 = lane & 31;
        const int g = pow2 ? l32 / W : 0, c = l32 - g * W;
        const bool live = l32 < RG * W;
        const int first = live ? g * RPG + row0 : 0, col = live ? c : 0;
        rs_f2 acc[K][2];
        float bsum = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) { acc[ky][0] = rs_f2{0.f, 0.f}; acc[ky][1] = rs_f2{0.f, 0.f}; }
        rb_wg_unit<K, CI, CO>(reinterpret_cast<const rs_f4*>(xt) + s * RS_XIN + first * XW + col + kx, reinterpret_cast<const rs_f4*>(dzt) + s * RS_XIN + (first + pad) * XW + col + pad,
                              XW, nrows, live, acc, bsum);
        // totals per half-wave; value index of a call = (ky of the pair) * 4 + ci.  Lanes 0 - 3 of each half keep two totals per call: their destinations are
        // computed ONCE here (the per-call address arithmetic of the first version was a third of this epilogue's instructions)
        const int vi = (l32 & 1) * 4 + ((l32 >> 1) & 1) * 2, ky0 = vi >> 2, ci = vi & 3;
        const int cin = o.cin, ky_stride = K * o.cout * cin;                                       // floats between filter rows of the [tap][co][ci] slab
        float* dst = o.dw_part + (size_t)(chunk0 + s) * wsize + ((ky0 * K + kx) * o.cout + CO) * cin + ci;
        float* sc = scratch + ((q * RS_SPB + s) * 4 + CO) * (K * 4) + ky0 * 4 + ci;
        const bool keep = l32 < 4, st0 = keep && ci < cin, st1 = keep && ci + 1 < cin;
#pragma unroll
        for (int k2 = 0; k2 < (K + 1) / 2; ++k2) {
            float v[8], r2[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = acc[2 * k2][i >> 1][i & 1]; v[4 + i] = 2 * k2 + 1 < K ? acc[2 * k2 + 1 < K ? 2 * k2 + 1 : 0][i >> 1][i & 1] : (i == 0 ? bsum : 0.f); }
            rs_reduce8_half(v, r2);
            if (2 * k2 + 1 < K || ky0 == 0) {                                                       // (compile time for all but the last pair of an odd K)
                if (pass == 1) { if (keep) { sc[2 * k2 * 4] = r2[0]; sc[2 * k2 * 4 + 1] = r2[1]; } }
                else {
                    if (st0) dst[2 * k2 * ky_stride] = r2[0];
                    if (st1) dst[2 * k2 * ky_stride + 1] = r2[1];
                }
            } else if (keep && ci == 0 && kx == 0 && pass == 0 && o.db_part) o.db_part[(size_t)(chunk0 + s) * o.cout + CO] = r2[0];   // (odd K: the spare slot of the last pair carries the sum of dz)
        }
    }
}

template <int K, int CI, int XWC>
__device__ __forceinline__ void rb_wgrad_c4(const dcv_resident_bwd_op& o, const float4* xt, const float4* dzt, float* scratch) {
    const int t = rs_tid(), wave = rs_wave_of(t), co = wave & 3;
    if (co == 0) rb_wgrad_c4_units<K, CI, 0, XWC>(o, xt, dzt, wave, scratch);
    else if (co == 1) rb_wgrad_c4_units<K, CI, 1, XWC>(o, xt, dzt, wave, scratch);
    else if (co == 2) rb_wgrad_c4_units<K, CI, 2, XWC>(o, xt, dzt, wave, scratch);
    else rb_wgrad_c4_units<K, CI, 3, XWC>(o, xt, dzt, wave, scratch);
    if constexpr (4 * K > 16) {
        static_assert(4 * K == 20, "the split of the last filter column assumes K = 5");
        rs_barrier();
        if (t < RS_SPB * 4 * K * 4) {                                              // (sample, co, ky, ci): the four quarters in a fixed order
            const int ss = t / (4 * K * 4), e = t - ss * (4 * K * 4), c2 = e / (K * 4), f = e - c2 * (K * 4), ky = f >> 2, ci = f & 3;
            const float* sc = scratch + (ss * 4 + c2) * (K * 4) + f;
            constexpr int QS = RS_SPB * 4 * K * 4;
            const float v = (sc[0] + sc[QS]) + (sc[2 * QS] + sc[3 * QS]);
            const int chunk = (int)blockIdx.x * RS_SPB + ss, wsize = o.cout * o.cin * K * K;
            if (ci < o.cin) o.dw_part[(size_t)chunk * wsize + ((ky * K + (K - 1)) * o.cout + c2) * o.cin + ci] = v;
        }
    }
}

// sum of dz per channel (bias gradient) of a 16-channel layer: unit = sample, lanes = pixels x channel vectors
__device__ __forceinline__ void rb_bias16(const dcv_resident_bwd_op& o, const float4* dzt, int unit_wave0) {
    const int tb = rs_tid();
    const int H = o.h, XW = 18, XH = H + 2, lane = tb & 63, wave = rs_wave_of(tb);
    if (!o.db_part) return;
    for (int s = 0; s < RS_SPB; ++s) {
        if (wave != ((unit_wave0 + s) & 15)) continue;
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int px = lane; px < H * 16; px += 64) {
            const int r = px >> 4, c = px & 15;
#pragma unroll
            for (int cv = 0; cv < 4; ++cv) { const float4 d = dzt[s * RS_XIN + (cv * XH + r + 1) * XW + c + 1]; acc[cv * 4] += d.x; acc[cv * 4 + 1] += d.y; acc[cv * 4 + 2] += d.z; acc[cv * 4 + 3] += d.w; }
        }
        float lo[8], hi[8], rl[2], rh[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) { lo[i] = acc[i]; hi[i] = acc[8 + i]; }
        rs_reduce8(lo, rl); rs_reduce8(hi, rh);
        if (lane < 4) {
            const int vi = (lane & 1) * 4 + ((lane >> 1) & 1) * 2, chunk = (int)blockIdx.x * RS_SPB + s;
            o.db_part[(size_t)chunk * 16 + vi] = rl[0]; o.db_part[(size_t)chunk * 16 + vi + 1] = rl[1];
            o.db_part[(size_t)chunk * 16 + 8 + vi] = rh[0]; o.db_part[(size_t)chunk * 16 + 8 + vi + 1] = rh[1];
        }
    }
}

// acc += A B over the pixel quads of rows [r0, r1) of a 16-pixel-wide map: the 16 operand words of two rows are requested before their 8 MFMAs (a loop of
// load, load, MFMA is a chain of LDS latencies: 44 % of the matrix pipe)
__device__ __forceinline__ void rb_mfma_rows(const float* xa, const float* db, int r0, int r1, int XW, rs_f4& acc) {
    int r = r0;
    rs_f4 acc2 = {0.f, 0.f, 0.f, 0.f};                                               // (two chains: an instruction does not wait for the one just issued)
    for (; r + 1 < r1; r += 2) {
        float a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int off = ((r + (k >> 2)) * XW + (k & 3) * 4) * 4; a[k] = xa[off]; b[k] = db[off]; }
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k + 1], b[k + 1], acc2, 0, 0, 0);
        }
    }
    acc[0] += acc2[0]; acc[1] += acc2[1]; acc[2] += acc2[2]; acc[3] += acc2[3];
    if (r < r1) {
#pragma unroll
        for (int c0 = 0; c0 < 16; c0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[(r * XW + c0) * 4], db[(r * XW + c0) * 4], acc, 0, 0, 0);
    }
}

// 16 -> 16, 3x3, rows of 16 pixels: unit = (tap, sample): D[ci][co] += A[ci][4 pixels] B[4 pixels][co] over the 64 pixel quads of the sample, one
// v_mfma_f32_16x16x4_f32 each (A: x at the tap's offset, B: dz; one ds_read_b32 per operand and lane).
// (eighteen units on sixteen waves: two SIMDs carried five units, two carried four.  Units 16 and 17 are cut into two row ranges each, one half unit per
// SIMD on waves 0 - 3: 4.5 units everywhere; the second halves' accumulators meet the first halves' in LDS — `wave_sums`, 2 x 256 floats, free here: see
// rb_wgrad_c4_mfma — even H only)
__device__ __forceinline__ void rb_wgrad_c16(const dcv_resident_bwd_op& o, const float4* xt, const float4* dzt, float* wave_sums, volatile unsigned* flag, unsigned flag_value) {
    const int tw = rs_tid();
    const int H = o.h, XW = 18, XH = H + 2, lane = tw & 63, wave = rs_wave_of(tw), m = lane & 15, kq = lane >> 4;
    const float* xf = reinterpret_cast<const float*>(xt); const float* df = reinterpret_cast<const float*>(dzt);
    auto unit = [&](int u, int r0, int r1, rs_f4& acc) {
        const int s = u >= 9 ? 1 : 0, tap = u - s * 9, ky = tap / 3, kx = tap - ky * 3;
        const float* xa = xf + ((size_t)s * RS_XIN + ((m >> 2) * XH + ky) * XW + kx + kq) * 4 + (m & 3);      // lane: input channel m, pixel kq of the quad
        const float* db = df + ((size_t)s * RS_XIN + ((m >> 2) * XH + 1) * XW + 1 + kq) * 4 + (m & 3);        // lane: output channel m, pixel kq of the quad
        acc = rs_f4{0.f, 0.f, 0.f, 0.f};
        rb_mfma_rows(xa, db, r0, r1, XW, acc);
    };
    auto store = [&](int u, const rs_f4& acc) {
        const int s = u >= 9 ? 1 : 0, tap = u - s * 9, chunk = (int)blockIdx.x * RS_SPB + s;
        *reinterpret_cast<float4*>(o.dw_part + (size_t)chunk * 2304 + (tap * 16 + m) * 16 + 4 * kq) = make_float4(acc[0], acc[1], acc[2], acc[3]);     // D: input channels 4 kq .. + 3, output channel m -> [tap][co][ci]: 1 KB per wave
    };
    const bool split = (H & 1) == 0;
    {
        rs_f4 acc;
        unit(wave, 0, H, acc);
        store(wave, acc);
    }
    if (split) {
        rs_f4 acc = {0.f, 0.f, 0.f, 0.f};
        const int u2 = 16 + (wave & 1), half = (wave >> 1) & 1;
        float* part = wave_sums + (wave & 1) * 256;
        if (wave < 4) {
            unit(u2, half * (H >> 1), (half + 1) * (H >> 1), acc);
            if (half == 1) {
                while (*flag != flag_value) { }
                *reinterpret_cast<rs_f4*>(part + lane * 4) = acc;
            }
        }
        rs_barrier();
        if (wave < 2) {
            const rs_f4 other = *reinterpret_cast<const rs_f4*>(part + lane * 4);
            acc[0] += other[0]; acc[1] += other[1]; acc[2] += other[2]; acc[3] += other[3];
            store(u2, acc);
        }
    } else if (wave < 2) {
        rs_f4 acc;
        unit(16 + wave, 0, H, acc);
        store(16 + wave, acc);
    }
    rb_bias16(o, dzt, 4);
}

// cin <= 4 -> 16, 3x3, rows of 16 pixels: unit = (filter row, sample): M = (filter column 0..3, input channel), the fourth column is not a tap (dropped)
// (six units of 64 dependent instructions kept six of the sixteen waves busy: 3.1 us.  Each unit is cut into two row ranges on two waves — twelve half
// units, 32 instructions each; the second halves' accumulators meet the first halves' in LDS: `part`, 192 floats per unit (rows kq < 3), in the per-wave
// moments' region and the quarter-unit scratch (free here: see rb_wgrad_c4_mfma) — even H only, else the six whole units as before)
__device__ __forceinline__ void rb_wgrad_c4x16(const dcv_resident_bwd_op& o, const float4* xt, const float4* dzt, float* scratch, float* wave_sums, volatile unsigned* flag, unsigned flag_value) {
    const int tw = rs_tid();
    const int H = o.h, XW = 18, XH = H + 2, lane = tw & 63, wave = rs_wave_of(tw), m = lane & 15, kq = lane >> 4;
    const float* xf = reinterpret_cast<const float*>(xt); const float* df = reinterpret_cast<const float*>(dzt);
    const bool split = (H & 1) == 0;
    auto unit = [&](int s, int ky, int r0, int r1, rs_f4& acc) {
        const float* xa = xf + ((size_t)s * RS_XIN + ky * XW + (m >> 2) + kq) * 4 + (m & 3);                    // lane: (column m / 4, input channel m % 4), pixel kq
        const float* db = df + ((size_t)s * RS_XIN + ((m >> 2) * XH + 1) * XW + 1 + kq) * 4 + (m & 3);
        acc = rs_f4{0.f, 0.f, 0.f, 0.f};
        rb_mfma_rows(xa, db, r0, r1, XW, acc);
    };
    auto store = [&](int s, int ky, const rs_f4& acc) {
        const int chunk = (int)blockIdx.x * RS_SPB + s, wsize = 16 * o.cin * 9;
        if (kq < 3) {                                                                // D rows 4 kq + i = (column kq, input channel i), output channel m
#pragma unroll
            for (int i = 0; i < 4; ++i) if (i < o.cin) o.dw_part[(size_t)chunk * wsize + ((ky * 3 + kq) * 16 + m) * o.cin + i] = acc[i];                  // [tap][co][ci]
        }
    };
    if (split) {
        rs_f4 acc = {0.f, 0.f, 0.f, 0.f};
        const int u = wave >> 1, half = wave & 1, s = u >= 3 ? 1 : 0, ky = u - s * 3;          // waves 0 - 11: unit u = wave / 2, row range wave % 2
        float* part = u < 5 ? wave_sums + u * 192 : scratch;
        if (wave < 12) {
            unit(s, ky, half * (H >> 1), (half + 1) * (H >> 1), acc);
            if (half == 1) {
                while (*flag != flag_value) { }
                if (kq < 3) *reinterpret_cast<rs_f4*>(part + lane * 4) = acc;
            }
        }
        rs_barrier();
        if (wave < 12 && half == 0) {
            if (kq < 3) { const rs_f4 other = *reinterpret_cast<const rs_f4*>(part + lane * 4); acc[0] += other[0]; acc[1] += other[1]; acc[2] += other[2]; acc[3] += other[3]; }
            store(s, ky, acc);
        }
        rb_bias16(o, dzt, 12);
        return;
    }
    for (int u = wave; u < RS_SPB * 3; u += 16) {
        const int s = u >= 3 ? 1 : 0, ky = u - s * 3;
        rs_f4 acc;
        unit(s, ky, 0, H, acc);
        store(s, ky, acc);
    }
    rb_bias16(o, dzt, 6);
}

// ---- cin <= 4 -> 4, 5 x 5 on 32-column maps (the stock model's first three layers) on v_mfma_f32_4x4x1: D_b[co][ci] += dz_b[co] x_b[ci], block b of an
// instruction = one of 16 consecutive pixels of an image row, so A is float `lane` of the dz tile's row segment and B float `lane` of the x tile's segment
// shifted by the tap ([pixel][channel] tiles: 64 consecutive floats, one conflict-free ds_read_b32 each).  A piece = (sample, filter column kx, row range):
// for each 16-column half it walks down the rows with the five x rows of the step in a register window — one new x read and one dz read per step feed FIVE
// instructions (1280 MACs), where the VALU version (rb_wgrad_c4) spent 15 VALU instructions per 10 packed FMAs per step and was issue-bound at ~9 us.  The
// 16 blocks of the accumulators are summed at the end (rb_sum_blocks24).  Work split, 200 instructions per wave: unit u = (kx = u / 2 < 4,
// sample u % 2) belongs to two waves; the first takes rows 0 - 19, the second rows 20 - 31 and then a quarter (rows 8 q .. 8 q + 7, q = u / 2, sample u % 2) of filter column
// 4.  The second waves' partials meet the first waves' in LDS (sc_main, the per-wave moments' region: the moments were consumed by the algebra of wave 0, which
// raises `flag` when it has read them), the quarters of column 4 in sc_q, in a fixed order.  The dz sums (bias gradient) ride along in the units of kx = 0
// as one VALU add per step.
// Sums of 24 values per lane over the 16 blocks of a 4x4x1 accumulator layout (lane = 4 b + j): two halving exchanges across the rows of 16 lanes
// (v_permlane32_swap / v_permlane16_swap: no LDS crossbar), then rotations by 8 and 4 lanes inside the rows.  On return out[n] is the total of value
// 4 n + 2 * (row & 1) + (row >> 1) (row = lane / 16) for the lane's j = lane % 4, in every lane.
// (inline asm: hipcc 7.2 adds the FIRST result of __builtin_amdgcn_permlane{32,16}_swap to itself when both are used — tools/probes/permlane_probe.hip shows
// the instructions themselves do what the ISA says; the s_nop around them stand in for the hazard recogniser, which does not look inside an asm statement)
template <bool AFTER_MFMA> __device__ __forceinline__ float rb_swap32_add(float a, float b) {
    if constexpr (AFTER_MFMA) asm volatile("s_nop 7\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));     // (operands fresh from the matrix pipe)
    else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float rb_swap16_add(float a, float b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); return a + b; }
__device__ __forceinline__ void rb_sum_blocks24(const float (&v)[24], float (&out)[6]) {
    float s1[12];
#pragma unroll
    for (int m = 0; m < 12; ++m) s1[m] = m == 0 ? rb_swap32_add<true>(v[2 * m], v[2 * m + 1]) : rb_swap32_add<false>(v[2 * m], v[2 * m + 1]);
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        float s = rb_swap16_add(s1[2 * n], s1[2 * n + 1]);
        s += rs_dpp_mov<0x128>(s);                                                  // row_ror:8
        s += rs_dpp_mov<0x124>(s);                                                  // row_ror:4
        out[n] = s;
    }
}
__host__ __device__ __forceinline__ bool rb_wgrad_c4_mfma_ok(int cin, int cout, int k, int h, int w) { return cin <= 4 && cout == 4 && k == 5 && w == 32 && h == 32; }
constexpr int RB_WGM_MAIN = 96, RB_WGM_Q = 80;                                       // floats per unit in sc_main ([6][4][4]) and sc_q ([5][4][4])
__device__ __forceinline__ void rb_wgrad_c4_mfma(const dcv_resident_bwd_op& o, const float4* xt, const float4* dzt, float* sc_q, float* sc_main, volatile unsigned* flag, unsigned flag_value) {
    constexpr int K = 5, pad = 2, XW = 32 + 2 * pad, RF = XW * 4;                    // RF: floats per tile row
    // (role by wave / 4: the four waves of a SIMD — wave % 4 — are two of each role, and the odd role's second reduction is spread over all SIMDs)
    const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t), u = (wave & 3) + 4 * (wave >> 3), role = (wave >> 2) & 1, kxu = u >> 1, su = u & 1;
    const int cin = o.cin;
    const float* xf = reinterpret_cast<const float*>(xt);
    const float* df = reinterpret_cast<const float*>(dzt);
    rs_f4 acc[K];
    float bsum;
    // a piece's rows are unrolled (H = 32: 20, 12 and 8 rows): every read carries its row as an immediate offset and the window of five x rows is a matter
    // of register names — in a rolled loop the four window moves, the address updates and the loop counter were 13 issue slots per five instructions, and
    // an issue slot of ANY wave of the SIMD costs the matrix pipe as much as half such an instruction (the first version of this function: 9.7 us)
    auto piece = [&](int s, int kx, int row0, auto rows_c, auto bias_c) {
        constexpr int ROWS = decltype(rows_c)::value;
        constexpr bool BIAS = decltype(bias_c)::value;
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = rs_f4{0.f, 0.f, 0.f, 0.f};
        bsum = 0.f;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const float* xb = xf + ((size_t)s * RS_XIN + row0 * XW + half * 16 + kx) * 4 + lane;
            const float* db = df + ((size_t)s * RS_XIN + (row0 + pad) * XW + pad + half * 16) * 4 + lane;
            // every operand of the half is requested before the first instruction (ROWS + 4 + ROWS registers): the LDS latency is paid once per half, the
            // waits are counted
            float w[ROWS + K - 1], d[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS + K - 1; ++r) { w[r] = xb[r * RF]; if (r < ROWS) d[r] = db[r * RF]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
#pragma unroll
                for (int ky = 0; ky < K; ++ky) acc[ky] = __builtin_amdgcn_mfma_f32_4x4x1f32(d[r], w[r + ky], acc[ky], 0, 0, 0);
                if constexpr (BIAS) bsum += d[r];
            }
        }
    };
    using rows_a = std::integral_constant<int, 20>; using rows_b = std::integral_constant<int, 12>; using rows_q = std::integral_constant<int, 8>;
    float tot[6];
    auto reduce = [&]() {
        float v[24];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[k * 4 + i] = acc[k][i];
        v[20] = bsum; v[21] = v[22] = v[23] = 0.f;                                  // (value 20: lane 4 b + j holds the dz sum of channel j over its pixels)
        rb_sum_blocks24(v, tot);
    };
    const int row = lane >> 4, j = lane & 3, co = 2 * (row & 1) + (row >> 1);       // tot[n]: filter row n, output channel co, input channel j (n = 5, row 0: dz sum of channel j)
    const bool writer = (lane & 12) == 0;
    const int chunk0 = (int)blockIdx.x * RS_SPB, wsize = 4 * cin * K * K;
    if (role == 1) {
        if (kxu == 0) piece(su, kxu, 20, rows_b{}, std::true_type{}); else piece(su, kxu, 20, rows_b{}, std::false_type{});
        reduce();
        while (*flag != flag_value) { }
        if (writer) {
#pragma unroll
            for (int n = 0; n < 6; ++n) sc_main[u * RB_WGM_MAIN + n * 16 + co * 4 + j] = tot[n];
        }
        piece(su, K - 1, (u >> 1) * 8, rows_q{}, std::false_type{});
        reduce();
        if (writer) {
#pragma unroll
            for (int n = 0; n < K; ++n) sc_q[u * RB_WGM_Q + n * 16 + co * 4 + j] = tot[n];
        }
    } else {
        if (kxu == 0) piece(su, kxu, 0, rows_a{}, std::true_type{}); else piece(su, kxu, 0, rows_a{}, std::false_type{});
        reduce();
    }
    rs_barrier();
    if (role == 0 && writer) {
        float* dst = o.dw_part + (size_t)(chunk0 + su) * wsize + (kxu * 4 + co) * cin + j;      // [tap][co][ci] slab, tap = ky * K + kx
        const float* part = sc_main + u * RB_WGM_MAIN + co * 4 + j;
#pragma unroll
        for (int n = 0; n < K; ++n) if (j < cin) dst[n * K * 4 * cin] = tot[n] + part[n * 16];
        if (kxu == 0 && row == 0 && o.db_part) o.db_part[(size_t)(chunk0 + su) * 4 + j] = tot[5] + sc_main[u * RB_WGM_MAIN + 5 * 16 + j];
    }
    if (t < RS_SPB * RB_WGM_Q) {                                                     // filter column 4: (sample, ky, co, ci), the four quarters in a fixed order
        const int ss = t >= RB_WGM_Q ? 1 : 0, e = t - ss * RB_WGM_Q, n = e >> 4, c2 = (e >> 2) & 3, ci = e & 3;
        const float* q = sc_q + ss * RB_WGM_Q + e;
        const float v = (q[0] + q[2 * RB_WGM_Q]) + (q[4 * RB_WGM_Q] + q[6 * RB_WGM_Q]);
        if (ci < cin) o.dw_part[(size_t)(chunk0 + ss) * wsize + ((n * K + (K - 1)) * 4 + c2) * cin + ci] = v;
    }
}
__device__ __forceinline__ void rb_wgrad(const dcv_resident_bwd_op& o, const float4* xt, const float4* dzt, float* scratch, float* wave_sums, volatile unsigned* flag, unsigned flag_value) {
    const int var = rb_wgrad_variant(o.cin, o.cout, o.k, o.w);
    if (var == 1) {
        if (rb_wgrad_c4_mfma_ok(o.cin, o.cout, o.k, o.h, o.w)) rb_wgrad_c4_mfma(o, xt, dzt, scratch, wave_sums, flag, flag_value);
        else if (o.k == 5) {
            if (o.w != 32) rb_wgrad_c4<5, 4, 0>(o, xt, dzt, scratch);
            else if (o.cin == 3) rb_wgrad_c4<5, 3, 36>(o, xt, dzt, scratch);
            else rb_wgrad_c4<5, 4, 36>(o, xt, dzt, scratch);
        } else rb_wgrad_c4<3, 4, 0>(o, xt, dzt, scratch);
    }
    else if (var == 2) rb_wgrad_c4x16(o, xt, dzt, scratch, wave_sums, flag, flag_value);
    else rb_wgrad_c16(o, xt, dzt, wave_sums, flag, flag_value);
}

__device__ __forceinline__ float rb_act_grad(float y, int act, float slope) { return dcv_act_grad_from_out(y, act, slope); }

// the head's resize gradient keeps, per input row / column of the referenced tensor, the output rows / columns whose two bilinear taps touch it
constexpr int RB_TAPS = 8;
inline bool rb_resize_taps_ok(int in_size, int out_size, int align_corners) {
    if (in_size <= 0 || out_size <= 0 || in_size > 4096 || out_size > 4096) return false;
    const float scale = align_corners ? (out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f) : (float)in_size / (float)out_size;
    int count[4096] = {0};
    for (int dst = 0; dst < out_size; ++dst) {
        float fs = align_corners ? scale * dst : scale * (dst + 0.5f) - 0.5f; if (fs < 0.f) fs = 0.f;
        int i0 = (int)fs; if (i0 > in_size - 1) i0 = in_size - 1;
        const int i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
        if (++count[i0] > RB_TAPS) return false;
        if (i1 != i0 && ++count[i1] > RB_TAPS) return false;
    }
    return true;
}

// ---- weight / bias gradient of the fully connected head inside the backward launch.  Its operands exist before the launch (the head's saved input x
// [n][K] in NHWC order, dlogits, the logits), so no grid synchronisation is involved: the first K / 8 blocks each take a GROUP of 8 consecutive input
// positions (one 32-B sector of every sample's row) and compute their gradient rows for ALL samples, final values straight into dw (the reference's
// [O][C * H * W] order) — no partial slabs.  The operands are requested at the very start of the launch and parked in the dz tile's LDS region (free
// until the first layer's dz): dz of all samples [N][O] (coalesced loads) and one float4 of x per thread = (sample, half of the group).  The arithmetic
// runs while the partials of the first all-reduce travel (that wait has no earlier layer's weight gradient to hide behind): products summed over the 64
// lanes (rs_reduce8), then over the eight waves of the half through LDS in a fixed order.  Groups are dealt so that the blocks of one XCD (block % 8)
// read one contiguous eighth of every row: each L2 fetches its part of x once.
constexpr int RB_FC_MAXO = 12, RB_FC_DZ_SLOTS = (512 * RB_FC_MAXO) / RS_T;          // classes; dz values staged per thread
static_assert(512 * RB_FC_MAXO * 4 + RS_T * 16 <= RS_SPB * RS_XIN * 16, "the staged operands of the head's weight gradient must fit the dz tile region");
__host__ __device__ __forceinline__ bool rb_fc_wgrad_ok(int n, int k, int classes) {
    const int g = (n + RS_SPB - 1) / RS_SPB;
    return n > 0 && n <= 512 && k > 0 && (k & 7) == 0 && classes > 0 && classes <= RB_FC_MAXO && (k >> 3) <= g;
}
// the block's group of input positions, or -1
__device__ __forceinline__ int rb_fc_group(int K) {
    const int G = (int)gridDim.x, GA = K >> 3, per = (GA + 7) >> 3;
    int pg = (int)blockIdx.x;
    if ((G & 7) == 0 && per * 8 <= G) { const int xcd = pg & 7, i = pg >> 3; pg = i < per ? xcd * per + i : GA; }
    return pg < GA ? pg : -1;
}
struct rb_fc_args { float* fc_dw; float* fc_db; int classes, cw, hwo, n, pg; };
// (not inlined, arguments by value: inlined, its registers pushed the allocation of the kernel's hot loops into spills — 0.2551 -> 0.2698 ms per step)
__device__ __attribute__((noinline)) void rb_fc_wgrad(const rb_fc_args hd, const float* staged, float* scratch) {
    const int O = hd.classes, Cw = hd.cw, HWo = hd.hwo, K = HWo * Cw, N = hd.n, pg = hd.pg;
    const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t), n = t & 511, half = __builtin_amdgcn_readfirstlane(t >> 9);
    const float4 x4 = reinterpret_cast<const float4*>(staged + 512 * RB_FC_MAXO)[t];
    float dz[RB_FC_MAXO];
#pragma unroll
    for (int j = 0; j < RB_FC_MAXO; ++j) dz[j] = (n < N && j < O) ? staged[n * O + j] : 0.f;
    const int vi = (lane & 1) * 4 + ((lane >> 1) & 1) * 2;
    float r2[2];
#pragma unroll
    for (int c = 0; c < RB_FC_MAXO / 2; ++c) {                                       // call c: classes 2c, 2c + 1 x the half's four positions
        if (2 * c < O) {
            const float v[8] = {x4.x * dz[2 * c], x4.y * dz[2 * c], x4.z * dz[2 * c], x4.w * dz[2 * c],
                                x4.x * dz[2 * c + 1], x4.y * dz[2 * c + 1], x4.z * dz[2 * c + 1], x4.w * dz[2 * c + 1]};
            rs_reduce8(v, r2);
            if (lane < 4) { scratch[(wave * 8 + c) * 8 + vi] = r2[0]; scratch[(wave * 8 + c) * 8 + vi + 1] = r2[1]; }
        }
    }
    if (pg == 0 && half == 0 && hd.fc_db != nullptr) {                               // the bias gradient: sums of dz over the samples (waves 0 - 7 of group 0)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 8 * c + i < RB_FC_MAXO ? dz[8 * c + i < RB_FC_MAXO ? 8 * c + i : 0] : 0.f;
            rs_reduce8(v, r2);
            if (lane < 4) { scratch[(16 * 8 + wave * 2 + c) * 8 + vi] = r2[0]; scratch[(16 * 8 + wave * 2 + c) * 8 + vi + 1] = r2[1]; }
        }
    }
    rs_barrier();
    if (t < 128) {
        const int pos = t >> 4, j = t & 15, hf = pos >> 2, q = pos & 3;
        if (j < O) {
            const float* sc = scratch + ((hf * 8) * 8 + (j >> 1)) * 8 + (j & 1) * 4 + q;
            const float v = ((sc[0] + sc[64]) + (sc[128] + sc[192])) + ((sc[256] + sc[320]) + (sc[384] + sc[448]));      // the half's eight waves, 8 * 8 floats apart
            const int m = pg * 8 + pos, px = m / Cw, ch = m - px * Cw;
            hd.fc_dw[(size_t)j * K + (size_t)ch * HWo + px] = v;
        }
    } else if (t < 144 && pg == 0 && hd.fc_db != nullptr) {
        const int j = t - 128;
        if (j < O) {
            const float* sc = scratch + (16 * 8 + (j >> 3)) * 8 + (j & 7);
            hd.fc_db[j] = ((sc[0] + sc[16]) + (sc[32] + sc[48])) + ((sc[64] + sc[80]) + (sc[96] + sc[112]));          // waves 0 - 7, 2 * 8 floats apart
        }
    }
}

__global__ __launch_bounds__(RS_T) void resident_bwd_kernel(dcv_resident_bwd_program p, unsigned* ctl, rs_u64* gran) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* dzt = reinterpret_cast<float4*>(smem);
    float* yreg = reinterpret_cast<float*>(smem + RS_XIN_BYTES);
    float* dor = reinterpret_cast<float*>(smem + RS_XIN_BYTES + RS_Y_BYTES);
    float4* xtt = reinterpret_cast<float4*>(smem + RS_XIN_BYTES + 2 * RS_Y_BYTES);
    __shared__ float s_wsum[16 * 4 * RS_MAXC];                                       // per-wave moments [wave][4][C]
    __shared__ double s_tot[RS_MAXQ], s_red[4];                                      // batch totals of the current synchronisation
    __shared__ unsigned s_w32[2 * RS_MAXQ];
    __shared__ __attribute__((aligned(16))) float s_coef[RS_SPB][3][RS_MAXC];        // dz = act'(y) (P d + Q u + R)
    __shared__ __attribute__((aligned(16))) float s_mr[2][RS_MAXC];                  // BatchNorm mean, rstd of the current layer
    __shared__ float s_abn[RS_SPB][RS_MAXC];                                         // BatchNorm gamma * rstd
    __shared__ __attribute__((aligned(16))) float s_par[DCV_RESIDENT_BWD_MAX_OPS][5][RS_MAXC];   // bn gamma, beta, mean, rstd, gn gamma of every layer
    __shared__ float s_scratch[4 * RS_SPB * 4 * 5 * 4];                              // data-gradient phases: the convolutions' unused sums; weight-gradient phases: quarter-unit partials (rb_wgrad_c4)
    float* const s_dummy = s_scratch;
    float* const s_wgq = s_scratch;
    __shared__ __attribute__((aligned(16))) float s_filt[RB_FILT_SLOTS][RS_FILT_FLOATS];  // data-gradient filters of the 4 <- 4 layers as [tap][co][ci] (rs_conv_c4_co4)                                  // quarter-unit partials of the 5x5 weight gradient (rb_wgrad_c4)
    __shared__ unsigned s_gen, s_algebra;
    __shared__ int s_fail;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int s_of_t = t >> 9, tt = t & 511;
    const int N = p.n, G = (int)gridDim.x;
    const bool valid0 = (int)blockIdx.x * RS_SPB < N, valid1 = (int)blockIdx.x * RS_SPB + 1 < N;
    if (rs_wave_of(rs_tid()) == 0) rs_kernarg_warm<(int)(sizeof(dcv_resident_bwd_program) + 2 * sizeof(void*))>();     // (wave 0 alone: see the forward launch)
    rs_barrier();
    int lz = 0; asm volatile("" : "+v"(lz));
    const unsigned ctl_mute = ctl[2 + lz], ctl_stamp = ctl[3 + lz];                  // (as in the forward launch: first needed at the end of the prologue)
    const unsigned gen0 = __hip_atomic_load((rs_gu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const rs_u64 t_start = wall_clock64();
    int n_stamp = 1;
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(ctl + 4);
    auto stamp = [&]() {
        if (blockIdx.x == ctl_stamp && t == 0 && n_stamp < RS_MAX_STAMPS) { if (n_stamp == 1) stamps[1] = t_start; stamps[1 + n_stamp] = wall_clock64(); }
        ++n_stamp;
    };
    // a sample's [HW * C] floats, HBM -> LDS region (flat), 8 float4 per thread in flight
    auto load_flat = [&](const float* src, float* region, int floats_per_sample) {
        const int n = (int)blockIdx.x * RS_SPB + s_of_t, v4 = floats_per_sample >> 2;
        const float4* g = reinterpret_cast<const float4*>(src) + (size_t)(n < N ? n : 0) * v4;
        float4* d = reinterpret_cast<float4*>(region + s_of_t * RS_Y);
        for (int i0 = 0; i0 < v4; i0 += 8 * 512) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = i0 + tt + k * 512; v[k] = g[i < v4 ? i : 0]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = i0 + tt + k * 512; if (i < v4) d[i] = n < N ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f); }
        }
    };
    // the head's weight gradient: its operands are requested first and parked in LDS at the end of this prologue (rb_fc_wgrad)
    const int fc_pg = (p.head.enabled && p.head.fc_dw != nullptr) ? rb_fc_group(p.head.h * p.head.w * (p.head.cc + p.head.cr)) : -1;
    float4 fc_x4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float fc_g[RB_FC_DZ_SLOTS], fc_l[RB_FC_DZ_SLOTS];
    const bool ce_here = p.head.enabled && p.head.ce_target != nullptr;            // the step's cross-entropy is computed by this launch (row t by thread t)
    const bool stage_dz = p.head.enabled && (p.head.fc_dw != nullptr || ce_here);  // every block: its own samples' dl (the head's data gradient) comes from there too
    long long ce_t = 0;
    if (stage_dz) {
        const int K = p.head.h * p.head.w * (p.head.cc + p.head.cr), total = N * p.head.classes;
        if (fc_pg >= 0 && tt < N) fc_x4 = *reinterpret_cast<const float4*>(p.head.fc_x + (size_t)tt * K + fc_pg * 8 + s_of_t * 4);
        if (ce_here) {
            // the sample's logits: slots 0 .. 11 of (fc_g, fc_l)
            const int O = p.head.classes, row = t < N ? t : 0;
#pragma unroll
            for (int k = 0; k < RB_FC_DZ_SLOTS; ++k) { fc_g[k] = p.head.logits[(size_t)row * O + (k < O ? k : 0)]; fc_l[k] = p.head.logits[(size_t)row * O + (RB_FC_DZ_SLOTS + k < O ? RB_FC_DZ_SLOTS + k : 0)]; }
            ce_t = p.head.ce_target[row];
        } else {
#pragma unroll
            for (int k = 0; k < RB_FC_DZ_SLOTS; ++k) { const int i = t + k * RS_T; fc_g[k] = p.head.dlogits[i < total ? i : 0]; fc_l[k] = p.head.logits ? p.head.logits[i < total ? i : 0] : 0.f; }
        }
    }
    int pub_seq_pre = 0;
    {   // per-channel parameters of every layer -> LDS; y and dout of the first op (the last layer) -> LDS
        const dcv_resident_bwd_op& o = p.op[0];
        const int pk = o.pool > 0 ? o.pool : 1;
        // y of the first op: requested now, committed to LDS behind the head's operand staging — the loss arithmetic of a captured step (2 us on half of the
        // waves) then runs while these loads travel
        const int yv4 = (o.h * o.w * o.cout) >> 2, yn = (int)blockIdx.x * RS_SPB + s_of_t;
        const bool y_two = yv4 <= 2 * 512;
        float4 y2[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        if (y_two) {
            const float4* g = reinterpret_cast<const float4*>(o.y) + (size_t)(yn < N ? yn : 0) * yv4;
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int i = tt + k * 512; y2[k] = g[i < yv4 ? i : 0]; }
        }
        // a pair of waves per op (128 slots: 5 rows of 16 channels used): the descriptor through scalar loads (rs_kernarg_warm), the row's pointer by selects —
        // indexed per lane, descriptor, pointer and value were three dependent round trips which every load below waited behind
        // (unconditional load, untouched until the commit below: see the forward launch)
        float pl, pd; int pslot; bool pok;
        {
            const int oi = __builtin_amdgcn_readfirstlane(t >> 7), oc = oi < p.count ? oi : 0, f = (t >> 4) & 7, c = t & 15;
            const dcv_resident_bwd_op& w = p.op[oc];
            const float* q0 = w.bn_gamma; const float* q1 = w.bn_beta; const float* q2 = w.bn_mean; const float* q3 = w.bn_rstd; const float* q4 = w.gn_gamma;
            asm volatile("" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3), "+s"(q4));   // (opaque: see the forward launch)
            const float* src = f == 0 ? q0 : f == 1 ? q1 : f == 2 ? q2 : f == 3 ? q3 : q4;
            const bool slot = oi < p.count && f < 5 && c < w.cout;
            pslot = slot ? (oi * 5 + f) * RS_MAXC + c : -1;
            pd = (f == 0 || f == 4) ? 1.f : 0.f;
            pok = slot && src != nullptr;
            pl = *(const rs_gf32*)(pok ? src + c : reinterpret_cast<const float*>(ctl));
        }
        // the filters of the 4 <- 4 data gradients: the n-th such op owns slot n of s_filt (rb_filt_slot)
        float fv[RB_FILT_SLOTS];
#pragma unroll
        for (int sl = 0; sl < RB_FILT_SLOTS; ++sl) fv[sl] = 0.f;
        {
            // which op owns which slot: scalar arithmetic first, then ONE load per slot (a load per (op, slot) pair under `ns == sl` had the compiler wait
            // for everything in flight between the ops: the pairs share their destination registers)
            int slot_op[RB_FILT_SLOTS];
#pragma unroll
            for (int sl = 0; sl < RB_FILT_SLOTS; ++sl) slot_op[sl] = -1;
            int ns = 0;
#pragma unroll
            for (int k = 0; k < DCV_RESIDENT_BWD_MAX_OPS; ++k) {
                if (k < p.count && p.op[k].need_dx && rb_dgrad_variant(p.op[k].cin, p.op[k].cout, p.op[k].k, p.op[k].w) == 1) {
#pragma unroll
                    for (int sl = 0; sl < RB_FILT_SLOTS; ++sl) if (ns == sl) slot_op[sl] = k;
                    ++ns;
                }
            }
#pragma unroll
            for (int sl = 0; sl < RB_FILT_SLOTS; ++sl)
                if (slot_op[sl] >= 0) { const dcv_resident_bwd_op& w = p.op[slot_op[sl]]; fv[sl] = rs_filt_fetch(w.filt, w.k, t); }
        }
        // the mailbox's sequence number for the loss publication (the writer block only; every other thread reads a control word: unconditional, see above)
        {
            const bool pub_here = p.head.enabled && p.head.pub_seq != nullptr && blockIdx.x == (G > RS_MAXQ + DCV_MAX_COPY_JOBS ? (unsigned)(RS_MAXQ + DCV_MAX_COPY_JOBS) : 0u);
            const int* sp = pub_here ? &p.head.pub_mailbox->seq : reinterpret_cast<const int*>(ctl);
            pub_seq_pre = *(const __attribute__((address_space(1))) int*)sp;
        }
        if (stage_dz) {
            float* staged = reinterpret_cast<float*>(dzt);
            const int total = N * p.head.classes;
            reinterpret_cast<float4*>(staged + 512 * RB_FC_MAXO)[t] = fc_x4;
            if (ce_here) {
                // softmax cross-entropy of row t, the arithmetic of ce_fwd_single_kernel (linear_loss.hip) expression by expression: the loss and its gradient
                // must not depend on which launch computes them
                const int O = p.head.classes;
                double l = 0.0; int ok = 0;
                float lse = 0.f;
                if (t < N) {
                    float v[2 * RB_FC_DZ_SLOTS];
    #pragma unroll
                    for (int k = 0; k < RB_FC_DZ_SLOTS; ++k) { v[k] = k < O ? fc_g[k] : -INFINITY; v[RB_FC_DZ_SLOTS + k] = RB_FC_DZ_SLOTS + k < O ? fc_l[k] : -INFINITY; }
                    float m = -INFINITY; int mi = 0x7fffffff;
    #pragma unroll
                    for (int c = 0; c < 2 * RB_FC_DZ_SLOTS; ++c) if (c < O && (v[c] > m || (v[c] != v[c]))) { m = v[c]; mi = c; }
                    float sexp = 0.f, zt = 0.f;
    #pragma unroll
                    for (int c = 0; c < 2 * RB_FC_DZ_SLOTS; ++c) if (c < O) { sexp += expf(v[c] - m); if ((long long)c == ce_t) zt = v[c]; }
                    lse = m + logf(sexp);
                    const float g = 1.f / (float)N;
    #pragma unroll
                    for (int c = 0; c < 2 * RB_FC_DZ_SLOTS; ++c)
                        if (c < O) staged[t * O + c] = ((expf(v[c] - lse) - ((long long)c == ce_t ? 1.f : 0.f)) * g) * dcv_act_grad_from_out(v[c], p.head.act, p.head.slope);
                    l = (double)(lse - zt);
                    ok = (mi == (int)ce_t) ? 1 : 0;
                }
                l = dcv_wave_sum(l); ok = dcv_wave_sum(ok);
                if (lane == 0) { s_tot[wave] = l; s_w32[wave] = (unsigned)ok; }
                rs_barrier();
                const bool writer = blockIdx.x == (G > RS_MAXQ + DCV_MAX_COPY_JOBS ? (unsigned)(RS_MAXQ + DCV_MAX_COPY_JOBS) : 0u);     // a block which owns no all-reduce quantity
                if (writer) {
                    if (t < N && p.head.ce_lse != nullptr) p.head.ce_lse[t] = lse;
                    if (t == 0) {
                        double sum = 0.0; int k = 0;
                        for (int i = 0; i < 16; ++i) { sum += s_tot[i]; k += (int)s_w32[i]; }
                        const float value = (float)(sum / (double)N);
                        p.head.ce_loss[0] = value;
                        if (p.head.ce_correct != nullptr) p.head.ce_correct[0] = k;
                        // early read-back, as dcv_cross_entropy_fwd_grad_publish (the sequence number was requested with the prologue's loads: this block
                        // is the last to reach the first synchronisation, and a memory round trip plus two fences here were 1.8 us of every block's wait)
                        if (p.head.pub_seq != nullptr) dcv_publish_value(p.head.pub_vals, p.head.pub_seq, p.head.pub_slots, value, pub_seq_pre);
                    }
                }
            } else {
    #pragma unroll
                for (int k = 0; k < RB_FC_DZ_SLOTS; ++k) {
                    const int i = t + k * RS_T;
                    if (i < total) staged[i] = p.head.logits ? fc_g[k] * dcv_act_grad_from_out(fc_l[k], p.head.act, p.head.slope) : fc_g[k];
                }
            }
        }
        if (y_two) {
            float4* d = reinterpret_cast<float4*>(yreg + s_of_t * RS_Y);
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int i = tt + k * 512; if (i < yv4) d[i] = yn < N ? y2[k] : make_float4(0.f, 0.f, 0.f, 0.f); }
        } else load_flat(o.y, yreg, o.h * o.w * o.cout);
        if (!p.head.enabled) load_flat(o.dout, dor, (o.h / pk) * (o.w / pk) * o.cout);
        if (pslot >= 0) (&s_par[0][0][0])[pslot] = pok ? pl : pd;
        if (t < RS_FILT_FLOATS) {
#pragma unroll
            for (int sl = 0; sl < RB_FILT_SLOTS; ++sl) s_filt[sl][t] = fv[sl];
        }
    }
    if (t == 0) { s_gen = gen0; s_fail = 0; s_algebra = 0; }
    if (stage_dz) rs_barrier();
    stamp();
    if (p.head.enabled) {
        // ---- head of the backward pass: dl = dlogits act'(logits); dwide = dl W; the current tensor's channels -> op 0's dout (LDS), the referenced tensor's
        // channels -> back through the bilinear resize (gather: each input pixel sums the output pixels whose footprint contains it) -> dref (HBM)
        const dcv_resident_bwd_head& hd = p.head;
        const int O = hd.classes, HWo = hd.h * hd.w, Cc = hd.cc, Cr = hd.cr, Cw = Cc + Cr, K = HWo * Cw;
        float* tmp = reinterpret_cast<float*>(xtt);                                  // [RS_SPB][HWo * Cr]: the referenced channels of dwide (the x tile is still free)
        if (stage_dz) {                                                              // dl of this block's samples: already in LDS with everybody's (the head's weight gradient)
            if (t < RS_SPB * O) { const int ss = t >= O ? 1 : 0, j = t - ss * O, n = (int)blockIdx.x * RS_SPB + ss; s_wsum[ss * 16 + j] = n < N ? reinterpret_cast<const float*>(dzt)[n * O + j] : 0.f; }
        } else if (t < RS_SPB * O) {
            const int ss = t >= O ? 1 : 0, j = t - ss * O, n = (int)blockIdx.x * RS_SPB + ss;
            float g = 0.f;
            if (n < N) { g = hd.dlogits[(size_t)n * O + j]; if (hd.logits) g *= dcv_act_grad_from_out(hd.logits[(size_t)n * O + j], hd.act, hd.slope); }
            s_wsum[ss * 16 + j] = g;
        }
        rs_barrier();
        {
            const float inv_cw = 1.f / (float)Cw;
            // a thread computes its positions for BOTH samples (every weight is loaded once per block), two positions per pass with all their weights
            // requested before the first use (a pass per position = a round trip to L2 each)
            for (int k0 = t; k0 < K; k0 += 2 * RS_T) {
                float wv[2][16];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 16; ++j) { const int k = k0 + q * RS_T; wv[q][j] = (j < O && k < K) ? hd.wp[(size_t)j * K + k] : 0.f; }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k = k0 + q * RS_T;
                    if (k < K) {
                        float a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int j = 0; j < 16; ++j) if (j < O) { a0 += s_wsum[j] * wv[q][j]; a1 += s_wsum[16 + j] * wv[q][j]; }
                        const int px = rs_fdiv(k, inv_cw), c = k - px * Cw;
                        if (c < Cc) { dor[px * Cc + c] = a0; dor[RS_Y + px * Cc + c] = a1; } else { tmp[px * Cr + (c - Cc)] = a0; tmp[RS_Y + px * Cr + (c - Cc)] = a1; }
                    }
                }
            }
        }
        {
            // the resize's weights are separable: per input row (column) the <= RB_TAPS output rows (columns) whose two taps touch it and their weights, once
            // per block into LDS (behind `tmp` in the x tile's region) — evaluating the tap arithmetic per element and candidate took 5 us of this prologue
            const int RH = hd.ref_h, RW = hd.ref_w;
            const float sh = hd.align_corners ? (hd.h > 1 ? (float)(RH - 1) / (float)(hd.h - 1) : 0.f) : (float)RH / (float)hd.h;
            const float sw = hd.align_corners ? (hd.w > 1 ? (float)(RW - 1) / (float)(hd.w - 1) : 0.f) : (float)RW / (float)hd.w;
            float* wtab = tmp + RS_SPB * RS_Y;                                       // [RH + RW][RB_TAPS] weights, then the same shape of output indices
            int* itab = reinterpret_cast<int*>(wtab + (RH + RW) * RB_TAPS);
            if (t < RH + RW) {
                const bool row = t < RH;
                const int i = row ? t : t - RH, in_size = row ? RH : RW, out_size = row ? hd.h : hd.w;
                const float scale = row ? sh : sw;
                int n = 0;
                for (int dst = 0; dst < out_size; ++dst) {
                    float fs = hd.align_corners ? scale * dst : scale * (dst + 0.5f) - 0.5f; if (fs < 0.f) fs = 0.f;
                    int i0 = (int)fs; if (i0 > in_size - 1) i0 = in_size - 1;
                    const int i1 = i0 + ((i0 < in_size - 1) ? 1 : 0); const float l1 = fs - (float)i0;
                    float wgt = 0.f;
                    if (i0 == i) wgt += 1.f - l1;
                    if (i1 == i) wgt += l1;
                    if (wgt != 0.f && n < RB_TAPS) { wtab[t * RB_TAPS + n] = wgt; itab[t * RB_TAPS + n] = dst; ++n; }
                }
                for (; n < RB_TAPS; ++n) { wtab[t * RB_TAPS + n] = 0.f; itab[t * RB_TAPS + n] = 0; }
            }
        }
        rs_barrier();
        stamp();
        {
            const int RH = hd.ref_h, RW = hd.ref_w, per = RH * RW * Cr;
            const float inv_cr = 1.f / (float)Cr, inv_rw = 1.f / (float)RW;
            const float* wtab = tmp + RS_SPB * RS_Y;
            const int* itab = reinterpret_cast<const int*>(wtab + (RH + RW) * RB_TAPS);
            for (int e = tt; e < per; e += 512) {
                const int px = rs_fdiv(e, inv_cr), cr = e - px * Cr, rh = rs_fdiv(px, inv_rw), rw = px - rh * RW;
                float acc = 0.f;
#pragma unroll
                for (int a = 0; a < RB_TAPS; ++a) {
                    const float wh = wtab[rh * RB_TAPS + a];
                    if (wh == 0.f) break;                                           // (the table packs its non-zero taps first: the first zero ends the row)
                    const int ho = itab[rh * RB_TAPS + a];
#pragma unroll
                    for (int c = 0; c < RB_TAPS; ++c) {
                        const float ww = wtab[(RH + rw) * RB_TAPS + c];
                        if (ww == 0.f) break;
                        acc += wh * ww * tmp[s_of_t * RS_Y + (ho * hd.w + itab[(RH + rw) * RB_TAPS + c]) * Cr + cr];
                    }
                }
                const int n = (int)blockIdx.x * RS_SPB + s_of_t;
                if (n < N) hd.dref[(size_t)n * per + e] = acc;
            }
        }
    }
    rs_barrier();
    const unsigned gen = s_gen;
    stamp();
    // One more pass than ops: the last one (`tail`) only computes the first layer's weight gradient, which has no later all-reduce to hide behind — ONE
    // call site of rb_wgrad in the loop instead of a second copy of all its variants behind it.
    bool moments_done = false;                                                       // the previous op's data gradient left this op's moments in s_wsum (rs_conv_c4_co4_32<.., true>)
    for (int oi = 0; oi <= p.count; ++oi) {
        const bool tail = oi == p.count;
        bool moments_next = false;
        const dcv_resident_bwd_op& o = p.op[tail ? oi - 1 : oi];
        const int t = rs_tid(), lane = t & 63, wave = rs_wave_of(t), s_of_t = __builtin_amdgcn_readfirstlane(t >> 9), tt = t & 511;     // (per op: see rs_tid)
        // (every wave runs this scalar set-up once per op: the usual extents skip the ~25-instruction integer divisions and the IEEE divisions — same values)
        const int C = o.cout, CV = C >> 2, H = o.h, W = o.w, HW = H * W, pk = o.pool > 0 ? o.pool : 1, Wo = pk == 1 ? W : pk == 2 ? W >> 1 : W / pk;
        const float pinv = pk == 1 ? 1.f : pk == 2 ? 0.25f : 1.f / (float)(pk * pk), inv_w = W == 32 ? 0.03125f : W == 16 ? 0.0625f : 1.f / (float)W, inv_pk = pk == 1 ? 1.f : pk == 2 ? 0.5f : 1.f / (float)pk;
        // the (sample, channel) threads of wave 0 request their GroupNorm statistics now; used after the moments
        const int ac_ss = t >= C ? 1 : 0, ac_c = t - ac_ss * C, ac_n = (int)blockIdx.x * RS_SPB + ac_ss;
        const int cpg = o.groups <= 0 ? 1 : o.groups == 1 ? C : (o.groups & (o.groups - 1)) == 0 ? C >> (31 - __builtin_clz(o.groups)) : C / o.groups;
        const unsigned epoch = gen * DCV_RESIDENT_MAX_OPS + (unsigned)oi + 1u;
        rs_u64* go = gran + (size_t)(tail ? 0 : oi) * RS_GRAN_OP;
        if (!tail) {
        float e_gr = 0.f, e_gm = 0.f;
        float e_sy = 0.f, e_syy = 0.f;                                               // the forward's sums of y, y^2 of (sample, channel), when it left them
        const bool have_sums = o.sample_sums != nullptr;
        if (t < RS_SPB * C && o.groups > 0 && ac_n < N) {
            const int g = (cpg & (cpg - 1)) == 0 ? ac_c >> (31 - __builtin_clz(cpg)) : rs_fdiv(ac_c, 1.f / (float)cpg);
            e_gr = o.gn_rstd[(size_t)ac_n * o.groups + g]; e_gm = o.gn_mean[(size_t)ac_n * o.groups + g];
            if (have_sums) { e_sy = o.sample_sums[((size_t)ac_n * 2) * C + ac_c]; e_syy = o.sample_sums[((size_t)ac_n * 2 + 1) * C + ac_c]; }
        }
        if (t < C) { s_mr[0][t] = s_par[oi][2][t]; s_mr[1][t] = s_par[oi][3][t]; }
        rs_barrier();
        // ---- four moments per (sample, channel): sum d, sum d u, sum u, sum u^2 with u = (y - mean) rstd, d = dout through the pooling map.  Wave w8 of the
        // sample owns channel vector w8 % CV and every (8 / CV)-th stripe of 64 pixels.
        if (!moments_done) {
            const int w8 = wave & 7, cv = w8 & (CV - 1), stripes = 8 / CV, st0 = w8 / CV;
            const float4 mu = *reinterpret_cast<const float4*>(s_mr[0] + cv * 4), rs = *reinterpret_cast<const float4*>(s_mr[1] + cv * 4);
            const float4* ys = reinterpret_cast<const float4*>(yreg + s_of_t * RS_Y);
            const float4* ds = reinterpret_cast<const float4*>(dor + s_of_t * RS_Y);
            float m8a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, m8b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // (d, du) and (u, uu) x 4 channels
            for (int px = st0 * 64 + lane; px < HW; px += stripes * 64) {
                const float4 y4 = ys[px * CV + cv];
                int di = px;
                if (pk > 1) { const int r = rs_fdiv(px, inv_w), c = px - r * W; di = rs_fdiv(r, inv_pk) * Wo + rs_fdiv(c, inv_pk); }
                float4 d4 = ds[di * CV + cv];
                d4.x *= pinv; d4.y *= pinv; d4.z *= pinv; d4.w *= pinv;
                const float u0 = (y4.x - mu.x) * rs.x, u1 = (y4.y - mu.y) * rs.y, u2 = (y4.z - mu.z) * rs.z, u3 = (y4.w - mu.w) * rs.w;
                m8a[0] += d4.x; m8a[1] += d4.y; m8a[2] += d4.z; m8a[3] += d4.w;
                m8a[4] += d4.x * u0; m8a[5] += d4.y * u1; m8a[6] += d4.z * u2; m8a[7] += d4.w * u3;
                if (!have_sums) {                                                    // (block-uniform) sum u, sum u^2: derived from the forward's sums of y, y^2 when it left them
                    m8b[0] += u0; m8b[1] += u1; m8b[2] += u2; m8b[3] += u3;
                    m8b[4] += u0 * u0; m8b[5] += u1 * u1; m8b[6] += u2 * u2; m8b[7] += u3 * u3;
                }
            }
            float ra[2], rb[2] = {0.f, 0.f};
            rs_reduce8(m8a, ra);
            if (!have_sums) rs_reduce8(m8b, rb);
            if (lane < 4) {
                const int vi = (lane & 1) * 4 + ((lane >> 1) & 1) * 2;                // values vi, vi + 1 = (quantity vi / 4, channel vi % 4)
                float* wr = s_wsum + wave * 4 * RS_MAXC;
                wr[(vi >> 2) * RS_MAXC + cv * 4 + (vi & 3)] = ra[0]; wr[(vi >> 2) * RS_MAXC + cv * 4 + (vi & 3) + 1] = ra[1];
                wr[(2 + (vi >> 2)) * RS_MAXC + cv * 4 + (vi & 3)] = rb[0]; wr[(2 + (vi >> 2)) * RS_MAXC + cv * 4 + (vi & 3) + 1] = rb[1];
            }
        }
        rs_barrier();
        stamp();
        // ---- the sample's algebra (norm_chain.hip, nc_sample_sums_kernel's epilogue) by thread (sample, channel) of wave 0; publish the block's T1, T2
        float mine = 0.f;
        if (t < RS_SPB * C) {
            const int ss = ac_ss, c = ac_c, ccv = c >> 2, j = c & 3, stripes = 8 / CV;
            float mq[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < stripes; ++k) {
                const float* wr = s_wsum + (ss * 8 + k * CV + ccv) * 4 * RS_MAXC + ccv * 4 + j;
#pragma unroll
                for (int q = 0; q < 4; ++q) mq[q] += wr[q * RS_MAXC];
            }
            const float gbn = s_par[oi][0][c], bbn = s_par[oi][1][c], brs = s_par[oi][3][c], ggn = s_par[oi][4][c];
            double u1 = (double)mq[2], u2 = (double)mq[3];
            if (have_sums) {                                                         // u = (y - mean) rstd: sum u = (sum y - HW mean) rstd, sum u^2 = (sum y^2 - 2 mean sum y + HW mean^2) rstd^2
                const double mean = (double)s_par[oi][2][c], rstd = (double)brs, hw = (double)HW, sy = (double)e_sy, syy = (double)e_syy;
                u1 = (sy - hw * mean) * rstd;
                u2 = (syy - 2.0 * mean * sy + hw * mean * mean) * rstd * rstd;
                // (the subtraction magnifies the fp32 rounding of sy, syy when a channel's mean is many standard deviations from zero.  Measured
                // with every convolution bias moved by 12 / 25, tools/offset_err.py, against the oracle in double precision: these derived sums 1.5e-4 / 5.3e-4
                // of a tensor's largest gradient, sums taken from the y tile instead 2.5e-3 / 1.4e-2, the per-layer kernels 7.3e-4 / 5.3e-3 — the forward's
                // GroupNorm statistics come from the SAME sy, syy, and the algebra below needs sums consistent with them, not better ones.)
            }
            const double m0 = (double)mq[0], m1 = (double)mq[1];
            const float a_bn = gbn * brs;
            double S1 = m0, S2 = m1, T1 = m0, T2 = m1;
            float P = a_bn, Q0 = 0.f, R0 = 0.f;
            if (o.groups > 0) {
                const double gr = (double)e_gr, al = (double)gbn * gr, be = ((double)bbn - (double)e_gm) * gr;
                const double s2 = al * m1 + be * m0;
                double A = (double)ggn * m0, Bs = (double)ggn * s2;
                for (int off = 1; off < cpg; off <<= 1) { A += __shfl_xor(A, off, 64); Bs += __shfl_xor(Bs, off, 64); }
                const double hw = (double)HW, ca = (double)ggn * gr, cb = A * o.inv_group * gr, cd = Bs * o.inv_group * gr;
                S2 = s2;
                T1 = ca * m0 - cb * hw - cd * (al * u1 + be * hw);
                T2 = ca * m1 - cb * u1 - cd * (al * u2 + be * u1);
                const double a = (double)gbn * (double)brs;
                P = (float)(a * ca); Q0 = (float)(-a * al * cd); R0 = (float)(-a * (cb + be * cd));
            }
            const bool v = ss == 0 ? valid0 : valid1;
            const float fS1 = v ? (float)S1 : 0.f, fS2 = v ? (float)S2 : 0.f, fT1 = v ? (float)T1 : 0.f, fT2 = v ? (float)T2 : 0.f;
            // the block's totals: the two samples of a channel sit C lanes apart (C: a power of two <= 16)
            const float bS1 = fS1 + __shfl_xor(fS1, C, 64), bS2 = fS2 + __shfl_xor(fS2, C, 64), bT1 = fT1 + __shfl_xor(fT1, C, 64), bT2 = fT2 + __shfl_xor(fT2, C, 64);
            mine = ss == 0 ? bT1 : bT2;                                             // quantity t of the synchronisation: T1 of channel t, T2 of channel t - C
            s_coef[ss][0][c] = P; s_coef[ss][1][c] = Q0; s_coef[ss][2][c] = R0; s_abn[ss][c] = a_bn;      // parked in LDS across the weight gradient below
            if (ss == 0 && o.groups > 0 && o.gn_part) { o.gn_part[((size_t)blockIdx.x * 2) * C + c] = bS1; o.gn_part[((size_t)blockIdx.x * 2 + 1) * C + c] = bS2; }
        }
        stamp();
        // ---- the batch totals of T1, T2: publish the block's partials, compute the PREVIOUS op's weight gradient while they travel (its dz and x tiles are
        // still intact), then reduce / collect
        rs_allreduce_publish(go, epoch, 2 * C, mine, ctl_mute == blockIdx.x + 1u);
        rs_allreduce_reduce_owned(go, go + RS_MAXQ * RS_MAXG, epoch, 2 * C, G, s_red, &s_fail);   // the few owner blocks: before their own weight gradient, so that nobody waits for it
        }
        // (s_wsum is free once wave 0 has read the moments in the algebra above: it says so in s_algebra, which the weight gradient's writers wait for)
        if (t == 0) *(volatile unsigned*)&s_algebra = (unsigned)oi + 1u;
        if (oi > 0 && p.op[oi - 1].dw_part != nullptr) rb_wgrad(p.op[oi - 1], xtt, dzt, s_wgq, s_wsum, &s_algebra, (unsigned)oi + 1u);
        else if (oi == 0 && fc_pg >= 0) {
            const rb_fc_args fa = {p.head.fc_dw, p.head.fc_db, p.head.classes, p.head.cc + p.head.cr, p.head.h * p.head.w, N, fc_pg};
            rb_fc_wgrad(fa, reinterpret_cast<const float*>(dzt), reinterpret_cast<float*>(xtt));      // (operands parked in the dz tile's region, the x tile is not in use yet)
        }
        if (tail) break;
        rs_barrier();
        stamp();
        // the next op's y is requested now (two float4 per thread: a sample holds at most RS_Y floats) and lands in LDS after the dz step, when this
        // layer's y is dead: the HBM round trip hides behind the all-reduce
        float4 ypre[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        int ypre_v4 = 0;
        if (oi + 1 < p.count) {
            const dcv_resident_bwd_op& nx = p.op[oi + 1];
            const int n = (int)blockIdx.x * RS_SPB + s_of_t;
            ypre_v4 = (nx.h * nx.w * nx.cout) >> 2;
            const float4* g = reinterpret_cast<const float4*>(nx.y) + (size_t)(n < N ? n : 0) * ypre_v4;
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int i = tt + k * 512; ypre[k] = g[i < ypre_v4 ? i : 0]; }
        }
        // this op's input x (for its weight gradient, which runs during the NEXT op's all-reduce): same scheme, NHWC with whole channel vectors
        const bool wg = o.dw_part != nullptr;
        const bool xvec = wg && !o.x_nchw && (o.cin & 3) == 0;
        float4 xpre[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        // (the model input, NCHW with <= 4 channels: a thread gathers the channels of its two pixels from the planes — one conflict-free float4 per pixel
        // into the tile instead of element-wise 4-byte stores, and requested here like the vector case instead of inside the tile phase)
        const bool xplanes = wg && o.x_nchw && o.cin <= 4 && HW <= 2 * 512;
        if (xvec) {
            const int n = (int)blockIdx.x * RS_SPB + s_of_t, v4 = (H * W * o.cin) >> 2;
            const float4* g = reinterpret_cast<const float4*>(o.x) + (size_t)(n < N ? n : 0) * v4;
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int i = tt + k * 512; xpre[k] = g[i < v4 ? i : 0]; }
        } else if (xplanes) {
            const int n = (int)blockIdx.x * RS_SPB + s_of_t;
            const float* g = o.x + (size_t)(n < N ? n : 0) * HW * o.cin;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int px = tt + k * 512, q = px < HW ? px : 0;
                xpre[k].x = g[q]; xpre[k].y = o.cin > 1 ? g[HW + q] : 0.f; xpre[k].z = o.cin > 2 ? g[2 * HW + q] : 0.f; xpre[k].w = o.cin > 3 ? g[3 * HW + q] : 0.f;
            }
        }
        if (!rs_allreduce_collect(go + RS_MAXQ * RS_MAXG, epoch, 2 * C, s_tot, s_w32, &s_fail)) {
            // a block of this grid is not resident: report (the error word: resident.check_status) and make the failure visible on the path a training loop
            // reads anyway — every gradient this launch owes is poisoned with NaN (the block's own chunks of every partial slab, the BatchNorm parameter
            // gradients, dz / dx where they leave the launch, the head's weight gradient), so the optimizer carries NaN into the weights and the NEXT loss the
            // host reads is NaN (the loss of this step was published by the head before anything could fail).  The launch generation still advances: a
            // later launch must not take this launch's granules for its own.
            const float qnan = __builtin_nanf("");
            if (t == 0) { __hip_atomic_store((rs_gu32*)(ctl + 1), (unsigned)oi + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            for (int k = 0; k < p.count; ++k) {
                const dcv_resident_bwd_op& q = p.op[k];
                const int wsize = q.cout * q.cin * q.k * q.k;
                if (q.dw_part) for (int e = t; e < RS_SPB * wsize; e += RS_T) q.dw_part[(size_t)blockIdx.x * RS_SPB * wsize + e] = qnan;
                if (q.db_part && t < RS_SPB * q.cout) q.db_part[(size_t)blockIdx.x * RS_SPB * q.cout + t] = qnan;
                if (blockIdx.x == 0 && t < q.cout) {
                    if (q.d_bn_gamma) q.d_bn_gamma[t] = qnan;
                    if (q.d_bn_beta) q.d_bn_beta[t] = qnan;
                    if (q.dz) q.dz[t] = qnan;
                    if (q.dx) q.dx[t] = qnan;
                    if (q.gn_part) q.gn_part[t] = qnan;
                }
            }
            if (blockIdx.x == 0 && t == 0) {
                if (p.head.enabled && p.head.fc_dw) p.head.fc_dw[0] = qnan;
                if (p.head.enabled && p.head.dref) p.head.dref[0] = qnan;
                __hip_atomic_store((rs_gu32*)ctl, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        stamp();
        if (t < RS_SPB * C) {
            const int ss = ac_ss, c = ac_c;
            const double t1 = s_tot[c], t2 = s_tot[C + c];
            const float abn = s_abn[ss][c];
            const float ad = abn * (float)(t2 * o.inv_count), ab = abn * (float)(t1 * o.inv_count);       // the batch terms (nc_param_finalize_kernel)
            s_coef[ss][1][c] -= ad; s_coef[ss][2][c] -= ab;
            if (blockIdx.x == 0 && ss == 0) { if (o.d_bn_beta) o.d_bn_beta[c] = (float)t1; if (o.d_bn_gamma) o.d_bn_gamma[c] = (float)t2; }
        }
        // the dz tile: zero it (halo; the scratch dirtied it), interior below
        rs_barrier();
        // the stock model's 4-channel 32 x 32 layers with every extent a constant (as the forward launch's apply step): only the halos are zeroed — the
        // interiors are written whole below, zeros for a missing sample —, no barrier between the two (disjoint words), one loop fills both tiles
        const bool fast_tiles = C == 4 && H == 32 && W == 32 && o.k == 5 && (pk == 1 || pk == 2) && (!wg || xvec || xplanes);
        // ... and its 16-channel 16 x 16 layers (3x3: channel-vector planes of 18 x 18; input of 16 or 4 channels)
        const bool fast16 = C == 16 && H == 16 && W == 16 && o.k == 3 && (pk == 1 || pk == 2) && (!wg || (xvec && (o.cin == 16 || o.cin == 4)));
        if (fast16) {
            auto halo = [&](int e, int& r, int& c) {                                 // the e-th of a plane's 68 halo vectors
                if (e < 36) { r = e >= 18 ? 17 : 0; c = e - (e >= 18 ? 18 : 0); }
                else { const int e2 = e - 36; r = 1 + (e2 >> 1); c = (e2 & 1) ? 17 : 0; }
            };
            if (t < 8 * 68) { const int pl = (t * 964) >> 16, e = t - pl * 68; int r, c; halo(e, r, c); dzt[(pl >> 2) * RS_XIN + ((pl & 3) * 18 + r) * 18 + c] = make_float4(0.f, 0.f, 0.f, 0.f); }
            if (wg) {
                const int cvi = o.cin >> 2;                                          // 4 or 1 planes per sample
                if (t < RS_SPB * cvi * 68) {
                    const int pl = (t * 964) >> 16, e = t - pl * 68, hs = cvi == 4 ? pl >> 2 : pl, cvp = cvi == 4 ? pl & 3 : 0;
                    int r, c; halo(e, r, c);
                    xtt[hs * RS_XIN + (cvp * 18 + r) * 18 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else if (fast_tiles) {
            for (int i = t; i < RS_SPB * 272; i += RS_T) {
                const int hs = i >= 272 ? 1 : 0, e = i - hs * 272;
                int r, c;
                if (e < 144) { r = (e * 1821) >> 16; c = e - r * 36; if (r >= 2) r += 32; }                     // (e / 36 for e < 144)
                else { const int e2 = e - 144; r = 2 + (e2 >> 2); c = e2 & 3; if (c >= 2) c += 32; }
                dzt[hs * RS_XIN + r * 36 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (wg) xtt[hs * RS_XIN + r * 36 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            for (int i = t; i < RS_SPB * RS_XIN; i += RS_T) { dzt[i] = make_float4(0.f, 0.f, 0.f, 0.f); if (wg) xtt[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
            rs_barrier();
        }
        stamp();
        if (fast16) {
            const int ss = s_of_t, n = (int)blockIdx.x * RS_SPB + ss, cv = tt & 3;
            const float4 Pv = *reinterpret_cast<const float4*>(s_coef[ss][0] + cv * 4), Qv = *reinterpret_cast<const float4*>(s_coef[ss][1] + cv * 4), Rv = *reinterpret_cast<const float4*>(s_coef[ss][2] + cv * 4);
            const float4 mu = *reinterpret_cast<const float4*>(s_mr[0] + cv * 4), rs = *reinterpret_cast<const float4*>(s_mr[1] + cv * 4);
            const bool relu_like = o.act == DCV_ACT_RELU || o.act == DCV_ACT_LEAKY_RELU;
            const float neg = o.act == DCV_ACT_RELU ? 0.f : o.slope;
            auto fill = [&](auto relu_c) {
                auto ag = [&](float y) { if constexpr (decltype(relu_c)::value) return y > 0.f ? 1.f : neg; else return rb_act_grad(y, o.act, o.slope); };
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = tt + k * 512, px = e >> 2, r = px >> 4, c = px & 15, at = ss * RS_XIN + (cv * 18 + r + 1) * 18 + c + 1;
                    if (wg) {
                        if (o.cin == 16) xtt[at] = n < N ? xpre[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                        else if (e < 256) xtt[ss * RS_XIN + ((e >> 4) + 1) * 18 + (e & 15) + 1] = n < N ? xpre[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    const float4 y4 = reinterpret_cast<const float4*>(yreg + ss * RS_Y)[e];
                    const int di = pk > 1 ? (r >> 1) * 8 + (c >> 1) : px;
                    const float4 d4 = reinterpret_cast<const float4*>(dor + ss * RS_Y)[di * 4 + cv];
                    float4 z;
                    z.x = (Pv.x * (d4.x * pinv) + Qv.x * ((y4.x - mu.x) * rs.x) + Rv.x) * ag(y4.x);
                    z.y = (Pv.y * (d4.y * pinv) + Qv.y * ((y4.y - mu.y) * rs.y) + Rv.y) * ag(y4.y);
                    z.z = (Pv.z * (d4.z * pinv) + Qv.z * ((y4.z - mu.z) * rs.z) + Rv.z) * ag(y4.z);
                    z.w = (Pv.w * (d4.w * pinv) + Qv.w * ((y4.w - mu.w) * rs.w) + Rv.w) * ag(y4.w);
                    if (n >= N) z = make_float4(0.f, 0.f, 0.f, 0.f);
                    else if (o.dz != nullptr) reinterpret_cast<float4*>(o.dz)[(size_t)n * 1024 + e] = z;
                    dzt[at] = z;
                }
            };
            if (relu_like) fill(std::true_type{}); else fill(std::false_type{});
        } else if (fast_tiles) {
            const int ss = s_of_t, n = (int)blockIdx.x * RS_SPB + ss;
            const float4 Pv = *reinterpret_cast<const float4*>(s_coef[ss][0]), Qv = *reinterpret_cast<const float4*>(s_coef[ss][1]), Rv = *reinterpret_cast<const float4*>(s_coef[ss][2]);
            const float4 mu = *reinterpret_cast<const float4*>(s_mr[0]), rs = *reinterpret_cast<const float4*>(s_mr[1]);
            // (the activation decided once: ReLU / LeakyReLU — y > 0 ? 1 : slope, slope 0 for ReLU, the same values dcv_act_grad_from_out returns — or the switch)
            const bool relu_like = o.act == DCV_ACT_RELU || o.act == DCV_ACT_LEAKY_RELU;
            const float neg = o.act == DCV_ACT_RELU ? 0.f : o.slope;
            auto fill = [&](auto relu_c) {
                auto ag = [&](float y) { if constexpr (decltype(relu_c)::value) return y > 0.f ? 1.f : neg; else return rb_act_grad(y, o.act, o.slope); };
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = tt + k * 512, r = e >> 5, c = e & 31, at = ss * RS_XIN + (r + 2) * 36 + c + 2;
                    if (wg) xtt[at] = n < N ? xpre[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 y4 = reinterpret_cast<const float4*>(yreg + ss * RS_Y)[e];
                    const int di = pk > 1 ? (r >> 1) * 16 + (c >> 1) : e;
                    const float4 d4 = reinterpret_cast<const float4*>(dor + ss * RS_Y)[di];
                    float4 z;
                    z.x = (Pv.x * (d4.x * pinv) + Qv.x * ((y4.x - mu.x) * rs.x) + Rv.x) * ag(y4.x);
                    z.y = (Pv.y * (d4.y * pinv) + Qv.y * ((y4.y - mu.y) * rs.y) + Rv.y) * ag(y4.y);
                    z.z = (Pv.z * (d4.z * pinv) + Qv.z * ((y4.z - mu.z) * rs.z) + Rv.z) * ag(y4.z);
                    z.w = (Pv.w * (d4.w * pinv) + Qv.w * ((y4.w - mu.w) * rs.w) + Rv.w) * ag(y4.w);
                    if (n >= N) z = make_float4(0.f, 0.f, 0.f, 0.f);
                    else if (o.dz != nullptr) reinterpret_cast<float4*>(o.dz)[(size_t)n * 1024 + e] = z;
                    dzt[at] = z;
                }
            };
            if (relu_like) fill(std::true_type{}); else fill(std::false_type{});
        } else {
        if (wg) {   // this op's x -> its zero-padded tile
            const int pad = o.k / 2, XW = W + 2 * pad, XH = H + 2 * pad, n = (int)blockIdx.x * RS_SPB + s_of_t;
            if (xvec) {
                const int CVI = o.cin >> 2, v4 = H * W * CVI;
                const float inv_cvi = 1.f / (float)CVI;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = tt + k * 512;
                    if (e < v4 && n < N) { const int px = rs_fdiv(e, inv_cvi), cv = e - px * CVI, r = rs_fdiv(px, inv_w), c = px - r * W; xtt[s_of_t * RS_XIN + (cv * XH + r + pad) * XW + c + pad] = xpre[k]; }
                }
            } else if (xplanes) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int px = tt + k * 512;
                    if (px < HW && n < N) { const int r = rs_fdiv(px, inv_w), c = px - r * W; xtt[s_of_t * RS_XIN + (r + pad) * XW + c + pad] = xpre[k]; }
                }
            } else if (n < N) {                                                      // NCHW / partial channel vector (the model input): scalar elements
                const int total = H * W * o.cin;
                const float inv_hw = 1.f / (float)HW, inv_ci = 1.f / (float)o.cin;
                const float* src = o.x + (size_t)n * total;
                float* dst = reinterpret_cast<float*>(xtt + s_of_t * RS_XIN);
                for (int e0 = 0; e0 < total; e0 += 8 * 512) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const int e = e0 + tt + k * 512; v[k] = src[e < total ? e : 0]; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int e = e0 + tt + k * 512;
                        if (e < total) {
                            int c, px;
                            if (o.x_nchw) { c = rs_fdiv(e, inv_hw); px = e - c * HW; } else { px = rs_fdiv(e, inv_ci); c = e - px * o.cin; }
                            const int r = rs_fdiv(px, inv_w), col = px - r * W;
                            dst[(((c >> 2) * XH + r + pad) * XW + col + pad) * 4 + (c & 3)] = v[k];
                        }
                    }
                }
            }
        }
        // ---- dz = act'(y) (P d + Q u + R): -> the padded tile of the data gradient (and of this layer's in-launch weight gradient); -> HBM only for a layer
        // whose weight gradient is a launch of its own (o.dz set)
        {
            const int pad = o.k / 2, XW = W + 2 * pad, XH = H + 2 * pad;
            const float inv_cv = 1.f / (float)CV;
            const int per = HW * CV;
            // waves 0 - 7: the first sample, 8 - 15: the second; a thread's channel vector is fixed (512 is a multiple of CV): coefficients in registers
            const int ss = s_of_t, cv = tt & (CV - 1), n = (int)blockIdx.x * RS_SPB + ss;
            const float4 Pv = *reinterpret_cast<const float4*>(s_coef[ss][0] + cv * 4), Qv = *reinterpret_cast<const float4*>(s_coef[ss][1] + cv * 4);
            const float4 Rv = *reinterpret_cast<const float4*>(s_coef[ss][2] + cv * 4);
            const float4 mu = *reinterpret_cast<const float4*>(s_mr[0] + cv * 4), rs = *reinterpret_cast<const float4*>(s_mr[1] + cv * 4);
            for (int e = tt; e < per; e += 512) {
                const int px = rs_fdiv(e, inv_cv), r = rs_fdiv(px, inv_w), c = px - r * W;
                const float4 y4 = reinterpret_cast<const float4*>(yreg + ss * RS_Y)[e];
                const int di = pk > 1 ? rs_fdiv(r, inv_pk) * Wo + rs_fdiv(c, inv_pk) : px;
                float4 d4 = reinterpret_cast<const float4*>(dor + ss * RS_Y)[di * CV + cv];
                float4 z;
                z.x = (Pv.x * (d4.x * pinv) + Qv.x * ((y4.x - mu.x) * rs.x) + Rv.x) * rb_act_grad(y4.x, o.act, o.slope);
                z.y = (Pv.y * (d4.y * pinv) + Qv.y * ((y4.y - mu.y) * rs.y) + Rv.y) * rb_act_grad(y4.y, o.act, o.slope);
                z.z = (Pv.z * (d4.z * pinv) + Qv.z * ((y4.z - mu.z) * rs.z) + Rv.z) * rb_act_grad(y4.z, o.act, o.slope);
                z.w = (Pv.w * (d4.w * pinv) + Qv.w * ((y4.w - mu.w) * rs.w) + Rv.w) * rb_act_grad(y4.w, o.act, o.slope);
                if (n >= N) z = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (o.dz != nullptr) reinterpret_cast<float4*>(o.dz)[(size_t)n * per + e] = z;      // only when a weight-gradient launch will read it (dw_part == NULL)
                dzt[ss * RS_XIN + (cv * XH + r + pad) * XW + c + pad] = z;
            }
        }
        }
        rs_barrier();
        stamp();
        // ---- the next op's y -> LDS (this layer's y is dead), then the data gradient: dz tile -> DO = the next op's dout
        if (oi + 1 < p.count) {
            const int n = (int)blockIdx.x * RS_SPB + s_of_t;
            float4* d = reinterpret_cast<float4*>(yreg + s_of_t * RS_Y);
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int i = tt + k * 512; if (i < ypre_v4) d[i] = n < N ? ypre[k] : make_float4(0.f, 0.f, 0.f, 0.f); }
        }
        if (o.need_dx) {
            dcv_resident_op cv{};
            cv.cin = o.cout; cv.cout = o.cin; cv.k = o.k; cv.h = H; cv.w = W; cv.act = DCV_ACT_NONE; cv.filt = o.filt; cv.bias = nullptr;
            const int var = rb_dgrad_variant(o.cin, o.cout, o.k, o.w);
            if (var == 1) {
                int slot = 0;                                                       // this op's filter: slot = the number of 4 <- 4 data gradients before it
                for (int k = 0; k < oi; ++k) if (p.op[k].need_dx && rb_dgrad_variant(p.op[k].cin, p.op[k].cout, p.op[k].k, p.op[k].w) == 1) ++slot;
                const float* wl = s_filt[slot < RB_FILT_SLOTS ? slot : 0];
                if (o.k == 3) rs_conv_c4_co4<3, 4, 0>(cv, dzt, dor, wl, nullptr, s_dummy);
                else if (o.w != 32 || o.h != 32) rs_conv_c4_co4<5, 4, 0>(cv, dzt, dor, wl, nullptr, s_dummy);
                else {
                    // the next op's moments in this convolution's epilogue: same map, no pooling in between, nothing added to the result, and the forward
                    // left that op's sums of y, y^2 (the moments pass would otherwise have to produce sum u, sum u^2 as well)
                    const bool fuse = oi + 1 < p.count && p.op[oi + 1].pool <= 1 && p.op[oi + 1].sample_sums != nullptr && o.addend == nullptr && p.op[oi + 1].cout == 4
                                      && p.op[oi + 1].h == H && p.op[oi + 1].w == W;
                    if (fuse) {
                        const rs_f4 nm = *reinterpret_cast<const rs_f4*>(s_par[oi + 1][2]), nr = *reinterpret_cast<const rs_f4*>(s_par[oi + 1][3]);
                        rs_conv_c4_co4_32<5, 4, true>(cv, dzt, dor, wl, nullptr, s_wsum, yreg, nm, nr);
                        moments_next = true;
                    } else rs_conv_c4_co4_32<5, 4>(cv, dzt, dor, wl, nullptr, s_dummy);
                }
            }
            else if (var == 4) rb_dgrad_c16_co4(o.filt, dzt, dor);
            else { float4 b4[9]; rs_load_bfrag_prepared(o.filt, b4); rs_conv_c16_mfma(cv, dzt, dor, b4, s_dummy); }
            rs_barrier();
            if (o.addend != nullptr || o.dx != nullptr) {
                const int v4 = (HW * o.cin) >> 2, n = (int)blockIdx.x * RS_SPB + s_of_t;
                float4* d = reinterpret_cast<float4*>(dor + s_of_t * RS_Y);
                for (int i = tt; i < v4 && n < N; i += 512) {
                    float4 v = d[i];
                    if (o.addend) { const float4 a = reinterpret_cast<const float4*>(o.addend)[(size_t)n * v4 + i]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; d[i] = v; }
                    if (o.dx) reinterpret_cast<float4*>(o.dx)[(size_t)n * v4 + i] = v;
                }
            }
        }
        rs_barrier();
        stamp();
        moments_done = moments_next;
    }
    stamp();
    if (blockIdx.x == ctl_stamp && t == 0) stamps[0] = (unsigned long long)n_stamp;
    if (blockIdx.x == 0 && t == 0) __hip_atomic_store((rs_gu32*)ctl, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

inline bool rb_program_ok(const dcv_resident_bwd_program* p) {
    if (!p || p->n <= 0 || p->n > RS_SPB * RS_MAXG || p->count <= 0 || p->count > DCV_RESIDENT_BWD_MAX_OPS) return false;
    if (p->head.enabled) {
        const dcv_resident_bwd_head& hd = p->head;
        const dcv_resident_bwd_op& o = p->op[0];
        const int pk = o.pool > 0 ? o.pool : 1;
        if (hd.classes <= 0 || hd.classes > 16 || hd.cc != o.cout || hd.h != o.h / pk || hd.w != o.w / pk || hd.cr <= 0 || hd.ref_h <= 0 || hd.ref_w <= 0) return false;
        if (hd.h * hd.w * hd.cr > RS_Y || hd.ref_h * hd.ref_w * hd.cr > 16 * RS_Y || !hd.wp || !hd.dref || hd.act < DCV_ACT_NONE || hd.act > DCV_ACT_TANH) return false;
        if (hd.act != DCV_ACT_NONE && !hd.logits) return false;
        if ((size_t)(hd.ref_h + hd.ref_w) * RB_TAPS * 8 + (size_t)RS_SPB * RS_Y * 4 > (size_t)RS_SPB * RS_XIN * 16 || hd.ref_h + hd.ref_w > RS_T) return false;
        if (!rb_resize_taps_ok(hd.ref_h, hd.h, hd.align_corners) || !rb_resize_taps_ok(hd.ref_w, hd.w, hd.align_corners)) return false;
        if (hd.ce_target && (!hd.logits || !hd.ce_loss || p->n > 512 || hd.classes > RB_FC_MAXO || (hd.pub_seq && (!hd.pub_mailbox || !hd.pub_vals || hd.pub_slots <= 0 || hd.pub_slots > DCV_MAX_COPY_JOBS)))) return false;
        if (!hd.ce_target && !hd.dlogits) return false;
        if (hd.fc_dw && (!hd.fc_x || !rb_fc_wgrad_ok(p->n, hd.h * hd.w * (hd.cc + hd.cr), hd.classes))) return false;
    } else if (!p->op[0].dout) return false;
    int c4_dgrads = 0;
    for (int i = 0; i < p->count; ++i) {
        const dcv_resident_bwd_op& o = p->op[i];
        if (o.cout != 4 && o.cout != 8 && o.cout != 16) return false;
        if (o.h <= 0 || o.w <= 0 || o.h * o.w * o.cout > RS_Y || (o.k & 1) == 0 || o.k > 5) return false;
        const int pad = o.k / 2;
        if ((o.cout >> 2) * (o.h + 2 * pad) * (o.w + 2 * pad) > RS_XIN) return false;
        if (o.groups < 0 || (o.groups > 0 && (o.cout % o.groups != 0 || ((o.cout / o.groups) & (o.cout / o.groups - 1)) || !o.gn_mean || !o.gn_rstd))) return false;
        if (o.pool < 0 || o.pool == 1 || (o.pool > 0 && (o.h % o.pool != 0 || o.w % o.pool != 0 || o.groups == 0))) return false;
        if (!o.y || (!o.dz && !o.dw_part) || !o.bn_mean || !o.bn_rstd || o.act < DCV_ACT_NONE || o.act > DCV_ACT_TANH) return false;   // dz in HBM: optional with the weight gradient in the launch
        if (o.dw_part) {
            const int wv = rb_wgrad_variant(o.cin, o.cout, o.k, o.w);
            if (wv == 0 || !o.x || (wv == 1 && !rb_wgrad_c4_rows_ok(o.h, o.w, o.k)) || ((o.cin + 3) >> 2) * (o.h + 2 * pad) * (o.w + 2 * pad) > RS_XIN) return false;
            if (o.x_nchw == 0 && (o.cin & 3) != 0 && o.cin > 4) return false;
        }
        if (o.need_dx) {
            const int var = rb_dgrad_variant(o.cin, o.cout, o.k, o.w);
            if (var == 0 || !o.filt) return false;
            if (var == 1 && ++c4_dgrads > RB_FILT_SLOTS) return false;                              // (its filter's place in LDS)
            if (var == 4 && o.h != 16) return false;
            if (o.h * o.w * o.cin > RS_Y || (o.cin & 3)) return false;
        } else if (i + 1 < p->count) return false;                   // only the last op (the model's first layer) may skip its data gradient
        if (i + 1 < p->count) {
            const dcv_resident_bwd_op& nx = p->op[i + 1];
            const int pk = nx.pool > 0 ? nx.pool : 1;
            if (nx.cout != o.cin || nx.h / pk != o.h || nx.w / pk != o.w) return false;      // this op's data gradient is the next op's dout
        }
    }
    return true;
}

inline bool rs_layer_ok(const dcv_resident_op& o, bool first) {
    const int var = rs_variant(o.cin, o.cout, o.k, o.w);
    if (var == 0 || o.h <= 0 || o.w <= 0 || o.cin <= 0) return false;
    const int pad = o.k / 2, cvin = (o.cin + 3) / 4;
    if (cvin * (o.h + 2 * pad) * (o.w + 2 * pad) > RS_XIN || o.h * o.w * o.cout > RS_Y) return false;
    if (var == 2 && (o.h * o.w) % 64 != 0) return false;
    if (!first && (o.cin & 3)) return false;                      // only the staged model input may have a partial channel vector
    if (o.act < DCV_ACT_NONE || o.act > DCV_ACT_TANH || !o.filt || !o.y || !o.out) return false;
    if (o.bn && !(o.bn_mean && o.bn_rstd)) return false;
    if (o.groups < 0 || (o.groups > 0 && (o.cout % o.groups != 0 || ((o.cout / o.groups) & (o.cout / o.groups - 1)) || !o.gn_mean || !o.gn_rstd))) return false;   // channels per group: a power of two
    if (o.pool < 0 || o.pool == 1 || (o.pool > 0 && (o.h % o.pool != 0 || o.w % o.pool != 0))) return false;
    return true;
}

inline bool rs_program_ok(const dcv_resident_program* p) {
    if (!p || p->n <= 0 || p->n > RS_SPB * RS_MAXG || p->count <= 0 || p->count > DCV_RESIDENT_MAX_OPS) return false;
    if (p->op[0].kind != DCV_RES_LAYER || !p->op[0].x) return false;
    if (p->mailbox && (p->x_job < 0 || p->x_job >= DCV_MAX_COPY_JOBS || !p->op[0].bn)) return false;      // (the other jobs are copied in the first layer's all-reduce wait)
    int c = 0, h = 0, w = 0;                                      // the current tensor
    for (int i = 0; i < p->count; ++i) {
        const dcv_resident_op& o = p->op[i];
        if (o.kind == DCV_RES_LAYER) {
            if (!rs_layer_ok(o, i == 0)) return false;
            if (rs_variant(o.cin, o.cout, o.k, o.w) == 1 && i >= RS_FILT_OPS) return false;      // (its filter's place in LDS)
            if (i > 0 && (p->op[i - 1].kind != DCV_RES_LAYER || o.cin != c || o.h != h || o.w != w)) return false;
            const int pk = o.pool > 0 ? o.pool : 1;
            c = o.cout; h = o.h / pk; w = o.w / pk;
        } else if (o.kind == DCV_RES_LINK) {
            if (i == 0 || p->op[i - 1].kind != DCV_RES_LAYER || o.cin != c || (c & 3) || o.h != h || o.w != w || !o.ref || !o.out || o.ref_c <= 0 || (o.ref_c & 3) || o.ref_h <= 0 || o.ref_w <= 0) return false;
            if (i + 1 < p->count && p->op[i + 1].kind != DCV_RES_FC) return false;
            bool found = false;                                   // ref must be an earlier layer's output (this block wrote its samples' part)
            for (int k = 0; k < i; ++k) if (p->op[k].kind == DCV_RES_LAYER && p->op[k].out == o.ref) found = true;
            if (!found || h * w * (c + o.ref_c) > RS_Y) return false;
            c += o.ref_c;
        } else if (o.kind == DCV_RES_FC) {
            if (i == 0 || o.cin != c || o.h != h || o.w != w || !o.filt || !o.out || o.cout <= 0 || o.cout > 16 || ((c * h * w) & 3) || c * h * w > RS_Y) return false;
            if (p->op[i - 1].kind == DCV_RES_LAYER && (size_t)c * h * w > (size_t)RS_XIN * 4) return false;
            if (i + 1 != p->count) return false;
            if (o.act < DCV_ACT_NONE || o.act > DCV_ACT_TANH) return false;
        } else return false;
    }
    return true;
}

// Both launches put one 1024-thread block on every CU they use and their blocks wait for each other: the grid must be co-resident.  With the current device's
// CU count known, a program whose grid is larger is refused here (the caller falls back to the per-layer kernels) instead of timing out in its first
// synchronisation; without a device (the build container) the count is unknown and nothing is refused.
inline bool rs_grid_fits(int n) {
    static int cus = -2;
    if (cus == -2) {
        int dev = 0, v = 0;
        cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : -1;
        (void)hipGetLastError();
    }
    return cus < 0 || (n + RS_SPB - 1) / RS_SPB <= cus;
}

}  // namespace

extern "C" {

int dcv_resident_supported(const dcv_resident_program* p) { return (rs_program_ok(p) && rs_grid_fits(p->n)) ? 1 : 0; }

size_t dcv_resident_ws_bytes(void) { return RS_CTL_BYTES + (size_t)DCV_RESIDENT_MAX_OPS * RS_GRAN_OP * sizeof(rs_u64); }

int dcv_resident_forward(const dcv_resident_program* p, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(p && ws);
    if (!rs_program_ok(p) || !rs_grid_fits(p->n)) return DCV_ERR_UNSUPPORTED;
    if (ws_bytes < dcv_resident_ws_bytes()) return DCV_ERR_WORKSPACE;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resident_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RS_SMEM) != hipSuccess) return DCV_ERR_LAUNCH;
        attr_set = true;
    }
    const unsigned grid = (unsigned)((p->n + RS_SPB - 1) / RS_SPB);
    hipLaunchKernelGGL(resident_fwd_kernel, dim3(grid), dim3(RS_T), RS_SMEM, dcv_stream(stream), *p, reinterpret_cast<unsigned*>(ws),
                       reinterpret_cast<rs_u64*>(reinterpret_cast<char*>(ws) + RS_CTL_BYTES));
    DCV_RETURN_LAUNCH();
}

int dcv_resident_bwd_supported(const dcv_resident_bwd_program* p) { return (rb_program_ok(p) && rs_grid_fits(p->n)) ? 1 : 0; }

int dcv_resident_head_wgrad_supported(int n, int k, int classes) { return rb_fc_wgrad_ok(n, k, classes) ? 1 : 0; }

int dcv_resident_wgrad_supported(int cin, int cout, int k, int h, int w) {
    const int wv = rb_wgrad_variant(cin, cout, k, w);
    return (wv != 0 && (wv != 1 || rb_wgrad_c4_rows_ok(h, w, k))) ? 1 : 0;
}

int dcv_resident_backward(const dcv_resident_bwd_program* p, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(p && ws);
    if (!rb_program_ok(p) || !rs_grid_fits(p->n)) return DCV_ERR_UNSUPPORTED;
    if (ws_bytes < dcv_resident_ws_bytes()) return DCV_ERR_WORKSPACE;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(resident_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RB_SMEM) != hipSuccess) return DCV_ERR_LAUNCH;
        attr_set = true;
    }
    const unsigned grid = (unsigned)((p->n + RS_SPB - 1) / RS_SPB);
    hipLaunchKernelGGL(resident_bwd_kernel, dim3(grid), dim3(RS_T), RB_SMEM, dcv_stream(stream), *p, reinterpret_cast<unsigned*>(ws),
                       reinterpret_cast<rs_u64*>(reinterpret_cast<char*>(ws) + RS_CTL_BYTES));
    DCV_RETURN_LAUNCH();
}

int dcv_resident_status(const void* ws, void* stream, int* err) {
    DCV_CHECK_ARG(ws && err);
    unsigned ctl[2] = {0, 0};
    if (hipMemcpyAsync(ctl, ws, sizeof(ctl), hipMemcpyDeviceToHost, dcv_stream(stream)) != hipSuccess) return DCV_ERR_LAUNCH;
    if (hipStreamSynchronize(dcv_stream(stream)) != hipSuccess) return DCV_ERR_LAUNCH;
    *err = (int)ctl[1];
    return DCV_OK;
}

}  // extern "C"
