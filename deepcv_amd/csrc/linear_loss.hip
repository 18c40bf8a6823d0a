// linear_loss.hip — fully-connected layer (fwd / dgrad / wgrad) and softmax cross-entropy (fwd / bwd).
//
// The classifier head is tiny next to the convolutions (stock: 1280 -> 10, 25.6 kFLOP/img), so these kernels are
// bandwidth/latency-bound wave-level dot products with fused bias + activation, not an MFMA GEMM.  Reductions are fixed-order.
//
// Replaces torch.nn.Linear (reference submodule_creators.py:268-269) and torch.nn.CrossEntropyLoss() (classification/image.py:70).
#include "common.h"
#include <float.h>

namespace {

#define LIN_OCHUNK 64
// one wave per sample row; lanes stride over K.  The row of x is read ONCE into registers (up to XR vectors per lane) and 8 output features are
// accumulated together, so that the loads of 8 weight rows are in flight at once and the 8 wave reductions are independent (a per-feature loop
// re-reading x was a chain of 10 dependent global-memory latencies for the 10-class head).
template <int VEC, int OB>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, int N, int K, int O, int act, float slope) {
    constexpr int XR = 8;                                      // OB: output features accumulated together (10 for the 10-class head: one pass, no idle slots)
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* xr = x + (size_t)n * K;
    const int KV = K / VEC;
    const int iters = (KV + 63) / 64;
    if (iters <= XR) {
        float xv[XR][VEC];
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int k = lane + 64 * i;
            if (i < iters && k < KV) {
                if constexpr (VEC == 4) { const float4 a = reinterpret_cast<const float4*>(xr)[k]; xv[i][0] = a.x; xv[i][1] = a.y; xv[i][2] = a.z; xv[i][3] = a.w; }
                else xv[i][0] = xr[k];
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) xv[i][v] = 0.f;
            }
        }
        // blockIdx.y selects a chunk of LIN_OCHUNK output features (wide heads: 1000 classes -> 16 chunks of waves instead of one serial loop)
        const int o_begin = blockIdx.y * LIN_OCHUNK, o_end = (o_begin + LIN_OCHUNK < O) ? o_begin + LIN_OCHUNK : O;
        for (int o0 = o_begin; o0 < o_end; o0 += OB) {
            float acc[OB];
#pragma unroll
            for (int j = 0; j < OB; ++j) acc[j] = 0.f;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                if (i < iters) {
                    const int k = lane + 64 * i;
                    const int kc = k < KV ? k : KV - 1;                // clamped: xv is zero there
#pragma unroll
                    for (int j = 0; j < OB; ++j) {
                        const int o = (o0 + j < O) ? o0 + j : O - 1;
                        if constexpr (VEC == 4) {
                            const float4 b = reinterpret_cast<const float4*>(w + (size_t)o * K)[kc];
                            acc[j] += xv[i][0] * b.x + xv[i][1] * b.y + xv[i][2] * b.z + xv[i][3] * b.w;
                        } else acc[j] += xv[i][0] * w[(size_t)o * K + kc];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < OB; ++j) acc[j] = dcv_wave_sum(acc[j]);
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < OB; ++j)
                    if (o0 + j < O) y[(size_t)n * O + o0 + j] = dcv_act_apply(acc[j] + (bias ? bias[o0 + j] : 0.f), act, slope);
            }
        }
        return;
    }
    for (int o = blockIdx.y * LIN_OCHUNK; o < O && o < (int)(blockIdx.y + 1) * LIN_OCHUNK; ++o) {
        const float* wr = w + (size_t)o * K;
        float acc = 0.f;
        for (int k = lane; k < KV; k += 64) {
            if constexpr (VEC == 4) {
                const float4 a = reinterpret_cast<const float4*>(xr)[k]; const float4 b = reinterpret_cast<const float4*>(wr)[k];
                acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            } else acc += xr[k] * wr[k];
        }
        acc = dcv_wave_sum(acc);
        if (lane == 0) y[(size_t)n * O + o] = dcv_act_apply(acc + (bias ? bias[o] : 0.f), act, slope);
    }
}

// Wide heads on the matrix cores (round 4, last session): y[16 rows][16 features] per block on v_mfma_f32_16x16x4_f32, the K steps of 16 dealt round-robin to the
// block's four waves and their accumulators summed through LDS in wave order (deterministic).  A lane's float4 of x (row li, k = 16 s + 4 lg ..) and of w (feature
// li, same k) are the A / B fragments of four MFMAs.  The kernels above keep one (or four) sample rows per WAVE and reduce every (row, feature) product across the
// wave: 26 us for 512 x 512 -> 100 (cifar100_deep), 77 us for 512 x 512 -> 1000 (resnet18_style) — a 0.05 / 0.5 GFLOP product.
typedef float lin_f32x4 __attribute__((ext_vector_type(4)));
inline bool mfma_head() { static int on = -1; if (on < 0) { const char* e = getenv("DCV_LINEAR_MFMA"); on = e ? atoi(e) : 1; } return on != 0; }
__global__ __launch_bounds__(256) void linear_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y, int N, int K, int O, int act, float slope) {
    __shared__ lin_f32x4 red[3][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16, o0 = blockIdx.y * 16;
    const int row = n0 + li < N ? n0 + li : N - 1, feat = o0 + li < O ? o0 + li : O - 1;      // (clamped: their results are not stored)
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * K) + lg;
    const float4* wr = reinterpret_cast<const float4*>(w + (size_t)feat * K) + lg;
    lin_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int steps = K >> 4;
#pragma unroll 4
    for (int s = wave; s < steps; s += 4) {
        const float4 a = xr[4 * s], b = wr[4 * s];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave > 0) return;
    // acc[reg]: row n0 + 4 lg + reg, feature o0 + li
#pragma unroll
    for (int k = 0; k < 3; ++k) { const lin_f32x4 v = red[k][lane]; acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3]; }
    const int o = o0 + li;
    if (o >= O) return;
    const float bv = bias ? bias[o] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int n = n0 + 4 * lg + reg;
        if (n < N) y[(size_t)n * O + o] = dcv_act_apply(acc[reg] + bv, act, slope);
    }
}

// Wide heads at large batches (round 4: 512 x 2048 -> 100 of cifar100_deep, 512 x 512 -> 1000 of resnet18_style): with one sample row per wave every wave reads
// the whole weight chunk — 410 MB / 1 GB of L2 traffic for a 0.2 / 0.5 GFLOP product, 27 / 77 us.  Here a wave owns RW = 4 sample rows: each weight vector it
// loads meets four rows of x in registers (a quarter of the traffic); per (row, feature) the same products are summed in the same order as linear_fwd_kernel<4, 8>
// (bit-identical results).
template <int RW, int XR>
__global__ __launch_bounds__(256) void linear_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y, int N, int K, int O, int act, float slope) {
    constexpr int OB = 8;                                       // XR: float4 of a row per lane (K <= 256 XR)
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RW;
    if (n0 >= N) return;
    const int KV = K >> 2, iters = (KV + 63) / 64;               // iters <= XR (the launcher checks)
    float4 xv[RW][XR];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const float4* xr = reinterpret_cast<const float4*>(x + (size_t)(n0 + r < N ? n0 + r : n0) * K);
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int k = lane + 64 * i;
            xv[r][i] = (i < iters && k < KV) ? xr[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // (eight output features per wave: a 0.2 GFLOP product is latency-bound — as many waves as the traffic allows)
    const int o_begin = blockIdx.y * OB, o_end = (o_begin + OB < O) ? o_begin + OB : O;
    for (int o0 = o_begin; o0 < o_end; o0 += OB) {
        float acc[RW][OB];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int j = 0; j < OB; ++j) acc[r][j] = 0.f;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            if (i < iters) {
                const int k = lane + 64 * i, kc = k < KV ? k : KV - 1;
#pragma unroll
                for (int j = 0; j < OB; ++j) {
                    const int o = (o0 + j < O) ? o0 + j : O - 1;
                    const float4 b = reinterpret_cast<const float4*>(w + (size_t)o * K)[kc];
#pragma unroll
                    for (int r = 0; r < RW; ++r) acc[r][j] += xv[r][i].x * b.x + xv[r][i].y * b.y + xv[r][i].z * b.z + xv[r][i].w * b.w;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int j = 0; j < OB; ++j) acc[r][j] = dcv_wave_sum(acc[r][j]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (n0 + r < N)
#pragma unroll
                    for (int j = 0; j < OB; ++j)
                        if (o0 + j < O) y[(size_t)(n0 + r) * O + o0 + j] = dcv_act_apply(acc[r][j] + (bias ? bias[o0 + j] : 0.f), act, slope);
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N, int K, int O) {
    const int KV = K / VEC;
    const size_t total = (size_t)N * KV;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / KV), k = (int)(i % KV);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        for (int o = 0; o < O; ++o) {
            const float g = dy[(size_t)n * O + o];
            if constexpr (VEC == 4) {
                const float4 b = reinterpret_cast<const float4*>(w + (size_t)o * K)[k];
                acc[0] += g * b.x; acc[1] += g * b.y; acc[2] += g * b.z; acc[3] += g * b.w;
            } else acc[0] += g * w[(size_t)o * K + k];
        }
        if constexpr (VEC == 4) { float4 o4; o4.x = acc[0]; o4.y = acc[1]; o4.z = acc[2]; o4.w = acc[3]; reinterpret_cast<float4*>(dx + (size_t)n * K)[k] = o4; }
        else dx[(size_t)n * K + k] = acc[0];
    }
}

#define LIN_WG_CHUNK 32          // samples per partial slab
// grid (k-vector tiles, O, chunks): partial[chunk][o][k] = sum_{n in chunk} dy[n][o] * x[n][k]
template <int VEC>
__global__ __launch_bounds__(256) void linear_wgrad_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, int N, int K, int O) {
    const int KV = K / VEC;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= KV) return;
    const int o = blockIdx.y, chunk = blockIdx.z;
    const int n0 = chunk * LIN_WG_CHUNK;
    int n1 = n0 + LIN_WG_CHUNK; if (n1 > N) n1 = N;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll 8
    for (int n = n0; n < n1; ++n) {                    // unrolled: 8 rows of x in flight per thread
        const float g = dy[(size_t)n * O + o];
        if constexpr (VEC == 4) {
            const float4 a = reinterpret_cast<const float4*>(x + (size_t)n * K)[k];
            acc[0] += g * a.x; acc[1] += g * a.y; acc[2] += g * a.z; acc[3] += g * a.w;
        } else acc[0] += g * x[(size_t)n * K + k];
    }
    float* dst = partial + ((size_t)chunk * O + o) * K;
    if constexpr (VEC == 4) { float4 o4; o4.x = acc[0]; o4.y = acc[1]; o4.z = acc[2]; o4.w = acc[3]; reinterpret_cast<float4*>(dst)[k] = o4; }
    else dst[k] = acc[0];
}

__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int chunks, size_t ok) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ok; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < chunks; ++c) s += partial[(size_t)c * ok + i];
        dw[i] = s;
    }
}

// db[o] = sum_n dy[n][o]: one wave per output feature, ordered shuffle reduction
__global__ __launch_bounds__(64) void linear_bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int N, int O) {
    const int o = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int n = lane; n < N; n += 64) acc += dy[(size_t)n * O + o];
    acc = dcv_wave_sum(acc);
    if (lane == 0) db[o] = acc;
}

// ---------------------------------------------------------------- cross entropy
// ---- fused backward of y = act(x W^T + b): the activation derivative is applied on the fly by every consumer of dz (no dz tensor, no separate pass),
// the bias gradient rides in the weight-gradient slabs
// Block (x = sample, y = tile of 32 k-vectors): 8 segments of the output features x 32 k-vectors; a thread sums every eighth output feature, the segments meet in
// LDS in a fixed order.  (Round 3: one thread per (sample, k-vector) looping over ALL outputs was a chain of O dependent iterations on N K / 4 threads — 72 us
// for the 1000-class head of resnet18_style at bs 64, on 32 blocks.)
template <int VEC>
__global__ __launch_bounds__(256) void linear_dgrad_act_kernel(const float* __restrict__ dy, const float* __restrict__ y, int act, float slope, const float* __restrict__ w,
                                                                float* __restrict__ dx, int N, int K, int O) {
    __shared__ float red[8][32][VEC];
    const int KV = K / VEC;
    const int kl = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int k = blockIdx.y * 32 + kl, n = blockIdx.x;
    const bool live = k < KV;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (live) {
#pragma unroll 4
        for (int o = seg; o < O; o += 8) {
            const float g = dy[(size_t)n * O + o] * (y ? dcv_act_grad_from_out(y[(size_t)n * O + o], act, slope) : 1.f);
            if constexpr (VEC == 4) {
                const float4 b = reinterpret_cast<const float4*>(w + (size_t)o * K)[k];
                acc[0] += g * b.x; acc[1] += g * b.y; acc[2] += g * b.z; acc[3] += g * b.w;
            } else acc[0] += g * w[(size_t)o * K + k];
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) red[seg][kl][v] = acc[v];
    __syncthreads();
    if (seg == 0 && live) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float s = red[0][kl][v];
#pragma unroll
            for (int q = 1; q < 8; ++q) s += red[q][kl][v];
            acc[v] = s;
        }
        if constexpr (VEC == 4) { float4 o4; o4.x = acc[0]; o4.y = acc[1]; o4.z = acc[2]; o4.w = acc[3]; reinterpret_cast<float4*>(dx + (size_t)n * K)[k] = o4; }
        else dx[(size_t)n * K + k] = acc[0];
    }
}

// grid (k-vector tiles, O, chunks): partial[chunk][o][k] = sum_{n in chunk} dz[n][o] * x[n][k];  partial_b[chunk][o] = sum_{n in chunk} dz[n][o]
template <int VEC>
__global__ __launch_bounds__(256) void linear_wgrad_act_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y, int act, float slope,
                                                                        float* __restrict__ partial, float* __restrict__ partial_b, int N, int K, int O) {
    const int KV = K / VEC;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= KV) return;
    const int o = blockIdx.y, chunk = blockIdx.z;
    const int n0 = chunk * LIN_WG_CHUNK;
    int n1 = n0 + LIN_WG_CHUNK; if (n1 > N) n1 = N;
    float acc[VEC], gsum = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll 8
    for (int n = n0; n < n1; ++n) {
        const float g = dy[(size_t)n * O + o] * (y ? dcv_act_grad_from_out(y[(size_t)n * O + o], act, slope) : 1.f);
        gsum += g;
        if constexpr (VEC == 4) {
            const float4 a = reinterpret_cast<const float4*>(x + (size_t)n * K)[k];
            acc[0] += g * a.x; acc[1] += g * a.y; acc[2] += g * a.z; acc[3] += g * a.w;
        } else acc[0] += g * x[(size_t)n * K + k];
    }
    float* dst = partial + ((size_t)chunk * O + o) * K;
    if constexpr (VEC == 4) { float4 o4; o4.x = acc[0]; o4.y = acc[1]; o4.z = acc[2]; o4.w = acc[3]; reinterpret_cast<float4*>(dst)[k] = o4; }
    else dst[k] = acc[0];
    if (k == 0 && partial_b) partial_b[(size_t)chunk * O + o] = gsum;
}

// Narrow heads (O <= 16: the 10-class classifier): one BLOCK per sample row, its four waves split K.  Every lane issues its x vector and the OB weight
// vectors of the same k together (one global round trip instead of K / 256 dependent ones per wave), the waves' sums meet in LDS in a fixed order.
template <int OB>
__global__ __launch_bounds__(256) void linear_fwd_row_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ y, int K, int O, int act, float slope) {
    __shared__ float part[4][OB];
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* __restrict__ xr = reinterpret_cast<const float4*>(x + (size_t)n * K);
    const int KV = K / 4;
    for (int o0 = 0; o0 < O; o0 += OB) {
        float acc[OB];
#pragma unroll
        for (int j = 0; j < OB; ++j) acc[j] = 0.f;
#pragma unroll 2
        for (int k = threadIdx.x; k < KV; k += 256) {
            const float4 a = xr[k];
#pragma unroll
            for (int j = 0; j < OB; ++j) {
                const int o = (o0 + j < O) ? o0 + j : O - 1;
                const float4 b = reinterpret_cast<const float4*>(w + (size_t)o * K)[k];
                acc[j] += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            }
        }
#pragma unroll
        for (int j = 0; j < OB; ++j) acc[j] = dcv_wave_sum(acc[j]);
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < OB; ++j) part[wave][j] = acc[j];
        }
        __syncthreads();
        if (threadIdx.x < OB && o0 + (int)threadIdx.x < O) {
            const int j = threadIdx.x;
            const float v = (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]);
            y[(size_t)n * O + o0 + j] = dcv_act_apply(v + (bias ? bias[o0 + j] : 0.f), act, slope);
        }
    }
}

// Small batches of few classes (N <= 1024, C <= 16: the stock classifier's 512 x 10 logits): the whole loss in ONE launch of one block, a thread per
// sample (its logits in registers: one global round trip), fixed-order block sum in fp64.
struct CePublish { const dcv_copy_jobs* mailbox; float* host_vals; int* host_seq; int slots; };
__global__ __launch_bounds__(1024) void ce_fwd_single_kernel(const float* __restrict__ z, const int64_t* __restrict__ target, float* __restrict__ lse_out,
                                                              float* __restrict__ loss, int32_t* __restrict__ correct, float* __restrict__ dz_unit, int N, int C, CePublish pub) {
    constexpr int CMAX = 16;
    __shared__ double wl[16];
    __shared__ int wc[16];
    const int n = threadIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double l = 0.0; int ok = 0;
    if (n < N) {
        const float* zr = z + (size_t)n * C;
        float v[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) v[c] = c < C ? zr[c] : -INFINITY;
        const int64_t t = target[n];
        float m = -INFINITY; int mi = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C && (v[c] > m || (v[c] != v[c]))) { m = v[c]; mi = c; }     // first maximum wins (NaN: last NaN, as the wave kernel)
        float sexp = 0.f, zt = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) { sexp += expf(v[c] - m); if ((int64_t)c == t) zt = v[c]; }
        const float lse = m + logf(sexp);
        lse_out[n] = lse;
        if (dz_unit != nullptr) {                                        // d loss / d logits for an upstream gradient of 1 (same expression as ce_bwd_kernel)
            const float g = 1.f / (float)N;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) if (c < C) dz_unit[(size_t)n * C + c] = (expf(v[c] - lse) - ((int64_t)c == t ? 1.f : 0.f)) * g;
        }
        l = (double)(lse - zt);
        ok = (mi == (int)t) ? 1 : 0;
    }
    l = dcv_wave_sum(l); ok = dcv_wave_sum(ok);
    if (lane == 0) { wl[wid] = l; wc[wid] = ok; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0; int k = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { s += wl[i]; k += wc[i]; }
        const float value = (float)(s / (double)N);
        loss[0] = value;
        if (correct) correct[0] = k;
        if (pub.host_seq != nullptr) {                                   // early read-back of a replayed step (dcv_publish_scalars without its launch)
            dcv_publish_value(pub.host_vals, pub.host_seq, pub.slots, value, pub.mailbox->seq);
        }
    }
}

#define CE_THREADS 256           // 4 waves per block, one sample per wave per iteration
// stage 1: per-block partial loss sum (double) and correct count
__global__ __launch_bounds__(CE_THREADS) void ce_fwd_partial_kernel(const float* __restrict__ z, const int64_t* __restrict__ target, float* __restrict__ lse_out,
                                                                    double* __restrict__ part_loss, int* __restrict__ part_correct, int N, int C) {
    __shared__ double wl[CE_THREADS / 64];
    __shared__ int wc[CE_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = CE_THREADS / 64;
    double lsum = 0.0; int correct = 0;
    for (int n = blockIdx.x * nw + wid; n < N; n += gridDim.x * nw) {
        const float* zr = z + (size_t)n * C;
        float m = -INFINITY; int mi = 0x7fffffff;
        for (int c = lane; c < C; c += 64) { const float v = zr[c]; if (v > m || (v != v)) { m = v; mi = c; } }
        // wave arg-max with "lowest index wins" tie-break
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float om = __shfl_xor(m, off, 64); const int oi = __shfl_xor(mi, off, 64);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(zr[c] - m);
        s = dcv_wave_sum(s);
        const float lse = m + logf(s);
        if (lane == 0) {
            const int64_t t = target[n];
            lse_out[n] = lse;
            const float zt = (t >= 0 && t < C) ? zr[t] : 0.f;
            lsum += (double)(lse - zt);
            correct += (mi == (int)t) ? 1 : 0;
        }
    }
    if (lane == 0) { wl[wid] = lsum; wc[wid] = correct; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0; int k = 0;
        for (int i = 0; i < nw; ++i) { s += wl[i]; k += wc[i]; }
        part_loss[blockIdx.x] = s; part_correct[blockIdx.x] = k;
    }
}

__global__ __launch_bounds__(64) void ce_fwd_final_kernel(const double* __restrict__ part_loss, const int* __restrict__ part_correct, int nb, int N, float* __restrict__ loss, int32_t* __restrict__ correct) {
    // one wave: lanes stride over the block partials (a single thread walking them was a chain of nb dependent loads), fixed-order butterfly
    const int lane = threadIdx.x;
    double s = 0.0; int k = 0;
    for (int i = lane; i < nb; i += 64) { s += part_loss[i]; k += part_correct[i]; }
    s = dcv_wave_sum(s); k = dcv_wave_sum(k);
    if (lane == 0) {
        loss[0] = (float)(s / (double)N);
        if (correct) correct[0] = k;
    }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ target, const float* __restrict__ lse,
                                                      const float* __restrict__ dloss, float* __restrict__ dz, int N, int C) {
    const float g = dloss[0] / (float)N;
    const size_t total = (size_t)N * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / C), c = (int)(i % C);
        const float p = expf(z[i] - lse[n]);
        dz[i] = (p - ((int64_t)c == target[n] ? 1.f : 0.f)) * g;
    }
}

#define CE_MAX_BLOCKS 256
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int dcv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int n, int K, int O, int act, float slope, void* stream) {
    DCV_CHECK_ARG(x && w && y && n > 0 && K > 0 && O > 0 && act >= DCV_ACT_NONE && act <= DCV_ACT_TANH);
    hipStream_t st = dcv_stream(stream);
    const dim3 grid((unsigned)dcv_ceil_div(n, 4), (unsigned)dcv_ceil_div(O, LIN_OCHUNK));
    const bool v4 = K % 4 == 0 && aligned16(x) && aligned16(w);
    if (v4 && O <= 16 && K >= 512) {
        if (O <= 8) hipLaunchKernelGGL((linear_fwd_row_kernel<8>), dim3((unsigned)n), dim3(256), 0, st, x, w, bias, y, K, O, act, slope);
        else if (O <= 10) hipLaunchKernelGGL((linear_fwd_row_kernel<10>), dim3((unsigned)n), dim3(256), 0, st, x, w, bias, y, K, O, act, slope);
        else hipLaunchKernelGGL((linear_fwd_row_kernel<16>), dim3((unsigned)n), dim3(256), 0, st, x, w, bias, y, K, O, act, slope);
    }
    else if (v4 && O > 16 && n >= 64 && K % 16 == 0 && K >= 64 && mfma_head()) {          // wide heads: the matrix cores (DCV_LINEAR_MFMA=0: the kernels below)
        hipLaunchKernelGGL(linear_fwd_mfma_kernel, dim3((unsigned)dcv_ceil_div(n, 16), (unsigned)dcv_ceil_div(O, 16)), dim3(256), 0, st, x, w, bias, y, n, K, O, act, slope);
    }
    else if (v4 && O > 16 && n >= 256 && K > 512 && K <= 4 * 64 * 8) {      // a wide head of long rows at a large batch: four sample rows per wave (27 -> 16 us on
                                                                            // 512 x 2048 -> 100; short rows — 512 -> 1000 — are bound by the per-wave reductions and stores, not by
                                                                            // the weight traffic: 77 us on the kernel below, 83 / 137 us with 4 / 16 rows per wave)
        const dim3 rgrid((unsigned)dcv_ceil_div(n, 16), (unsigned)dcv_ceil_div(O, 8));
        hipLaunchKernelGGL((linear_fwd_rows_kernel<4, 8>), rgrid, dim3(256), 0, st, x, w, bias, y, n, K, O, act, slope);
    }
    else if (v4 && O <= 10 && O > 8) hipLaunchKernelGGL((linear_fwd_kernel<4, 10>), grid, dim3(256), 0, st, x, w, bias, y, n, K, O, act, slope);
    else if (v4) hipLaunchKernelGGL((linear_fwd_kernel<4, 8>), grid, dim3(256), 0, st, x, w, bias, y, n, K, O, act, slope);
    else hipLaunchKernelGGL((linear_fwd_kernel<1, 8>), grid, dim3(256), 0, st, x, w, bias, y, n, K, O, act, slope);
    DCV_RETURN_LAUNCH();
}

int dcv_linear_dgrad(const float* dy, const float* w, float* dx, int n, int K, int O, void* stream) {
    DCV_CHECK_ARG(dy && w && dx && n > 0 && K > 0 && O > 0);
    hipStream_t st = dcv_stream(stream);
    if (K % 4 == 0 && aligned16(w) && aligned16(dx)) hipLaunchKernelGGL((linear_dgrad_kernel<4>), dim3(dcv_stream_grid((size_t)n * (K / 4), 256)), dim3(256), 0, st, dy, w, dx, n, K, O);
    else hipLaunchKernelGGL((linear_dgrad_kernel<1>), dim3(dcv_stream_grid((size_t)n * K, 256)), dim3(256), 0, st, dy, w, dx, n, K, O);
    DCV_RETURN_LAUNCH();
}

size_t dcv_linear_wgrad_workspace_bytes(int n, int K, int O) {
    if (n <= 0 || K <= 0 || O <= 0) return 0;
    return dcv_align_up(dcv_ceil_div(n, LIN_WG_CHUNK) * (size_t)O * K * sizeof(float), 256);
}

int dcv_linear_wgrad(const float* x, const float* dy, float* dw, float* db, int n, int K, int O, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(x && dy && dw && n > 0 && K > 0 && O > 0 && O <= 65535);
    if (!ws || ws_bytes < dcv_linear_wgrad_workspace_bytes(n, K, O)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    const int chunks = (int)dcv_ceil_div(n, LIN_WG_CHUNK);
    DCV_CHECK_ARG(chunks <= 65535);
    float* partial = reinterpret_cast<float*>(ws);
    if (K % 4 == 0 && aligned16(x)) {
        dim3 grid((unsigned)dcv_ceil_div(K / 4, 256), (unsigned)O, (unsigned)chunks);
        hipLaunchKernelGGL((linear_wgrad_partial_kernel<4>), grid, dim3(256), 0, st, x, dy, partial, n, K, O);
    } else {
        dim3 grid((unsigned)dcv_ceil_div(K, 256), (unsigned)O, (unsigned)chunks);
        hipLaunchKernelGGL((linear_wgrad_partial_kernel<1>), grid, dim3(256), 0, st, x, dy, partial, n, K, O);
    }
    const size_t ok = (size_t)O * K;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3(dcv_stream_grid(ok, 256)), dim3(256), 0, st, partial, dw, chunks, ok);
    if (db) hipLaunchKernelGGL(linear_bias_grad_kernel, dim3((unsigned)O), dim3(64), 0, st, dy, db, n, O);
    DCV_RETURN_LAUNCH();
}

size_t dcv_cross_entropy_workspace_bytes(int n) {
    (void)n;
    return dcv_align_up(CE_MAX_BLOCKS * (sizeof(double) + sizeof(int)), 256);
}

int dcv_cross_entropy_fwd(const float* logits, const int64_t* target, float* loss, float* lse, int32_t* correct, int n, int classes,
                          void* ws, size_t ws_bytes, void* stream) {
    return dcv_cross_entropy_fwd_grad(logits, target, loss, lse, correct, nullptr, nullptr, n, classes, ws, ws_bytes, stream);
}

int dcv_cross_entropy_fwd_grad(const float* logits, const int64_t* target, float* loss, float* lse, int32_t* correct, float* dlogits_unit, int* produced,
                               int n, int classes, void* ws, size_t ws_bytes, void* stream) {
    return dcv_cross_entropy_fwd_grad_publish(logits, target, loss, lse, correct, dlogits_unit, produced, n, classes, ws, ws_bytes, nullptr, nullptr, nullptr, 0, nullptr, stream);
}

int dcv_cross_entropy_fwd_grad_publish(const float* logits, const int64_t* target, float* loss, float* lse, int32_t* correct, float* dlogits_unit, int* produced,
                                       int n, int classes, void* ws, size_t ws_bytes, const dcv_copy_jobs* mailbox, float* host_vals, int* host_seq, int slots,
                                       int* published, void* stream) {
    DCV_CHECK_ARG(logits && target && loss && lse && n > 0 && classes > 0 && (dlogits_unit == nullptr || produced != nullptr));
    DCV_CHECK_ARG(host_seq == nullptr || (mailbox && host_vals && published && slots > 0 && slots <= DCV_MAX_COPY_JOBS));
    if (produced) *produced = 0;
    if (published) *published = 0;
    if (!ws || ws_bytes < dcv_cross_entropy_workspace_bytes(n)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    double* part_loss = reinterpret_cast<double*>(ws);
    int* part_correct = reinterpret_cast<int*>(part_loss + CE_MAX_BLOCKS);
    if (n <= 1024 && classes <= 16) {
        const CePublish pub{mailbox, host_vals, host_seq, slots};
        hipLaunchKernelGGL(ce_fwd_single_kernel, dim3(1), dim3((unsigned)dcv_align_up(n, 64)), 0, st, logits, target, lse, loss, correct, dlogits_unit, n, classes, pub);
        if (produced && dlogits_unit) *produced = 1;
        if (published && host_seq) *published = 1;
        DCV_RETURN_LAUNCH();
    }
    unsigned nb = (unsigned)dcv_ceil_div(n, CE_THREADS / 64);
    if (nb > CE_MAX_BLOCKS) nb = CE_MAX_BLOCKS;
    hipLaunchKernelGGL(ce_fwd_partial_kernel, dim3(nb), dim3(CE_THREADS), 0, st, logits, target, lse, part_loss, part_correct, n, classes);
    hipLaunchKernelGGL(ce_fwd_final_kernel, dim3(1), dim3(64), 0, st, part_loss, part_correct, (int)nb, n, loss, correct);
    DCV_RETURN_LAUNCH();
}

int dcv_cross_entropy_bwd(const float* logits, const int64_t* target, const float* lse, const float* dloss, float* dlogits, int n, int classes, void* stream) {
    DCV_CHECK_ARG(logits && target && lse && dloss && dlogits && n > 0 && classes > 0);
    hipStream_t st = dcv_stream(stream);
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(dcv_stream_grid((size_t)n * classes, 256)), dim3(256), 0, st, logits, target, lse, dloss, dlogits, n, classes);
    DCV_RETURN_LAUNCH();
}


// The backward of a wide head on the matrix cores (same shape of kernel as linear_fwd_mfma_kernel: a 16 x 16 result tile per block, the reduction axis dealt to the
// four waves in steps of 16, accumulators summed through LDS in wave order).  dz = dy act'(y) is formed while the fragments are loaded.
//   data gradient    dx[n][k]  = sum over o of dz[n][o] w[o][k]:  A = dz rows (float4 along o), B = w (four rows o of 16 consecutive k)
//   weight gradient  dw[o][k]  = sum over n of dz[n][o] x[n][k]:  A = dz columns (four samples of 16 consecutive o), B = x (four samples of 16 consecutive k);
//                    db[o] = sum over n of dz[n][o] by the blocks of k tile 0, from the same fragments.  The whole batch in one block: no partial slabs.
// The row-per-block kernels above took 41 + 48 us (+ 2 MB partial slabs per 32 samples for the step's slab sum) on 512 x 512 -> 1000.
__global__ __launch_bounds__(256) void linear_dgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ y, int act, float slope, const float* __restrict__ w,
                                                                 float* __restrict__ dx, int N, int K, int O) {
    __shared__ lin_f32x4 red[3][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16, k0 = blockIdx.y * 16;
    const int row = n0 + li < N ? n0 + li : N - 1, k = k0 + li < K ? k0 + li : K - 1;
    const float* dyr = dy + (size_t)row * O; const float* yr = y ? y + (size_t)row * O : nullptr;
    lin_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int steps = (O + 15) >> 4;
#pragma unroll 2
    for (int s = wave; s < steps; s += 4) {
        const int o = 16 * s + 4 * lg;                               // this lane's four features (O % 4 == 0: all inside or all outside)
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (o < O) {
            a = *reinterpret_cast<const float4*>(dyr + o);
            if (yr) {
                const float4 yv = *reinterpret_cast<const float4*>(yr + o);
                a.x *= dcv_act_grad_from_out(yv.x, act, slope); a.y *= dcv_act_grad_from_out(yv.y, act, slope);
                a.z *= dcv_act_grad_from_out(yv.z, act, slope); a.w *= dcv_act_grad_from_out(yv.w, act, slope);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = w[(size_t)(o + j) * K + k];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], acc, 0, 0, 0);
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) { const lin_f32x4 v = red[q][lane]; acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3]; }
    if (k0 + li >= K) return;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int n = n0 + 4 * lg + reg;                             // acc[reg]: row n0 + 4 lg + reg, column k0 + li
        if (n < N) dx[(size_t)n * K + k0 + li] = acc[reg];
    }
}

__global__ __launch_bounds__(256) void linear_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y, int act, float slope,
                                                                 float* __restrict__ dw, float* __restrict__ db, int N, int K, int O) {
    __shared__ lin_f32x4 red[3][64];
    __shared__ float redb[4][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int o0 = blockIdx.x * 16, k0 = blockIdx.y * 16;
    const int o = o0 + li < O ? o0 + li : O - 1, k = k0 + li < K ? k0 + li : K - 1;
    lin_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int steps = (N + 15) >> 4;
#pragma unroll 2
    for (int s = wave; s < steps; s += 4) {
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 16 * s + 4 * lg + j;
            const bool in = n < N;
            const size_t at = (size_t)(in ? n : 0) * O + o;
            float g = in ? dy[at] : 0.f;
            if (y && in) g *= dcv_act_grad_from_out(y[at], act, slope);
            a[j] = g;
            b[j] = in ? x[(size_t)n * K + k] : 0.f;
            bsum += g;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    redb[wave][lane] = bsum;
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) { const lin_f32x4 v = red[q][lane]; acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3]; }
    if (db != nullptr && blockIdx.y == 0 && lg == 0 && o0 + li < O) {
        float sb = 0.f;                                              // feature o0 + li: the four K groups of the four waves, in order
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) sb += redb[q][16 * g + li];
        db[o0 + li] = sb;
    }
    if (dw == nullptr || k0 + li >= K) return;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int oo = o0 + 4 * lg + reg;                            // acc[reg]: row (feature) o0 + 4 lg + reg, column k0 + li
        if (oo < O) dw[(size_t)oo * K + k0 + li] = acc[reg];
    }
}

size_t dcv_linear_bwd_workspace_bytes(int n, int K, int O) {
    if (n <= 0 || K <= 0 || O <= 0) return 0;
    return dcv_align_up(dcv_ceil_div(n, LIN_WG_CHUNK) * ((size_t)O * K + O) * sizeof(float), 256);
}

static int linear_bwd_impl(const float* dy, const float* y, int act, float slope, const float* x, const float* w, float* dx, float* dw, float* db,
                           int n, int K, int O, void* ws, size_t ws_bytes, void* stream, dcv_slab_segment* defer, int* ndefer) {
    DCV_CHECK_ARG(dy && x && w && n > 0 && K > 0 && O > 0 && O <= 65535 && act >= DCV_ACT_NONE && act <= DCV_ACT_TANH && (act == DCV_ACT_NONE || y));
    hipStream_t st = dcv_stream(stream);
    const float* yy = act == DCV_ACT_NONE ? nullptr : y;
    const bool v4 = K % 4 == 0 && aligned16(x) && aligned16(w) && (!dx || aligned16(dx));
    if (mfma_head() && O > 16 && O % 4 == 0 && n >= 64 && K >= 64 && aligned16(dy) && (!yy || aligned16(yy)) && dcv_ceil_div(K, 16) <= 65535) {
        // a wide head: both gradients on the matrix cores, the weight gradient over the whole batch in one block (no partial slabs, nothing to defer)
        if (dx) hipLaunchKernelGGL(linear_dgrad_mfma_kernel, dim3((unsigned)dcv_ceil_div(n, 16), (unsigned)dcv_ceil_div(K, 16)), dim3(256), 0, st, dy, yy, act, slope, w, dx, n, K, O);
        if (dw || db) hipLaunchKernelGGL(linear_wgrad_mfma_kernel, dim3((unsigned)dcv_ceil_div(O, 16), (unsigned)dcv_ceil_div(K, 16)), dim3(256), 0, st, x, dy, yy, act, slope, dw, db, n, K, O);
        DCV_RETURN_LAUNCH();
    }
    if (dx) {
        DCV_CHECK_ARG(dcv_ceil_div(K, 32) <= 65535);
        if (v4) hipLaunchKernelGGL((linear_dgrad_act_kernel<4>), dim3((unsigned)n, (unsigned)dcv_ceil_div(K / 4, 32)), dim3(256), 0, st, dy, yy, act, slope, w, dx, n, K, O);
        else hipLaunchKernelGGL((linear_dgrad_act_kernel<1>), dim3((unsigned)n, (unsigned)dcv_ceil_div(K, 32)), dim3(256), 0, st, dy, yy, act, slope, w, dx, n, K, O);
    }
    if (dw || db) {
        if (!ws || ws_bytes < dcv_linear_bwd_workspace_bytes(n, K, O)) return DCV_ERR_WORKSPACE;
        const int chunks = (int)dcv_ceil_div(n, LIN_WG_CHUNK);
        DCV_CHECK_ARG(chunks <= 65535);
        float* partial = reinterpret_cast<float*>(ws);
        float* partial_b = partial + (size_t)chunks * O * K;
        if (v4) {
            dim3 grid((unsigned)dcv_ceil_div(K / 4, 256), (unsigned)O, (unsigned)chunks);
            hipLaunchKernelGGL((linear_wgrad_act_partial_kernel<4>), grid, dim3(256), 0, st, x, dy, yy, act, slope, partial, partial_b, n, K, O);
        } else {
            dim3 grid((unsigned)dcv_ceil_div(K, 256), (unsigned)O, (unsigned)chunks);
            hipLaunchKernelGGL((linear_wgrad_act_partial_kernel<1>), grid, dim3(256), 0, st, x, dy, yy, act, slope, partial, partial_b, n, K, O);
        }
        const size_t ok = (size_t)O * K;
        dcv_slab_segment seg[2]; int ns = 0;
        if (dw) seg[ns++] = dcv_slab_segment{partial, dw, ok, ok, chunks, 0, 0, 0};
        if (db) seg[ns++] = dcv_slab_segment{partial_b, db, (unsigned long long)O, (unsigned long long)O, chunks, 0, 0, 0};
        if (defer) { for (int i = 0; i < ns; ++i) defer[i] = seg[i]; *ndefer = ns; }
        else return dcv_launch_slab_reduce(ns, seg, st);
    }
    DCV_RETURN_LAUNCH();
}

int dcv_linear_bwd(const float* dy, const float* y, int act, float slope, const float* x, const float* w, float* dx, float* dw, float* db,
                   int n, int K, int O, void* ws, size_t ws_bytes, void* stream) {
    return linear_bwd_impl(dy, y, act, slope, x, w, dx, dw, db, n, K, O, ws, ws_bytes, stream, nullptr, nullptr);
}

int dcv_linear_bwd_deferred(const float* dy, const float* y, int act, float slope, const float* x, const float* w, float* dx, float* dw, float* db,
                            int n, int K, int O, void* ws, size_t ws_bytes, dcv_slab_segment* segments, int* n_segments, void* stream) {
    DCV_CHECK_ARG(segments && n_segments);
    *n_segments = 0;
    return linear_bwd_impl(dy, y, act, slope, x, w, dx, dw, db, n, K, O, ws, ws_bytes, stream, segments, n_segments);
}


}  // extern "C"
