// norm.hip — BatchNorm (train / eval / backward) and GroupNorm (forward / backward) for NHWC fp32 activations.
//
// Roofline: HBM.  Algorithmic bytes: BN train fwd = 3 * 4 B * rows * C (two reads + one write),
// BN bwd = 5 * 4 B * rows * C, GN fwd = 3x, GN bwd = 5x.  All loads/stores are 16-B vectors when C % 4 == 0.
//
// Replaces torch.nn.BatchNorm2d (reference src/deepcv/meta/nn.py:418,487-491,501) and torch.nn.GroupNorm (nn.py:494-495,501).
#include "colreduce.h"
#include <stdlib.h>

namespace {

#define FIN_THREADS 256           // threads per channel of the finalize kernels

// ---------------------------------------------------------------- load functors for dcv_colreduce_block
struct SumSqLoader {            // q0 = x, q1 = x^2
    const float* x;
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int, float (&q)[2][VEC]) const {
        if constexpr (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4*>(x + idx);
            q[0][0] = v.x; q[0][1] = v.y; q[0][2] = v.z; q[0][3] = v.w;
        } else {
            q[0][0] = x[idx];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) q[1][i] = q[0][i] * q[0][i];
    }
};

struct BnBwdLoader {            // q0 = dy, q1 = dy * xhat   with per-channel mean / rstd
    const float* dy; const float* x; const float* mean; const float* rstd;
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int c, float (&q)[2][VEC]) const {
        float xv[VEC];
        if constexpr (VEC == 4) {
            const float4 d = *reinterpret_cast<const float4*>(dy + idx);
            const float4 v = *reinterpret_cast<const float4*>(x + idx);
            q[0][0] = d.x; q[0][1] = d.y; q[0][2] = d.z; q[0][3] = d.w;
            xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
        } else {
            q[0][0] = dy[idx]; xv[0] = x[idx];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) q[1][i] = q[0][i] * ((xv[i] - mean[c + i]) * rstd[c + i]);
    }
};

struct GnBwdLoader {            // q0 = dy, q1 = dy * xhat   with the sample's per-group mean / rstd
    const float* dy; const float* x; const float* mean; const float* rstd; int cpg;   // mean/rstd already offset to sample n
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int c, float (&q)[2][VEC]) const {
        float xv[VEC];
        if constexpr (VEC == 4) {
            const float4 d = *reinterpret_cast<const float4*>(dy + idx);
            const float4 v = *reinterpret_cast<const float4*>(x + idx);
            q[0][0] = d.x; q[0][1] = d.y; q[0][2] = d.z; q[0][3] = d.w;
            xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
        } else {
            q[0][0] = dy[idx]; xv[0] = x[idx];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) { const int g = (c + i) / cpg; q[1][i] = q[0][i] * ((xv[i] - mean[g]) * rstd[g]); }
    }
};

// ---------------------------------------------------------------- stage 1: per-block partial sums -> ws [nb][2][C] (double)
template <int VEC, typename Loader>
__global__ __launch_bounds__(DCV_CR_THREADS) void colreduce2_kernel(Loader ld, size_t rows, int C, double* __restrict__ partial) {
    __shared__ float scratch[2 * DCV_CR_THREADS * VEC];
    const size_t rpb = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * rpb;
    size_t r1 = r0 + rpb; if (r1 > rows) r1 = rows;
    dcv_colreduce_block<2, VEC>(ld, r0 < rows ? r0 : rows, r1, C, scratch, partial + (size_t)blockIdx.x * 2 * C);
}

// ---------------------------------------------------------------- BN stage 2 (forward): statistics, running stats, scale/shift
__global__ void bn_finalize_fwd_kernel(const double* __restrict__ partial, int nb, size_t rows, int C,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                                       float* __restrict__ save_mean, float* __restrict__ save_rstd, float* __restrict__ scale_shift, int transposed = 0) {
    // one block (FIN_THREADS) per channel; threads stride over the per-block partials (short load chains), fixed-order merge (butterfly, then waves)
    __shared__ double fred[FIN_THREADS / 64];
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c == 0 && lane == 0 && nbt) nbt[0] += 1;
    double s = 0.0, ss = 0.0;
    if (transposed) {        // [2][C][nb]: the rows of one (quantity, channel) are contiguous -> coalesced 8-B reads (convolution-epilogue statistics)
        const double* p0 = partial + (size_t)c * nb; const double* p1 = partial + ((size_t)C + c) * nb;
        for (int b = lane; b < nb; b += FIN_THREADS) { s += p0[b]; ss += p1[b]; }
    } else {
        for (int b = lane; b < nb; b += FIN_THREADS) { s += partial[((size_t)b * 2 + 0) * C + c]; ss += partial[((size_t)b * 2 + 1) * C + c]; }
    }
    s = dcv_block_sum(s, fred); ss = dcv_block_sum(ss, fred);
    if (lane != 0) return;
    const double n = (double)rows;
    const double mean = s / n;
    double var = ss / n - mean * mean;               // fp64: no cancellation issue at these magnitudes
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_rstd[c] = rstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * rstd;
    scale_shift[c] = sc;
    scale_shift[C + c] = b - (float)mean * sc;
}

__global__ void bn_finalize_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ rm, const float* __restrict__ rv, float eps, float* __restrict__ scale_shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale_shift[c] = g * rstd;
    scale_shift[C + c] = b - rm[c] * g * rstd;
}

// ---------------------------------------------------------------- stage 3: y = x * scale[c] + shift[c]
template <int VEC>
__global__ __launch_bounds__(256) void affine_apply_kernel(const float* __restrict__ x, float* __restrict__ y, size_t numel_v, int C,
                                                            const float* __restrict__ scale_shift) {
    const int CV = C / VEC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel_v; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * VEC;
        if constexpr (VEC == 4) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            const float4 sc = *reinterpret_cast<const float4*>(scale_shift + c);
            const float4 sh = *reinterpret_cast<const float4*>(scale_shift + C + c);
            float4 o; o.x = v.x * sc.x + sh.x; o.y = v.y * sc.y + sh.y; o.z = v.z * sc.z + sh.z; o.w = v.w * sc.w + sh.w;
            reinterpret_cast<float4*>(y)[i] = o;
        } else {
            y[i] = x[i] * scale_shift[c] + scale_shift[C + c];
        }
    }
}

// ---------------------------------------------------------------- BN backward stage 2 / 3
// coef layout (float [3][C]): a = gamma*rstd, b = sum(dy)/rows, d = sum(dy*xhat)/rows   ->  dx = a * (dy - b - xhat * d)
__global__ void bn_finalize_bwd_kernel(const double* __restrict__ partial, int nb, size_t rows, int C, const float* __restrict__ gamma,
                                       const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double fred[FIN_THREADS / 64];
    const int c = blockIdx.x, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < nb; b += FIN_THREADS) { s1 += partial[((size_t)b * 2 + 0) * C + c]; s2 += partial[((size_t)b * 2 + 1) * C + c]; }
    s1 = dcv_block_sum(s1, fred); s2 = dcv_block_sum(s2, fred);
    if (lane != 0) return;
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    const float g = gamma ? gamma[c] : 1.f;
    coef[c] = g * rstd[c];
    coef[C + c] = (float)(s1 / (double)rows);
    coef[2 * C + c] = (float)(s2 / (double)rows);
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                                            size_t numel_v, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ coef) {
    const int CV = C / VEC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel_v; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * VEC;
        float d[VEC], v[VEC], o[VEC];
        if constexpr (VEC == 4) {
            const float4 d4 = reinterpret_cast<const float4*>(dy)[i]; const float4 v4 = reinterpret_cast<const float4*>(x)[i];
            d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w; v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
        } else { d[0] = dy[i]; v[0] = x[i]; }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float xhat = (v[k] - mean[c + k]) * rstd[c + k];
            o[k] = coef[c + k] * (d[k] - coef[C + c + k] - xhat * coef[2 * C + c + k]);
        }
        if constexpr (VEC == 4) { float4 o4; o4.x = o[0]; o4.y = o[1]; o4.z = o[2]; o4.w = o[3]; reinterpret_cast<float4*>(dx)[i] = o4; }
        else dx[i] = o[0];
    }
}

// ---------------------------------------------------------------- GroupNorm: one block per sample (its [HW][C] slab is re-read from L2 for pass 2)
template <int VEC>
__global__ __launch_bounds__(DCV_CR_THREADS) void gn_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C, int G,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                 float* __restrict__ save_mean, float* __restrict__ save_rstd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* chan = reinterpret_cast<double*>(smem);                      // [2][C]
    float* scale = reinterpret_cast<float*>(chan + 2 * (size_t)C);       // [C]
    float* shift = scale + C;                                            // [C]
    float* scratch = shift + C;                                          // [2 * 256 * VEC]
    const int n = blockIdx.x;
    const size_t base = (size_t)n * HW * C;
    SumSqLoader ld{x + base};
    dcv_colreduce_block<2, VEC>(ld, 0, (size_t)HW, C, scratch, chan);
    const int cpg = C / G;
    // per-group statistics, then per-channel scale/shift for this sample
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        double s = 0.0, ss = 0.0;
        for (int k = g * cpg; k < (g + 1) * cpg; ++k) { s += chan[k]; ss += chan[C + k]; }
        const double m = (double)cpg * (double)HW;
        const double mean = s / m;
        double var = ss / m - mean * mean; if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        if (c == g * cpg) { save_mean[(size_t)n * G + g] = (float)mean; save_rstd[(size_t)n * G + g] = rstd; }
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        scale[c] = gm * rstd;
        shift[c] = bt - (float)mean * gm * rstd;
    }
    __syncthreads();
    const int CV = C / VEC;
    const size_t nv = (size_t)HW * CV;
    for (size_t i = threadIdx.x; i < nv; i += blockDim.x) {
        const int c = (int)(i % CV) * VEC;
        if constexpr (VEC == 4) {
            const float4 v = reinterpret_cast<const float4*>(x + base)[i];
            float4 o; o.x = v.x * scale[c] + shift[c]; o.y = v.y * scale[c + 1] + shift[c + 1];
            o.z = v.z * scale[c + 2] + shift[c + 2]; o.w = v.w * scale[c + 3] + shift[c + 3];
            reinterpret_cast<float4*>(y + base)[i] = o;
        } else {
            y[base + i] = x[base + i] * scale[c] + shift[c];
        }
    }
}

// backward: dx = rstd_g * (gamma_c*dy - A_g/m - xhat * B_g/m),  A_g = sum_{c in g} gamma_c * S1_c,  B_g = sum gamma_c * S2_c,
// S1_c = sum_hw dy, S2_c = sum_hw dy*xhat;  per-sample S1/S2 are also written to `partial` [N][2][C] for the dgamma/dbeta reduce.
template <int VEC>
__global__ __launch_bounds__(DCV_CR_THREADS) void gn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                                                 int HW, int C, int G, const float* __restrict__ gamma,
                                                                 const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                                 double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* chan = reinterpret_cast<double*>(smem);                      // [2][C]  S1, S2
    float* ca = reinterpret_cast<float*>(chan + 2 * (size_t)C);          // [C] gamma*rstd
    float* cb = ca + C;                                                  // [C] rstd * A_g / m
    float* cd = cb + C;                                                  // [C] rstd * B_g / m
    float* scratch = cd + C;
    const int n = blockIdx.x;
    const size_t base = (size_t)n * HW * C;
    const int cpg = C / G;
    const float* mean = save_mean + (size_t)n * G;
    const float* rstd = save_rstd + (size_t)n * G;
    GnBwdLoader ld{dy + base, x + base, mean, rstd, cpg};
    dcv_colreduce_block<2, VEC>(ld, 0, (size_t)HW, C, scratch, chan);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        double A = 0.0, B = 0.0;
        for (int k = g * cpg; k < (g + 1) * cpg; ++k) { const double gm = gamma ? (double)gamma[k] : 1.0; A += gm * chan[k]; B += gm * chan[C + k]; }
        const double m = (double)cpg * (double)HW;
        const float gm = gamma ? gamma[c] : 1.f;
        ca[c] = gm * rstd[g];
        cb[c] = (float)(A / m) * rstd[g];
        cd[c] = (float)(B / m) * rstd[g];
        if (partial) { partial[((size_t)n * 2 + 0) * C + c] = chan[c]; partial[((size_t)n * 2 + 1) * C + c] = chan[C + c]; }
    }
    __syncthreads();
    const int CV = C / VEC;
    const size_t nv = (size_t)HW * CV;
    for (size_t i = threadIdx.x; i < nv; i += blockDim.x) {
        const int c = (int)(i % CV) * VEC;
        float d[VEC], v[VEC], o[VEC];
        if constexpr (VEC == 4) {
            const float4 d4 = reinterpret_cast<const float4*>(dy + base)[i]; const float4 v4 = reinterpret_cast<const float4*>(x + base)[i];
            d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w; v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
        } else { d[0] = dy[base + i]; v[0] = x[base + i]; }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int g = (c + k) / cpg;
            const float xhat = (v[k] - mean[g]) * rstd[g];
            o[k] = ca[c + k] * d[k] - cb[c + k] - xhat * cd[c + k];
        }
        if constexpr (VEC == 4) { float4 o4; o4.x = o[0]; o4.y = o[1]; o4.z = o[2]; o4.w = o[3]; reinterpret_cast<float4*>(dx + base)[i] = o4; }
        else dx[base + i] = o[0];
    }
}

// dgamma[c] = sum_n partial[n][1][c], dbeta[c] = sum_n partial[n][0][c]   (ordered, fp64)
__global__ void gn_param_grad_kernel(const double* __restrict__ partial, int N, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int n = lane; n < N; n += DCV_WAVE) { s1 += partial[((size_t)n * 2 + 0) * C + c]; s2 += partial[((size_t)n * 2 + 1) * C + c]; }
    s1 = dcv_wave_sum(s1); s2 = dcv_wave_sum(s2);
    if (lane != 0) return;
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
}

// ================================================================================================ fused normalisation chain
// The reference layer is Conv -> act -> BatchNorm [-> GroupNorm] (src/deepcv/meta/nn.py:553).  On the stock model every standalone kernel of that
// chain runs for ~5 us, which is the launch-to-launch floor rather than its data movement, so the chain is expressed in fewer, fatter launches:
//   forward : statistics partials + finalize (as the standalone BatchNorm), then ONE kernel applying the BN affine and the GroupNorm of the BN output,
//             which is never written to HBM (3 launches instead of 4, one 8 MB write + read less);
//   backward: (1) GroupNorm backward on the recomputed BN output fused with the BatchNorm reduction terms, (2) one finalize for ALL parameter gradients
//             of the chain, (3) BatchNorm backward apply * act'(x) -> dz (what the convolution's dgrad / wgrad consume) with the bias-gradient partials,
//             (4) their ordered sum  (4 launches instead of 7, two 8 MB round trips less).
// A single-launch variant in which the last block to finish (integer ticket + __threadfence) performs the finalize was measured on MI355X and rejected:
// the device-scope release of every block writes back its XCD's L2, which made each of those kernels 40-70 us instead of ~8.
struct AffineSumSqLoader {      // v = x * scale[c] + shift[c];  q0 = v, q1 = v^2      (scale / shift in LDS)
    const float* x; const float* sc; const float* sh;
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int c, float (&q)[2][VEC]) const {
        if constexpr (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4*>(x + idx);
            q[0][0] = v.x; q[0][1] = v.y; q[0][2] = v.z; q[0][3] = v.w;
        } else {
            q[0][0] = x[idx];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) { q[0][i] = q[0][i] * sc[c + i] + sh[c + i]; q[1][i] = q[0][i] * q[0][i]; }
    }
};

// forward (2): out = GroupNorm(x * bn_scale + bn_shift), one block per sample; the BN output only exists in registers.
#define NF_NR 4                   // rows of a sample a thread keeps in registers in the CACHED variants (HW <= NF_NR * 256 / (C / VEC))

__device__ __forceinline__ void nf_load_vec4(const float* p, float (&v)[4]) { const float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
__device__ __forceinline__ void nf_load_vec4(const float* p, float (&v)[1]) { v[0] = p[0]; }
__device__ __forceinline__ void nf_store_vec4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void nf_store_vec4(float* p, const float (&v)[1]) { p[0] = v[0]; }

// CACHED: the sample is small enough for every thread to keep its rows in registers between the statistics pass and the apply pass: each element is
// read from memory once.
template <int VEC, bool CACHED>
__global__ __launch_bounds__(DCV_CR_THREADS) void nf_bn_gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C, int G,
                                                                         const float* __restrict__ bn_scale_shift, const float* __restrict__ gamma,
                                                                         const float* __restrict__ beta, float eps,
                                                                         float* __restrict__ save_mean, float* __restrict__ save_rstd, int W, int pk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* chan = reinterpret_cast<double*>(smem);                      // [2][C]
    float* bsc = reinterpret_cast<float*>(chan + 2 * (size_t)C);         // [C] BN scale
    float* bsh = bsc + C;                                                // [C] BN shift
    float* scale = bsh + C;                                              // [C] combined scale
    float* shift = scale + C;                                            // [C] combined shift
    float* scratch = shift + C;                                          // [2 * 256 * VEC]
    const int n = blockIdx.x;
    const size_t base = (size_t)n * HW * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { bsc[c] = bn_scale_shift[c]; bsh[c] = bn_scale_shift[C + c]; }
    __syncthreads();
    const int CV = C / VEC;
    const int rpi = CACHED ? DCV_CR_THREADS / CV : 1, cv0 = (threadIdx.x % CV) * VEC, rsub = threadIdx.x / CV;
    const bool row_thread = CACHED && (int)threadIdx.x < rpi * CV;
    float xc[NF_NR][VEC];
    if constexpr (CACHED) {
        float acc[2][VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { acc[0][v] = 0.f; acc[1][v] = 0.f; }
        if (row_thread) {
#pragma unroll
            for (int k = 0; k < NF_NR; ++k) {
                const int r = rsub + k * rpi;
                if (r < HW) {
                    nf_load_vec4(x + base + (size_t)r * C + cv0, xc[k]);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) { const float u = xc[k][v] * bsc[cv0 + v] + bsh[cv0 + v]; acc[0][v] += u; acc[1][v] += u * u; }
                }
            }
        }
        dcv_colreduce_merge<2, VEC>(acc, C, scratch, chan);
    } else {
        AffineSumSqLoader ld{x + base, bsc, bsh};
        dcv_colreduce_block<2, VEC>(ld, 0, (size_t)HW, C, scratch, chan);
    }
    const int cpg = C / G;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        double s = 0.0, ss = 0.0;
        for (int k = g * cpg; k < (g + 1) * cpg; ++k) { s += chan[k]; ss += chan[C + k]; }
        const double m = (double)cpg * (double)HW;
        const double mean = s / m;
        double var = ss / m - mean * mean; if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        if (c == g * cpg) { save_mean[(size_t)n * G + g] = (float)mean; save_rstd[(size_t)n * G + g] = rstd; }
        const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        // out = gm * ((x * bsc + bsh) - mean) * rstd + bt
        scale[c] = bsc[c] * rstd * gm;
        shift[c] = (bsh[c] - (float)mean) * rstd * gm + bt;
    }
    __syncthreads();
    if (pk > 0) {
        // fused average pooling (pk x pk windows, stride pk): the mean of the affine is the affine of the mean; only the pooled tensor is written.
        // The sample's pixels were read a moment ago (L2 / vector cache hits); one thread per pooled (pixel, channel vector).
        const int H = HW / W, Wo = W / pk, HoWo = (H / pk) * Wo;
        const float inv = 1.f / (float)(pk * pk);
        float* yp = y + (size_t)n * HoWo * C;
        for (int i = threadIdx.x; i < HoWo * CV; i += blockDim.x) {
            const int po = i / CV, c = (i - po * CV) * VEC;
            const int ho = po / Wo, wo = po - ho * Wo;
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
            for (int kh = 0; kh < pk; ++kh)
                for (int kw = 0; kw < pk; ++kw) {
                    float xv[VEC];
                    nf_load_vec4(x + base + ((size_t)(ho * pk + kh) * W + wo * pk + kw) * C + c, xv);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] += xv[v] * scale[c + v] + shift[c + v];     // same per-element rounding as the unfused affine
                }
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = acc[v] * inv;
            nf_store_vec4(yp + (size_t)po * C + c, o);
        }
        return;
    }
    if constexpr (CACHED) {
        if (row_thread) {
#pragma unroll
            for (int k = 0; k < NF_NR; ++k) {
                const int r = rsub + k * rpi;
                if (r < HW) {
                    float o[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o[v] = xc[k][v] * scale[cv0 + v] + shift[cv0 + v];
                    nf_store_vec4(y + base + (size_t)r * C + cv0, o);
                }
            }
        }
        return;
    }
    const size_t nv = (size_t)HW * CV;
    for (size_t i = threadIdx.x; i < nv; i += blockDim.x) {
        const int c = (int)(i % CV) * VEC;
        if constexpr (VEC == 4) {
            const float4 v = reinterpret_cast<const float4*>(x + base)[i];
            float4 o; o.x = v.x * scale[c] + shift[c]; o.y = v.y * scale[c + 1] + shift[c + 1];
            o.z = v.z * scale[c + 2] + shift[c + 2]; o.w = v.w * scale[c + 3] + shift[c + 3];
            reinterpret_cast<float4*>(y + base)[i] = o;
        } else {
            y[base + i] = x[base + i] * scale[c] + shift[c];
        }
    }
}

// finalize of the fused backward: one wave per channel, lanes stride over the per-sample partials [4][C][N] (S1, S2, T1, T2), fixed-order butterfly
__global__ void nf_gn_bn_finalize_kernel(const double* __restrict__ partial, int N, size_t rows, int C, const float* __restrict__ bn_gamma,
                                         const float* __restrict__ bn_rstd, float* __restrict__ d_bn_gamma, float* __restrict__ d_bn_beta,
                                         float* __restrict__ d_gn_gamma, float* __restrict__ d_gn_beta, float* __restrict__ coef) {
    __shared__ double fred[FIN_THREADS / 64];
    const int c = blockIdx.x, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0, t1 = 0.0, t2 = 0.0;
    const double* p0 = partial + (size_t)c * N; const double* p1 = partial + ((size_t)C + c) * N;
    const double* p2 = partial + (2 * (size_t)C + c) * N; const double* p3 = partial + (3 * (size_t)C + c) * N;
    for (int n = lane; n < N; n += FIN_THREADS) { s1 += p0[n]; s2 += p1[n]; t1 += p2[n]; t2 += p3[n]; }
    s1 = dcv_block_sum(s1, fred); s2 = dcv_block_sum(s2, fred); t1 = dcv_block_sum(t1, fred); t2 = dcv_block_sum(t2, fred);
    if (lane != 0) return;
    if (d_gn_beta) d_gn_beta[c] = (float)s1;
    if (d_gn_gamma) d_gn_gamma[c] = (float)s2;
    if (d_bn_beta) d_bn_beta[c] = (float)t1;
    if (d_bn_gamma) d_bn_gamma[c] = (float)t2;
    const float g = bn_gamma ? bn_gamma[c] : 1.f;
    coef[c] = g * bn_rstd[c];
    coef[C + c] = (float)(t1 / (double)rows);
    coef[2 * C + c] = (float)(t2 / (double)rows);
}

__global__ __launch_bounds__(64) void nf_sum_partials_kernel(const double* __restrict__ partial, int nb, int C, float* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int b = lane; b < nb; b += DCV_WAVE) s += partial[(size_t)b * C + c];
    s = dcv_wave_sum(s);
    if (lane == 0) out[c] = (float)s;
}

// Gradient w.r.t. the chain's output at element `idx` (= pixel * C + c) of one sample.  With fused average pooling (pk > 0) `dout` is the gradient of the
// POOLED tensor: every pixel of a window receives 1 / pk^2 of it.
struct NfPoolMap {
    int C, W, Wo, pk; float inv;
    __device__ __forceinline__ size_t src(size_t idx, int c) const {
        const int pix = (int)(idx / C);
        const int h = pix / W, w = pix - h * W;
        return ((size_t)(h / pk) * Wo + w / pk) * C + c;
    }
};
template <int VEC>
__device__ __forceinline__ void nf_load_dout(const float* dout, const NfPoolMap& pm, size_t idx, int c, float (&d)[VEC]) {
    if (pm.pk > 0) {
        nf_load_vec4(dout + pm.src(idx, c), d);
#pragma unroll
        for (int v = 0; v < VEC; ++v) d[v] *= pm.inv;
    } else nf_load_vec4(dout + idx, d);
}

struct NfGnBwdLoaderA {         // pass A: q0 = dout, q1 = dout * w,  w = (x * bsc + bsh - mean_g) * rstd_g   (mean / rstd expanded per CHANNEL in LDS)
    const float* dout; const float* x; const float* bsc; const float* bsh; const float* mean; const float* rstd; NfPoolMap pm;
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int c, float (&q)[2][VEC]) const {
        float xv[VEC];
        nf_load_dout<VEC>(dout, pm, idx, c, q[0]);
        nf_load_vec4(x + idx, xv);
#pragma unroll
        for (int i = 0; i < VEC; ++i) q[1][i] = q[0][i] * ((xv[i] * bsc[c + i] + bsh[c + i] - mean[c + i]) * rstd[c + i]);
    }
};

struct NfGnBwdLoaderB {         // pass B: dv = ca * dout - cb - w * cd  (stored to dv_out);  q0 = dv, q1 = dv * u,  u = (x - bn_mean) * bn_rstd
    const float* dout; const float* x; float* dv_out; const float* bsc; const float* bsh; const float* mean; const float* rstd;
    const float* ca; const float* cb; const float* cd; const float* bmean; const float* brstd; NfPoolMap pm;
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int c, float (&q)[2][VEC]) const {
        float xv[VEC], dv[VEC];
        nf_load_dout<VEC>(dout, pm, idx, c, dv);
        nf_load_vec4(x + idx, xv);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float w = (xv[i] * bsc[c + i] + bsh[c + i] - mean[c + i]) * rstd[c + i];
            dv[i] = ca[c + i] * dv[i] - cb[c + i] - w * cd[c + i];
            q[0][i] = dv[i];
            q[1][i] = dv[i] * ((xv[i] - bmean[c + i]) * brstd[c + i]);
        }
        if constexpr (VEC == 4) { float4 o; o.x = dv[0]; o.y = dv[1]; o.z = dv[2]; o.w = dv[3]; *reinterpret_cast<float4*>(dv_out + idx) = o; }
        else dv_out[idx] = dv[0];
    }
};

// backward (1), BN + GN: one block per sample.  GroupNorm backward on the recomputed BN output -> dv (gradient w.r.t. the BN output, written to `dv_out`),
// per-sample sums S1 = sum dout, S2 = sum dout*w (GN parameter gradients), T1 = sum dv, T2 = sum dv*u (BN terms) -> partial [4][C][N].
template <int VEC, bool CACHED>
__global__ __launch_bounds__(DCV_CR_THREADS) void nf_gn_bn_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x, float* __restrict__ dv_out,
                                                                       int HW, int C, int G, const float* __restrict__ bn_gamma, const float* __restrict__ bn_beta,
                                                                       const float* __restrict__ bn_mean, const float* __restrict__ bn_rstd,
                                                                       const float* __restrict__ gn_gamma, const float* __restrict__ gn_mean,
                                                                       const float* __restrict__ gn_rstd, double* __restrict__ partial, NfPoolMap pm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* chan = reinterpret_cast<double*>(smem);                      // [4][C]
    float* bsc = reinterpret_cast<float*>(chan + 4 * (size_t)C);         // [C] each:
    float* bsh = bsc + C; float* bm = bsh + C; float* br = bm + C; float* ca = br + C; float* cb = ca + C; float* cd = cb + C;
    float* gmc = cd + C; float* grc = gmc + C;                           // this sample's GroupNorm mean / rstd expanded per channel
    float* scratch = grc + C;                                            // [2 * 256 * VEC]
    const int n = blockIdx.x;
    const size_t base = (size_t)n * HW * C;
    const size_t dbase = pm.pk > 0 ? (size_t)n * (HW / (pm.pk * pm.pk)) * C : base;      // this sample inside dout (the pooled gradient when pooling is fused)
    const int cpg = C / G;
    const float* mean = gn_mean + (size_t)n * G;
    const float* rstd = gn_rstd + (size_t)n * G;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float g = bn_gamma ? bn_gamma[c] : 1.f, b = bn_beta ? bn_beta[c] : 0.f;
        const float sc = g * bn_rstd[c];
        bsc[c] = sc; bsh[c] = b - bn_mean[c] * sc; bm[c] = bn_mean[c]; br[c] = bn_rstd[c];
        gmc[c] = mean[c / cpg]; grc[c] = rstd[c / cpg];
    }
    __syncthreads();
    const int CV = C / VEC;
    const int rpi = CACHED ? DCV_CR_THREADS / CV : 1, cv0 = (threadIdx.x % CV) * VEC, rsub = threadIdx.x / CV;
    const bool row_thread = CACHED && (int)threadIdx.x < rpi * CV;
    float xc[NF_NR][VEC], dc[NF_NR][VEC];                               // CACHED: this thread's rows of x and dout, read from memory once
    if constexpr (CACHED) {
        float acc[2][VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { acc[0][v] = 0.f; acc[1][v] = 0.f; }
        if (row_thread) {
#pragma unroll
            for (int k = 0; k < NF_NR; ++k) {
                const int r = rsub + k * rpi;
                if (r < HW) {
                    nf_load_vec4(x + base + (size_t)r * C + cv0, xc[k]);
                    nf_load_dout<VEC>(dout + dbase, pm, (size_t)r * C + cv0, cv0, dc[k]);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const int c = cv0 + v;
                        const float w = (xc[k][v] * bsc[c] + bsh[c] - gmc[c]) * grc[c];
                        acc[0][v] += dc[k][v]; acc[1][v] += dc[k][v] * w;
                    }
                }
            }
        }
        dcv_colreduce_merge<2, VEC>(acc, C, scratch, chan);
    } else {
        NfGnBwdLoaderA la{dout + dbase, x + base, bsc, bsh, gmc, grc, pm};
        dcv_colreduce_block<2, VEC>(la, 0, (size_t)HW, C, scratch, chan);
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        double A = 0.0, B = 0.0;
        for (int k = g * cpg; k < (g + 1) * cpg; ++k) { const double gm = gn_gamma ? (double)gn_gamma[k] : 1.0; A += gm * chan[k]; B += gm * chan[C + k]; }
        const double m = (double)cpg * (double)HW;
        const float gm = gn_gamma ? gn_gamma[c] : 1.f;
        ca[c] = gm * rstd[g];
        cb[c] = (float)(A / m) * rstd[g];
        cd[c] = (float)(B / m) * rstd[g];
    }
    __syncthreads();
    if constexpr (CACHED) {
        float acc[2][VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { acc[0][v] = 0.f; acc[1][v] = 0.f; }
        if (row_thread) {
#pragma unroll
            for (int k = 0; k < NF_NR; ++k) {
                const int r = rsub + k * rpi;
                if (r < HW) {
                    float dvv[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const int c = cv0 + v;
                        const float w = (xc[k][v] * bsc[c] + bsh[c] - gmc[c]) * grc[c];
                        dvv[v] = ca[c] * dc[k][v] - cb[c] - w * cd[c];
                        acc[0][v] += dvv[v]; acc[1][v] += dvv[v] * ((xc[k][v] - bm[c]) * br[c]);
                    }
                    nf_store_vec4(dv_out + base + (size_t)r * C + cv0, dvv);
                }
            }
        }
        dcv_colreduce_merge<2, VEC>(acc, C, scratch, chan + 2 * (size_t)C);
    } else {
        NfGnBwdLoaderB lb{dout + dbase, x + base, dv_out + base, bsc, bsh, gmc, grc, ca, cb, cd, bm, br, pm};
        dcv_colreduce_block<2, VEC>(lb, 0, (size_t)HW, C, scratch, chan + 2 * (size_t)C);
    }
    for (int i = threadIdx.x; i < 4 * C; i += blockDim.x) partial[(size_t)i * gridDim.x + n] = chan[i];     // [4][C][N]: the finalize reads sample-contiguous
}

struct NfApplyActLoader {       // dz = act'(x) * a * (dv - b - u * d), stored to dz;  q0 = dz  (bias gradient)
    const float* dv; const float* x; float* dz; const float* mean; const float* rstd; const float* coef; int C; int act; float slope;
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int c, float (&q)[1][VEC]) const {
        float d[VEC], v[VEC];
        if constexpr (VEC == 4) {
            const float4 d4 = *reinterpret_cast<const float4*>(dv + idx); const float4 v4 = *reinterpret_cast<const float4*>(x + idx);
            d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w; v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
        } else { d[0] = dv[idx]; v[0] = x[idx]; }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float u = (v[k] - mean[c + k]) * rstd[c + k];
            q[0][k] = coef[c + k] * (d[k] - coef[C + c + k] - u * coef[2 * C + c + k]) * dcv_act_grad_from_out(v[k], act, slope);
        }
        if constexpr (VEC == 4) { float4 o4; o4.x = q[0][0]; o4.y = q[0][1]; o4.z = q[0][2]; o4.w = q[0][3]; *reinterpret_cast<float4*>(dz + idx) = o4; }
        else dz[idx] = q[0][0];
    }
};

// backward (3): BN backward apply fused with the activation derivative; per-block column sums of dz (bias gradient) -> partial [nb][C]
template <int VEC>
__global__ __launch_bounds__(DCV_CR_THREADS) void nf_bn_apply_act_bwd_kernel(NfApplyActLoader ld, size_t rows, int C, double* __restrict__ partial) {
    __shared__ float scratch[DCV_CR_THREADS * VEC];
    const size_t rpb = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * rpb;
    size_t r1 = r0 + rpb; if (r1 > rows) r1 = rows;
    dcv_colreduce_block<1, VEC>(ld, r0 < rows ? r0 : rows, r1, C, scratch, partial + (size_t)blockIdx.x * C);
}

// backward (3) without bias gradient (the convolution's weight-gradient kernel sums dz itself): plain streaming kernel
template <int VEC>
__global__ __launch_bounds__(256) void nf_bn_apply_act_bwd_ew_kernel(NfApplyActLoader ld, size_t numel_v, int C) {
    const int CV = C / VEC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel_v; i += (size_t)gridDim.x * blockDim.x) {
        float q[1][VEC];
        ld.template load<VEC>(i * VEC, (int)(i % CV) * VEC, q);
    }
}

// ---- single-pass backward of the chain (samples which fit the threads' registers, no bias gradient wanted).  The two-pass kernel above forms dv, the
// gradient w.r.t. the BatchNorm output, only to sum it: T1 = sum dv, T2 = sum dv * u.  With u = (x - bn_mean) * bn_rstd the GroupNorm input is affine in u
// (v = bn_gamma * u + bn_beta, w = (v - mean_g) * rstd_g = alpha * u + beta'), so every sum of the chain follows from four moments per (sample, channel):
// M0 = sum dout, M1 = sum dout * u, U1 = sum u, U2 = sum u^2 — one pass, one merge, no dv tensor (its 8 MB store was 3.7 of the kernel's 10.9 us).  The
// block then does the per-sample algebra itself: the same per-sample partials S1, S2, T1, T2 as the two-pass kernel (-> the same finalize), plus three
// coefficients per (sample, channel) with which the apply pass evaluates dz = act'(x) * (P * dout + (Q0 - a * d) * u + (R0 - a * b)) straight from dout
// and x (a, b, d: the finalize's per-channel BatchNorm-backward terms).
template <int VEC>
__global__ __launch_bounds__(DCV_CR_THREADS) void nf_moments_kernel(const float* __restrict__ dout, const float* __restrict__ x, int HW, int C, int G,
                                                                     const float* __restrict__ bn_gamma, const float* __restrict__ bn_beta,
                                                                     const float* __restrict__ bn_mean, const float* __restrict__ bn_rstd,
                                                                     const float* __restrict__ gn_gamma, const float* __restrict__ gn_mean, const float* __restrict__ gn_rstd,
                                                                     double* __restrict__ partial, float* __restrict__ pre, NfPoolMap pm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* chan = reinterpret_cast<double*>(smem);                      // [4][C]
    float* bm = reinterpret_cast<float*>(chan + 4 * (size_t)C);          // [C] BatchNorm mean, rstd
    float* br = bm + C;
    float* scratch = br + C;                                             // [4 * 256 * VEC]
    const int n = blockIdx.x;
    const size_t base = (size_t)n * HW * C;
    const size_t dbase = pm.pk > 0 ? (size_t)n * (HW / (pm.pk * pm.pk)) * C : base;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { bm[c] = bn_mean[c]; br[c] = bn_rstd[c]; }
    __syncthreads();
    const int CV = C / VEC;
    const int rpi = DCV_CR_THREADS / CV, cv0 = (threadIdx.x % CV) * VEC, rsub = threadIdx.x / CV;
    float acc[4][VEC];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[q][v] = 0.f;
    if ((int)threadIdx.x < rpi * CV) {
#pragma unroll
        for (int k = 0; k < NF_NR; ++k) {
            const int r = rsub + k * rpi;
            if (r < HW) {
                float xv[VEC], dv[VEC];
                nf_load_vec4(x + base + (size_t)r * C + cv0, xv);
                nf_load_dout<VEC>(dout + dbase, pm, (size_t)r * C + cv0, cv0, dv);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float u = (xv[v] - bm[cv0 + v]) * br[cv0 + v];
                    acc[0][v] += dv[v]; acc[1][v] += dv[v] * u; acc[2][v] += u; acc[3][v] += u * u;
                }
            }
        }
    }
    dcv_colreduce_merge<4, VEC>(acc, C, scratch, chan);                  // M0, M1, U1, U2 of this sample; ends with a barrier
    const int cpg = C / G;
    const double hw = (double)HW, m = (double)cpg * hw;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const double gr = (double)gn_rstd[(size_t)n * G + g], gmean = (double)gn_mean[(size_t)n * G + g];
        double A = 0.0, B = 0.0;                                          // group sums of gamma * S1, gamma * S2
        for (int k = g * cpg; k < (g + 1) * cpg; ++k) {
            const double gmk = gn_gamma ? (double)gn_gamma[k] : 1.0;
            const double al = (bn_gamma ? (double)bn_gamma[k] : 1.0) * gr, be = ((bn_beta ? (double)bn_beta[k] : 0.0) - gmean) * gr;
            A += gmk * chan[k]; B += gmk * (al * chan[C + k] + be * chan[k]);
        }
        const double gbc = bn_gamma ? (double)bn_gamma[c] : 1.0;
        const double al = gbc * gr, be = ((bn_beta ? (double)bn_beta[c] : 0.0) - gmean) * gr;
        const double ca = (gn_gamma ? (double)gn_gamma[c] : 1.0) * gr, cb = (A / m) * gr, cd = (B / m) * gr;
        const double m0 = chan[c], m1 = chan[C + c], u1 = chan[2 * C + c], u2 = chan[3 * C + c];
        const size_t N = gridDim.x;
        partial[(size_t)c * N + n] = m0;                                                         // S1 = sum dout
        partial[((size_t)C + c) * N + n] = al * m1 + be * m0;                                    // S2 = sum dout * w
        partial[(2 * (size_t)C + c) * N + n] = ca * m0 - cb * hw - cd * (al * u1 + be * hw);     // T1 = sum dv
        partial[(3 * (size_t)C + c) * N + n] = ca * m1 - cb * u1 - cd * (al * u2 + be * u1);     // T2 = sum dv * u
        const double a = gbc * (double)br[c];
        float* pf = pre + (size_t)n * 3 * C;
        pf[c] = (float)(a * ca); pf[C + c] = (float)(-a * al * cd); pf[2 * C + c] = (float)(-a * (cb + be * cd));
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void nf_moments_apply_kernel(const float* __restrict__ dout, const float* __restrict__ x, float* __restrict__ dz, int HW, int C,
                                                                const float* __restrict__ pre, const float* __restrict__ coef /* a, b, d [C] each */,
                                                                const float* __restrict__ bn_mean, const float* __restrict__ bn_rstd,
                                                                int act, float slope, NfPoolMap pm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cf = reinterpret_cast<float*>(smem);                          // [3][C]: P, Q, R of this sample, then BatchNorm mean, rstd [C] each
    float* bm = cf + 3 * C; float* br = bm + C;
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float* pf = pre + (size_t)n * 3 * C;
        const float a = coef[c], b = coef[C + c], d = coef[2 * C + c];
        cf[c] = pf[c]; cf[C + c] = pf[C + c] - a * d; cf[2 * C + c] = pf[2 * C + c] - a * b;
        bm[c] = bn_mean[c]; br[c] = bn_rstd[c];
    }
    __syncthreads();
    const size_t base = (size_t)n * HW * C;
    const size_t dbase = pm.pk > 0 ? (size_t)n * (HW / (pm.pk * pm.pk)) * C : base;
    const int CV = C / VEC, nv = HW * CV;
    for (int i = threadIdx.x; i < nv; i += blockDim.x) {
        const int c = (i % CV) * VEC;
        float xv[VEC], dv[VEC], o[VEC];
        nf_load_vec4(x + base + (size_t)i * VEC, xv);
        nf_load_dout<VEC>(dout + dbase, pm, (size_t)i * VEC, c, dv);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float u = (xv[v] - bm[c + v]) * br[c + v];
            o[v] = (cf[c + v] * dv[v] + cf[C + c + v] * u + cf[2 * C + c + v]) * dcv_act_grad_from_out(xv[v], act, slope);
        }
        nf_store_vec4(dz + base + (size_t)i * VEC, o);
    }
}

// ================================================================================================ split-phase BatchNorm (cross-GPU statistics)
// `use_sync_batch_norm` (reference ignite_training.py:386-388, torch.nn.SyncBatchNorm): every reduction over the batch is cut in two so that the caller
// can all-reduce the per-channel sums between the phases (collective C4 of SURVEY.md §2.3, 2 * C doubles).
//   forward : (1) sums[2][C] = sum x, sum x^2 of the local rows   ->  all-reduce  ->  (2) statistics over `total_rows`, running stats, y
//   backward: (1) sums[2][C] = sum dy, sum dy * xhat (local; also the local d_beta, d_gamma, which the gradient all-reduce averages like any parameter
//             gradient)                                            ->  all-reduce  ->  (2) dx with the GLOBAL means of those sums
__global__ void reduce_partials_kernel(const double* __restrict__ partial, int nb, int Q, int C, double* __restrict__ sums) {
    // one wave per (quantity, channel); lanes stride over the partial rows, fixed-order butterfly
    const int qc = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int b = lane; b < nb; b += DCV_WAVE) s += partial[(size_t)b * Q * C + qc];
    s = dcv_wave_sum(s);
    if (lane == 0) sums[qc] = s;
}

__global__ void bn_finalize_from_sums_kernel(const double* __restrict__ sums, double total_rows, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                                             float* __restrict__ save_mean, float* __restrict__ save_rstd, float* __restrict__ scale_shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) nbt[0] += 1;
    if (c >= C) return;
    const double n = total_rows;
    const double mean = sums[c] / n;
    double var = sums[C + c] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_rstd[c] = rstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * rstd;
    scale_shift[c] = sc;
    scale_shift[C + c] = b - (float)mean * sc;
}

__global__ void bn_coef_from_sums_kernel(const double* __restrict__ sums, double total_rows, int C, const float* __restrict__ gamma, const float* __restrict__ rstd,
                                         float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.f;
    coef[c] = g * rstd[c];
    coef[C + c] = (float)(sums[c] / total_rows);
    coef[2 * C + c] = (float)(sums[C + c] / total_rows);
}

inline size_t bn_ws_partial_bytes(size_t rows, int C) { return dcv_align_up((size_t)dcv_colreduce_blocks(rows, C) * 2 * C * sizeof(double), 256); }

// ------------------------------------------------------------------------------------------------ LayerNorm's elementwise affine
// torch.nn.LayerNorm over the last two ([H][W]: what the reference's factory builds, normalized_shape = input_shape[1:]) or three ([C][H][W]) dimensions of an
// [N][C][H][W] tensor stored NHWC: the statistics are GroupNorm's with one channel per group (resp. one group) and no affine — gn_fwd_kernel / gn_bwd_kernel
// above —, what is left is the per-ELEMENT affine y = xhat * w + b with w indexed [H][W] (wc_stride = 0: shared by the channels) or [C][H][W]
// (wc_stride = HW) while the activations are [HW][C].  Forward: in place over xhat.  Backward: a thread per (pixel, channel) element walks the samples in
// order (xhat recomputed from x and the saved statistics), leaves dxhat = dy * w for gn_bwd_kernel and its fp64 sums of dy * xhat and dy; a second kernel
// adds the channels of a pixel in order where the weight is shared (fixed summation order: the same bits on every run).
__global__ __launch_bounds__(256) void ln_affine_fwd_kernel(float* __restrict__ y, size_t total, int per_sample, int C, int wc_stride,
                                                            const float* __restrict__ w, const float* __restrict__ b) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int e = (int)(i % (size_t)per_sample), p = e / C, c = e - p * C, wi = c * wc_stride + p;
        y[i] = y[i] * w[wi] + (b ? b[wi] : 0.f);
    }
}

__global__ __launch_bounds__(256) void ln_affine_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int n, int per_sample, int C, int G, int wc_stride,
                                                            const float* __restrict__ w, float* __restrict__ dxhat, double* __restrict__ part) {
    const int e = (int)(blockIdx.x * 256 + threadIdx.x);
    if (e >= per_sample) return;
    const int p = e / C, c = e - p * C, g = G == 1 ? 0 : c;
    const float wv = w[c * wc_stride + p];
    double s_dyx = 0.0, s_dy = 0.0;
#pragma unroll 4
    for (int s = 0; s < n; ++s) {
        const size_t i = (size_t)s * per_sample + e;
        const float d = dy[i], xh = (x[i] - mean[(size_t)s * G + g]) * rstd[(size_t)s * G + g];
        s_dy += (double)d; s_dyx += (double)d * (double)xh;
        dxhat[i] = d * wv;
    }
    part[e] = s_dyx; part[(size_t)per_sample + e] = s_dy;
}

__global__ __launch_bounds__(256) void ln_affine_param_grad_kernel(const double* __restrict__ part, int per_sample, int C, int hw, int wc_stride,
                                                                   float* __restrict__ dw, float* __restrict__ db) {
    const int wi = (int)(blockIdx.x * 256 + threadIdx.x);
    if (wi >= (wc_stride ? per_sample : hw)) return;
    double a = 0.0, b = 0.0;
    if (wc_stride) { const int c = wi / hw, p = wi - c * hw; a = part[p * C + c]; b = part[(size_t)per_sample + p * C + c]; }
    else for (int c = 0; c < C; ++c) { a += part[wi * C + c]; b += part[(size_t)per_sample + wi * C + c]; }
    if (dw) dw[wi] = (float)a;
    if (db) db[wi] = (float)b;
}

// ------------------------------------------------------------------------------------------------ LocalResponseNorm
// torch.nn.LocalResponseNorm over the channels of a pixel (NHWC: the window is contiguous): scale_c = k + alpha / size * sum of x^2 over the channels
// [c - size / 2, c + (size - 1) / 2] inside the tensor, y_c = x_c * scale_c^-beta.  Backward: dx_c = dy_c scale_c^-beta - (2 alpha beta / size) x_c * sum over the
// channels c' whose window holds c of dy_c' x_c' scale_c'^(-beta - 1).  A thread per element; the window's values come from the row the neighbouring lanes just read (L1 / L2).
__device__ __forceinline__ float lrn_scale(const float* __restrict__ row, int c, int C, int size, float alpha_over_size, float k) {
    const int lo = max(c - size / 2, 0), hi = min(c + (size - 1) / 2, C - 1);
    float s = 0.f;
    for (int j = lo; j <= hi; ++j) s += row[j] * row[j];
    return k + alpha_over_size * s;
}

__global__ __launch_bounds__(256) void lrn_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int C, int size, float alpha_over_size, float beta, float k) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (size_t)C);
        const float* row = x + (i - c);
        y[i] = row[c] * powf(lrn_scale(row, c, C, size, alpha_over_size, k), -beta);
    }
}

__global__ __launch_bounds__(256) void lrn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, size_t total, int C, int size,
                                                      float alpha_over_size, float beta, float k) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (size_t)C);
        const float* row = x + (i - c);
        const float* drow = dy + (i - c);
        // c lies in the window of c' when c' - size / 2 <= c <= c' + (size - 1) / 2
        const int lo = max(c - (size - 1) / 2, 0), hi = min(c + size / 2, C - 1);
        float acc = 0.f;
        for (int j = lo; j <= hi; ++j) acc += drow[j] * row[j] * powf(lrn_scale(row, j, C, size, alpha_over_size, k), -beta - 1.f);
        dx[i] = drow[c] * powf(lrn_scale(row, c, C, size, alpha_over_size, k), -beta) - 2.f * alpha_over_size * beta * row[c] * acc;
    }
}

}  // namespace

extern "C" {

size_t dcv_batchnorm_workspace_bytes(size_t rows, int channels) {
    if (rows == 0 || channels <= 0) return 0;
    return bn_ws_partial_bytes(rows, channels) + dcv_align_up((size_t)3 * channels * sizeof(float), 256);
}

int dcv_batchnorm_fwd_train(const float* x, float* y, size_t rows, int C, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                            float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(x && y && save_mean && save_rstd && ws && rows > 0 && C > 0);
    if (ws_bytes < dcv_batchnorm_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    double* partial = reinterpret_cast<double*>(ws);
    float* scale_shift = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + bn_ws_partial_bytes(rows, C));
    const unsigned nb = dcv_colreduce_blocks(rows, C);
    SumSqLoader ld{x};
    if (C % 4 == 0) hipLaunchKernelGGL((colreduce2_kernel<4, SumSqLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    else hipLaunchKernelGGL((colreduce2_kernel<1, SumSqLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((unsigned)C), dim3(FIN_THREADS), 0, st, partial, (int)nb, rows, C, gamma, beta,
                       running_mean, running_var, nbt, momentum, eps, save_mean, save_rstd, scale_shift);
    if (C % 4 == 0) {
        const size_t nv = rows * (size_t)C / 4;
        hipLaunchKernelGGL((affine_apply_kernel<4>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, x, y, nv, C, scale_shift);
    } else {
        const size_t nv = rows * (size_t)C;
        hipLaunchKernelGGL((affine_apply_kernel<1>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, x, y, nv, C, scale_shift);
    }
    DCV_RETURN_LAUNCH();
}

int dcv_batchnorm_fwd_eval(const float* x, float* y, size_t rows, int C, const float* gamma, const float* beta,
                           const float* running_mean, const float* running_var, float eps, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(x && y && running_mean && running_var && ws && rows > 0 && C > 0);
    if (ws_bytes < dcv_batchnorm_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    float* scale_shift = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + bn_ws_partial_bytes(rows, C));
    hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3((unsigned)dcv_ceil_div(C, 64)), dim3(64), 0, st, C, gamma, beta, running_mean, running_var, eps, scale_shift);
    if (C % 4 == 0) {
        const size_t nv = rows * (size_t)C / 4;
        hipLaunchKernelGGL((affine_apply_kernel<4>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, x, y, nv, C, scale_shift);
    } else {
        const size_t nv = rows * (size_t)C;
        hipLaunchKernelGGL((affine_apply_kernel<1>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, x, y, nv, C, scale_shift);
    }
    DCV_RETURN_LAUNCH();
}

int dcv_batchnorm_bwd(const float* dy, const float* x, float* dx, size_t rows, int C, const float* gamma,
                      const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(dy && x && dx && save_mean && save_rstd && ws && rows > 0 && C > 0);
    if (ws_bytes < dcv_batchnorm_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    double* partial = reinterpret_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + bn_ws_partial_bytes(rows, C));
    const unsigned nb = dcv_colreduce_blocks(rows, C);
    BnBwdLoader ld{dy, x, save_mean, save_rstd};
    if (C % 4 == 0) hipLaunchKernelGGL((colreduce2_kernel<4, BnBwdLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    else hipLaunchKernelGGL((colreduce2_kernel<1, BnBwdLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((unsigned)C), dim3(FIN_THREADS), 0, st, partial, (int)nb, rows, C, gamma, save_rstd, dgamma, dbeta, coef);
    if (C % 4 == 0) {
        const size_t nv = rows * (size_t)C / 4;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<4>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, dy, x, dx, nv, C, save_mean, save_rstd, coef);
    } else {
        const size_t nv = rows * (size_t)C;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, dy, x, dx, nv, C, save_mean, save_rstd, coef);
    }
    DCV_RETURN_LAUNCH();
}

size_t dcv_groupnorm_workspace_bytes(int n, int channels) {
    if (n <= 0 || channels <= 0) return 0;
    return dcv_align_up((size_t)n * 2 * channels * sizeof(double), 256);
}

static size_t gn_smem_bytes(int C, int vec, int nfloat_arrays) {
    return 2 * (size_t)C * sizeof(double) + (size_t)nfloat_arrays * C * sizeof(float) + (size_t)2 * DCV_CR_THREADS * vec * sizeof(float);
}

int dcv_groupnorm_fwd(const float* x, float* y, int n, int hw, int C, int G, const float* gamma, const float* beta, float eps,
                      float* save_mean, float* save_rstd, void* stream) {
    DCV_CHECK_ARG(x && y && save_mean && save_rstd && n > 0 && hw > 0 && C > 0 && G > 0 && C % G == 0);
    hipStream_t st = dcv_stream(stream);
    const int vec = (C % 4 == 0) ? 4 : 1;
    const size_t smem = gn_smem_bytes(C, vec, 2);
    if (smem > 64 * 1024) return DCV_ERR_UNSUPPORTED;
    if (vec == 4) hipLaunchKernelGGL((gn_fwd_kernel<4>), dim3(n), dim3(DCV_CR_THREADS), smem, st, x, y, hw, C, G, gamma, beta, eps, save_mean, save_rstd);
    else hipLaunchKernelGGL((gn_fwd_kernel<1>), dim3(n), dim3(DCV_CR_THREADS), smem, st, x, y, hw, C, G, gamma, beta, eps, save_mean, save_rstd);
    DCV_RETURN_LAUNCH();
}

int dcv_groupnorm_bwd(const float* dy, const float* x, float* dx, int n, int hw, int C, int G, const float* gamma,
                      const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(dy && x && dx && save_mean && save_rstd && n > 0 && hw > 0 && C > 0 && G > 0 && C % G == 0);
    const bool need_param_grads = (dgamma != nullptr) || (dbeta != nullptr);
    if (need_param_grads && (!ws || ws_bytes < dcv_groupnorm_workspace_bytes(n, C))) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    const int vec = (C % 4 == 0) ? 4 : 1;
    const size_t smem = gn_smem_bytes(C, vec, 3);
    if (smem > 64 * 1024) return DCV_ERR_UNSUPPORTED;
    double* partial = need_param_grads ? reinterpret_cast<double*>(ws) : nullptr;
    if (vec == 4) hipLaunchKernelGGL((gn_bwd_kernel<4>), dim3(n), dim3(DCV_CR_THREADS), smem, st, dy, x, dx, hw, C, G, gamma, save_mean, save_rstd, partial);
    else hipLaunchKernelGGL((gn_bwd_kernel<1>), dim3(n), dim3(DCV_CR_THREADS), smem, st, dy, x, dx, hw, C, G, gamma, save_mean, save_rstd, partial);
    if (need_param_grads)
        hipLaunchKernelGGL(gn_param_grad_kernel, dim3((unsigned)C), dim3(64), 0, st, partial, n, C, dgamma, dbeta);
    DCV_RETURN_LAUNCH();
}


size_t dcv_layernorm_affine_workspace_bytes(int hw, int channels) {
    if (hw <= 0 || channels <= 0) return 0;
    return dcv_align_up((size_t)2 * hw * channels * sizeof(double), 256);
}

int dcv_layernorm_affine_fwd(float* y, int n, int hw, int C, int per_channel, const float* weight, const float* bias, void* stream) {
    DCV_CHECK_ARG(y && weight && n > 0 && hw > 0 && C > 0 && (size_t)hw * C <= 0x7fffffffu);
    const size_t total = (size_t)n * hw * C;
    const size_t blocks = dcv_ceil_div(total, (size_t)256);
    hipLaunchKernelGGL(ln_affine_fwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, dcv_stream(stream), y, total, hw * C, C, per_channel ? hw : 0, weight, bias);
    DCV_RETURN_LAUNCH();
}

int dcv_layernorm_affine_bwd(const float* dy, const float* x, const float* save_mean, const float* save_rstd, int n, int hw, int C, int per_channel,
                             const float* weight, float* dxhat, float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(dy && x && save_mean && save_rstd && weight && dxhat && n > 0 && hw > 0 && C > 0 && (size_t)hw * C <= 0x7fffffffu);
    if (!ws || ws_bytes < dcv_layernorm_affine_workspace_bytes(hw, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    const int per_sample = hw * C, nw = per_channel ? per_sample : hw;
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(ln_affine_bwd_kernel, dim3((unsigned)dcv_ceil_div((size_t)per_sample, (size_t)256)), dim3(256), 0, st, dy, x, save_mean, save_rstd, n, per_sample, C,
                       per_channel ? 1 : C, per_channel ? hw : 0, weight, dxhat, part);
    if (dweight || dbias)
        hipLaunchKernelGGL(ln_affine_param_grad_kernel, dim3((unsigned)dcv_ceil_div((size_t)nw, (size_t)256)), dim3(256), 0, st, part, per_sample, C, hw, per_channel ? hw : 0, dweight, dbias);
    DCV_RETURN_LAUNCH();
}


int dcv_lrn_fwd(const float* x, float* y, size_t rows, int C, int size, float alpha, float beta, float k, void* stream) {
    DCV_CHECK_ARG(x && y && rows > 0 && C > 0 && size > 0);
    const size_t total = rows * (size_t)C, blocks = dcv_ceil_div(total, (size_t)256);
    hipLaunchKernelGGL(lrn_fwd_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, dcv_stream(stream), x, y, total, C, size, alpha / (float)size, beta, k);
    DCV_RETURN_LAUNCH();
}

int dcv_lrn_bwd(const float* dy, const float* x, float* dx, size_t rows, int C, int size, float alpha, float beta, float k, void* stream) {
    DCV_CHECK_ARG(dy && x && dx && rows > 0 && C > 0 && size > 0);
    const size_t total = rows * (size_t)C, blocks = dcv_ceil_div(total, (size_t)256);
    hipLaunchKernelGGL(lrn_bwd_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, dcv_stream(stream), dy, x, dx, total, C, size, alpha / (float)size, beta, k);
    DCV_RETURN_LAUNCH();
}


// ------------------------------------------------------------------------------------------------ fused chain entry points
static inline bool nf_sample_fits_registers(int hw, int C, int vec) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_NORM_CACHED"); enabled = e ? atoi(e) : 1; }
    const int cv = C / vec;
    if (!enabled || cv > DCV_CR_THREADS) return false;
    const int rpi = DCV_CR_THREADS / cv;
    return (hw + rpi - 1) / rpi <= NF_NR;
}

static inline size_t nf_partial_bytes(int n, int hw, int C) {
    const size_t rows = (size_t)n * hw;
    size_t a = (size_t)dcv_colreduce_blocks(rows, C) * 2 * C, b = (size_t)n * 4 * C;
    return dcv_align_up((a > b ? a : b) * sizeof(double), 256);
}

size_t dcv_norm_chain_workspace_bytes(int n, int hw, int channels) {
    if (n <= 0 || hw <= 0 || channels <= 0) return 0;
    return nf_partial_bytes(n, hw, channels) + dcv_align_up((size_t)3 * channels * sizeof(float), 256) + dcv_align_up((size_t)n * 3 * channels * sizeof(float), 256);
}

int dcv_bn_gn_fwd_train(const float* x, float* y, int n, int hw, int C,
                        const float* bn_gamma, const float* bn_beta, float* running_mean, float* running_var, int64_t* nbt, float momentum, float bn_eps,
                        float* bn_save_mean, float* bn_save_rstd,
                        int G, const float* gn_gamma, const float* gn_beta, float gn_eps, float* gn_save_mean, float* gn_save_rstd,
                        const double* stats, int stats_rows, void* ws, size_t ws_bytes, void* stream) {
    return dcv_bn_gn_pool_fwd_train(x, y, n, hw, C, bn_gamma, bn_beta, running_mean, running_var, nbt, momentum, bn_eps, bn_save_mean, bn_save_rstd,
                                    G, gn_gamma, gn_beta, gn_eps, gn_save_mean, gn_save_rstd, stats, stats_rows, 0, 0, ws, ws_bytes, stream);
}

int dcv_bn_gn_pool_fwd_train(const float* x, float* y, int n, int hw, int C,
                             const float* bn_gamma, const float* bn_beta, float* running_mean, float* running_var, int64_t* nbt, float momentum, float bn_eps,
                             float* bn_save_mean, float* bn_save_rstd,
                             int G, const float* gn_gamma, const float* gn_beta, float gn_eps, float* gn_save_mean, float* gn_save_rstd,
                             const double* stats, int stats_rows, int width, int pool, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(x && y && bn_save_mean && bn_save_rstd && ws && n > 0 && hw > 0 && C > 0 && G >= 0 && (stats == nullptr || stats_rows > 0));
    DCV_CHECK_ARG(pool >= 0 && (pool == 0 || (G > 0 && width > 0 && hw % width == 0 && width % pool == 0 && (hw / width) % pool == 0)));
    DCV_CHECK_ARG(G == 0 || (C % G == 0 && gn_save_mean && gn_save_rstd));
    if (ws_bytes < dcv_norm_chain_workspace_bytes(n, hw, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    const size_t rows = (size_t)n * hw;
    double* partial = reinterpret_cast<double*>(ws);
    float* scale_shift = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + nf_partial_bytes(n, hw, C));
    const unsigned nb = dcv_colreduce_blocks(rows, C);
    const int vec = (C % 4 == 0) ? 4 : 1;
    if (stats == nullptr) {
        SumSqLoader ld{x};
        if (vec == 4) hipLaunchKernelGGL((colreduce2_kernel<4, SumSqLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
        else hipLaunchKernelGGL((colreduce2_kernel<1, SumSqLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    }
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((unsigned)C), dim3(FIN_THREADS), 0, st, stats ? stats : partial, stats ? stats_rows : (int)nb, rows, C, bn_gamma, bn_beta,
                       running_mean, running_var, nbt, momentum, bn_eps, bn_save_mean, bn_save_rstd, scale_shift, stats ? 1 : 0);
    if (G > 0) {
        const size_t smem = 2 * (size_t)C * sizeof(double) + (size_t)4 * C * sizeof(float) + (size_t)2 * DCV_CR_THREADS * vec * sizeof(float);
        if (smem > 64 * 1024) return DCV_ERR_UNSUPPORTED;
        const bool cached = nf_sample_fits_registers(hw, C, vec);
        if (vec == 4 && cached) hipLaunchKernelGGL((nf_bn_gn_apply_kernel<4, true>), dim3(n), dim3(DCV_CR_THREADS), smem, st, x, y, hw, C, G, scale_shift, gn_gamma, gn_beta, gn_eps, gn_save_mean, gn_save_rstd, width, pool);
        else if (vec == 4) hipLaunchKernelGGL((nf_bn_gn_apply_kernel<4, false>), dim3(n), dim3(DCV_CR_THREADS), smem, st, x, y, hw, C, G, scale_shift, gn_gamma, gn_beta, gn_eps, gn_save_mean, gn_save_rstd, width, pool);
        else if (cached) hipLaunchKernelGGL((nf_bn_gn_apply_kernel<1, true>), dim3(n), dim3(DCV_CR_THREADS), smem, st, x, y, hw, C, G, scale_shift, gn_gamma, gn_beta, gn_eps, gn_save_mean, gn_save_rstd, width, pool);
        else hipLaunchKernelGGL((nf_bn_gn_apply_kernel<1, false>), dim3(n), dim3(DCV_CR_THREADS), smem, st, x, y, hw, C, G, scale_shift, gn_gamma, gn_beta, gn_eps, gn_save_mean, gn_save_rstd, width, pool);
    } else if (vec == 4) {
        const size_t nv = rows * (size_t)C / 4;
        hipLaunchKernelGGL((affine_apply_kernel<4>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, x, y, nv, C, scale_shift);
    } else {
        const size_t nv = rows * (size_t)C;
        hipLaunchKernelGGL((affine_apply_kernel<1>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, x, y, nv, C, scale_shift);
    }
    DCV_RETURN_LAUNCH();
}

int dcv_bn_gn_act_bwd(const float* dout, const float* x, float* dz, int n, int hw, int C,
                      const float* bn_gamma, const float* bn_beta, const float* bn_save_mean, const float* bn_save_rstd,
                      int G, const float* gn_gamma, const float* gn_save_mean, const float* gn_save_rstd,
                      int act, float slope,
                      float* d_bn_gamma, float* d_bn_beta, float* d_gn_gamma, float* d_gn_beta, float* d_bias,
                      void* ws, size_t ws_bytes, void* stream) {
    return dcv_bn_gn_pool_act_bwd(dout, x, dz, n, hw, C, bn_gamma, bn_beta, bn_save_mean, bn_save_rstd, G, gn_gamma, gn_save_mean, gn_save_rstd, act, slope,
                                  d_bn_gamma, d_bn_beta, d_gn_gamma, d_gn_beta, d_bias, 0, 0, ws, ws_bytes, stream);
}

int dcv_bn_gn_pool_act_bwd(const float* dout, const float* x, float* dz, int n, int hw, int C,
                           const float* bn_gamma, const float* bn_beta, const float* bn_save_mean, const float* bn_save_rstd,
                           int G, const float* gn_gamma, const float* gn_save_mean, const float* gn_save_rstd,
                           int act, float slope,
                           float* d_bn_gamma, float* d_bn_beta, float* d_gn_gamma, float* d_gn_beta, float* d_bias,
                           int width, int pool, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(dout && x && dz && bn_save_mean && bn_save_rstd && ws && n > 0 && hw > 0 && C > 0 && G >= 0);
    DCV_CHECK_ARG(pool >= 0 && (pool == 0 || (G > 0 && width > 0 && hw % width == 0 && width % pool == 0 && (hw / width) % pool == 0)));
    const NfPoolMap pm{C, width > 0 ? width : 1, pool > 0 ? width / pool : 1, pool, pool > 0 ? 1.f / (float)(pool * pool) : 1.f};
    DCV_CHECK_ARG(G == 0 || (C % G == 0 && gn_save_mean && gn_save_rstd));
    if (ws_bytes < dcv_norm_chain_workspace_bytes(n, hw, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    const size_t rows = (size_t)n * hw;
    double* partial = reinterpret_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + nf_partial_bytes(n, hw, C));
    const unsigned nb = dcv_colreduce_blocks(rows, C);
    const int vec = (C % 4 == 0) ? 4 : 1;
    const float* dv = dout;
    if (G > 0) {
        const size_t smem = 4 * (size_t)C * sizeof(double) + (size_t)9 * C * sizeof(float) + (size_t)2 * DCV_CR_THREADS * vec * sizeof(float);
        if (smem > 64 * 1024) return DCV_ERR_UNSUPPORTED;
        static int moments = -1;
        if (moments < 0) { const char* e = getenv("DCV_NORM_MOMENTS"); moments = e ? atoi(e) : 1; }
        if (moments && !d_bias && nf_sample_fits_registers(hw, C, vec)) {
            // single-pass variant: moments + per-sample algebra -> the regular finalize (parameter gradients, a / b / d) -> apply from dout and x
            float* pre = coef + dcv_align_up((size_t)3 * C * sizeof(float), 256) / sizeof(float);              // [N][3][C]
            const size_t msm = 4 * (size_t)C * sizeof(double) + (size_t)2 * C * sizeof(float) + (size_t)4 * DCV_CR_THREADS * vec * sizeof(float);
            if (msm > 64 * 1024) return DCV_ERR_UNSUPPORTED;
            if (vec == 4) hipLaunchKernelGGL((nf_moments_kernel<4>), dim3(n), dim3(DCV_CR_THREADS), msm, st, dout, x, hw, C, G, bn_gamma, bn_beta, bn_save_mean, bn_save_rstd,
                                             gn_gamma, gn_save_mean, gn_save_rstd, partial, pre, pm);
            else hipLaunchKernelGGL((nf_moments_kernel<1>), dim3(n), dim3(DCV_CR_THREADS), msm, st, dout, x, hw, C, G, bn_gamma, bn_beta, bn_save_mean, bn_save_rstd,
                                    gn_gamma, gn_save_mean, gn_save_rstd, partial, pre, pm);
            hipLaunchKernelGGL(nf_gn_bn_finalize_kernel, dim3((unsigned)C), dim3(FIN_THREADS), 0, st, partial, n, rows, C, bn_gamma, bn_save_rstd,
                               d_bn_gamma, d_bn_beta, d_gn_gamma, d_gn_beta, coef);
            if (vec == 4) hipLaunchKernelGGL((nf_moments_apply_kernel<4>), dim3(n), dim3(256), (size_t)5 * C * sizeof(float), st, dout, x, dz, hw, C, pre, coef,
                                             bn_save_mean, bn_save_rstd, act, slope, pm);
            else hipLaunchKernelGGL((nf_moments_apply_kernel<1>), dim3(n), dim3(256), (size_t)5 * C * sizeof(float), st, dout, x, dz, hw, C, pre, coef,
                                    bn_save_mean, bn_save_rstd, act, slope, pm);
            DCV_RETURN_LAUNCH();
        }
#define NF_BWD_LAUNCH(V_, C_) hipLaunchKernelGGL((nf_gn_bn_bwd_kernel<V_, C_>), dim3(n), dim3(DCV_CR_THREADS), smem, st, dout, x, dz, hw, C, G, bn_gamma, bn_beta, \
                                                 bn_save_mean, bn_save_rstd, gn_gamma, gn_save_mean, gn_save_rstd, partial, pm)
        const bool cached = nf_sample_fits_registers(hw, C, vec);
        if (vec == 4 && cached) NF_BWD_LAUNCH(4, true); else if (vec == 4) NF_BWD_LAUNCH(4, false); else if (cached) NF_BWD_LAUNCH(1, true); else NF_BWD_LAUNCH(1, false);
#undef NF_BWD_LAUNCH
        hipLaunchKernelGGL(nf_gn_bn_finalize_kernel, dim3((unsigned)C), dim3(FIN_THREADS), 0, st, partial, n, rows, C, bn_gamma, bn_save_rstd,
                           d_bn_gamma, d_bn_beta, d_gn_gamma, d_gn_beta, coef);
        dv = dz;                                             // the apply pass runs in place on dz
    } else {
        BnBwdLoader ld{dout, x, bn_save_mean, bn_save_rstd};
        if (vec == 4) hipLaunchKernelGGL((colreduce2_kernel<4, BnBwdLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
        else hipLaunchKernelGGL((colreduce2_kernel<1, BnBwdLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
        hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((unsigned)C), dim3(FIN_THREADS), 0, st, partial, (int)nb, rows, C, bn_gamma, bn_save_rstd, d_bn_gamma, d_bn_beta, coef);
    }
    NfApplyActLoader ld{dv, x, dz, bn_save_mean, bn_save_rstd, coef, C, act, slope};
    if (d_bias) {
        if (vec == 4) hipLaunchKernelGGL((nf_bn_apply_act_bwd_kernel<4>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
        else hipLaunchKernelGGL((nf_bn_apply_act_bwd_kernel<1>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
        hipLaunchKernelGGL(nf_sum_partials_kernel, dim3((unsigned)C), dim3(64), 0, st, partial, (int)nb, C, d_bias);
    } else {
        const size_t nv = rows * (size_t)C / vec;
        if (vec == 4) hipLaunchKernelGGL((nf_bn_apply_act_bwd_ew_kernel<4>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, ld, nv, C);
        else hipLaunchKernelGGL((nf_bn_apply_act_bwd_ew_kernel<1>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, ld, nv, C);
    }
    DCV_RETURN_LAUNCH();
}


// ------------------------------------------------------------------------------------------------ split-phase BatchNorm entry points
int dcv_batchnorm_stats(const float* x, size_t rows, int C, double* sums, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(x && sums && ws && rows > 0 && C > 0);
    if (ws_bytes < dcv_batchnorm_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    double* partial = reinterpret_cast<double*>(ws);
    const unsigned nb = dcv_colreduce_blocks(rows, C);
    SumSqLoader ld{x};
    if (C % 4 == 0) hipLaunchKernelGGL((colreduce2_kernel<4, SumSqLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    else hipLaunchKernelGGL((colreduce2_kernel<1, SumSqLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)(2 * C)), dim3(64), 0, st, partial, (int)nb, 2, C, sums);
    DCV_RETURN_LAUNCH();
}

int dcv_batchnorm_fwd_from_sums(const float* x, float* y, size_t rows, int C, const double* sums, double total_rows, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                                float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(x && y && sums && save_mean && save_rstd && ws && rows > 0 && C > 0 && total_rows >= (double)rows);
    if (ws_bytes < dcv_batchnorm_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    float* scale_shift = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + bn_ws_partial_bytes(rows, C));
    hipLaunchKernelGGL(bn_finalize_from_sums_kernel, dim3((unsigned)dcv_ceil_div(C, 64)), dim3(64), 0, st, sums, total_rows, C, gamma, beta, running_mean, running_var, nbt,
                       momentum, eps, save_mean, save_rstd, scale_shift);
    if (C % 4 == 0) {
        const size_t nv = rows * (size_t)C / 4;
        hipLaunchKernelGGL((affine_apply_kernel<4>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, x, y, nv, C, scale_shift);
    } else {
        const size_t nv = rows * (size_t)C;
        hipLaunchKernelGGL((affine_apply_kernel<1>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, x, y, nv, C, scale_shift);
    }
    DCV_RETURN_LAUNCH();
}

int dcv_batchnorm_bwd_stats(const float* dy, const float* x, size_t rows, int C, const float* save_mean, const float* save_rstd, double* sums,
                            void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(dy && x && save_mean && save_rstd && sums && ws && rows > 0 && C > 0);
    if (ws_bytes < dcv_batchnorm_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    double* partial = reinterpret_cast<double*>(ws);
    const unsigned nb = dcv_colreduce_blocks(rows, C);
    BnBwdLoader ld{dy, x, save_mean, save_rstd};
    if (C % 4 == 0) hipLaunchKernelGGL((colreduce2_kernel<4, BnBwdLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    else hipLaunchKernelGGL((colreduce2_kernel<1, BnBwdLoader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)(2 * C)), dim3(64), 0, st, partial, (int)nb, 2, C, sums);
    DCV_RETURN_LAUNCH();
}

int dcv_batchnorm_bwd_from_sums(const float* dy, const float* x, float* dx, size_t rows, int C, const float* gamma, const float* save_mean, const float* save_rstd,
                                const double* sums, double total_rows, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(dy && x && dx && save_mean && save_rstd && sums && ws && rows > 0 && C > 0 && total_rows >= (double)rows);
    if (ws_bytes < dcv_batchnorm_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + bn_ws_partial_bytes(rows, C));
    hipLaunchKernelGGL(bn_coef_from_sums_kernel, dim3((unsigned)dcv_ceil_div(C, 64)), dim3(64), 0, st, sums, total_rows, C, gamma, save_rstd, coef);
    if (C % 4 == 0) {
        const size_t nv = rows * (size_t)C / 4;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<4>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, dy, x, dx, nv, C, save_mean, save_rstd, coef);
    } else {
        const size_t nv = rows * (size_t)C;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3(dcv_stream_grid(nv, 256)), dim3(256), 0, st, dy, x, dx, nv, C, save_mean, save_rstd, coef);
    }
    DCV_RETURN_LAUNCH();
}

}  // extern "C"
