// norm_chain.hip — the normalisation chain of one reference layer, Conv -> act -> BatchNorm [-> GroupNorm] [-> AvgPool] (op order of `layer()`, reference
// src/deepcv/meta/nn.py:553; norm stack of `normalization_techniques()`, nn.py:448-516), as STREAMING kernels around tiny finalize kernels.
//
// With x the activated convolution output, u = (x - bn_mean[c]) * bn_rstd[c] and v = bn_gamma * u + bn_beta the BatchNorm output, the GroupNorm of v is, per
// sample, an affine function of x:  out[n, p, c] = S[n, c] * x[n, p, c] + T[n, c].  Every statistic that S and T need follows from per-(sample, channel) sums
// of x and x^2 — which the convolution kernel's epilogue already emits — so the forward pass is
//     [producer epilogue: sums]  ->  nc_bn_fwd_finalize (batch statistics, running statistics, a = gamma rstd, b = beta - mean a)
//                                 ->  nc_apply (prologue: the block's sample -> GroupNorm statistics, S, T from the sample's sums; then out = S x + T)
// and the backward pass, with four per-(sample, channel) moments of (dout, u)  (the algebra of norm.hip's nf_moments_kernel),
//     nc_sample_sums<DOTS> (epilogue: the sample's algebra -> P, Q0, R0 and its terms of the parameter gradients)
//                                 ->  nc_param_finalize (ordered sums over the samples: ALL parameter gradients, the BatchNorm-backward terms)
//                                 ->  nc_bwd_apply (dz = act'(x) (P dout + Q u + R)).
// Everything per-sample runs in the sample's own block (512 blocks in flight on the stock model), the finalize kernels only reduce a few KB: a 4-block
// kernel doing the per-sample algebra of 512 samples itself measured 7 - 16 us (latency-bound fp64 chains on 16 waves).  The apply kernels are streaming
// kernels (16-B loads, coefficients in registers): a dependent 8 MB -> 8 MB kernel of a replayed hipGraph costs 3.0 us on MI355X
// (tools/probes/floor_probe.hip), where the fused statistics + apply kernels of norm.hip took 6.8 - 8.9 us.
//
// Roofline: HBM.  Algorithmic bytes per element of x (4 B each): forward 1 read + 1 write (1/pool^2 of it with fused pooling), backward moments 2 reads, apply
// 2 reads + 1 write.  All reductions have a fixed order (no float atomics): results are bit-reproducible.
#include "common.h"
#include <stdlib.h>

namespace {

#define NC_THREADS 256

// pooled tensors: pixel (h, w) of the full-size map belongs to window (h / pk, w / pk); pk == 0: no pooling
struct NcPool { int W, Wo, pk; float inv; int sh; };      // sh: log2(pk) when pk is a power of two (the window index by a shift), else -1
__device__ __forceinline__ int nc_pdiv(int v, const NcPool& pm) { return pm.sh >= 0 ? v >> pm.sh : v / pm.pk; }

template <int VEC> struct NcVec { float v[VEC]; };
template <int VEC> __device__ __forceinline__ NcVec<VEC> nc_load(const float* p) {
    NcVec<VEC> r;
    if constexpr (VEC == 8) {
        const float4 q = *reinterpret_cast<const float4*>(p), q2 = *reinterpret_cast<const float4*>(p + 4);
        r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w; r.v[4] = q2.x; r.v[5] = q2.y; r.v[6] = q2.z; r.v[7] = q2.w;
    } else if constexpr (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(p); r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w; }
    else r.v[0] = p[0];
    return r;
}
template <int VEC> __device__ __forceinline__ void nc_store(float* p, const NcVec<VEC>& r) {
    if constexpr (VEC == 8) { *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]); *reinterpret_cast<float4*>(p + 4) = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]); }
    else if constexpr (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    else p[0] = r.v[0];
}
// bf16 activations (io_dtype DCV_DTYPE_BF16: the layers of the bf16 storage convolution path): widened on load, rounded to nearest even on store
typedef unsigned short nc_bf16;
template <int VEC> __device__ __forceinline__ NcVec<VEC> nc_load(const nc_bf16* p) {
    NcVec<VEC> r;
    if constexpr (VEC == 8) {      // 16-byte accesses on bf16 tensors (round 3: with four channels per lane the streaming kernels moved 8 B per lane and load)
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        r.v[0] = __uint_as_float(q.x << 16); r.v[1] = __uint_as_float(q.x & 0xffff0000u); r.v[2] = __uint_as_float(q.y << 16); r.v[3] = __uint_as_float(q.y & 0xffff0000u);
        r.v[4] = __uint_as_float(q.z << 16); r.v[5] = __uint_as_float(q.z & 0xffff0000u); r.v[6] = __uint_as_float(q.w << 16); r.v[7] = __uint_as_float(q.w & 0xffff0000u);
    } else if constexpr (VEC == 4) {
        const uint2 q = *reinterpret_cast<const uint2*>(p);
        r.v[0] = __uint_as_float(q.x << 16); r.v[1] = __uint_as_float(q.x & 0xffff0000u); r.v[2] = __uint_as_float(q.y << 16); r.v[3] = __uint_as_float(q.y & 0xffff0000u);
    } else r.v[0] = __uint_as_float((unsigned)p[0] << 16);
    return r;
}
template <int VEC> __device__ __forceinline__ void nc_store(nc_bf16* p, const NcVec<VEC>& r) {
    typedef __bf16 nc_bf16x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 nc_bf16x8 __attribute__((ext_vector_type(8)));
    if constexpr (VEC == 8) *reinterpret_cast<nc_bf16x8*>(p) = (nc_bf16x8){(__bf16)r.v[0], (__bf16)r.v[1], (__bf16)r.v[2], (__bf16)r.v[3], (__bf16)r.v[4], (__bf16)r.v[5], (__bf16)r.v[6], (__bf16)r.v[7]};
    else if constexpr (VEC == 4) *reinterpret_cast<nc_bf16x4*>(p) = (nc_bf16x4){(__bf16)r.v[0], (__bf16)r.v[1], (__bf16)r.v[2], (__bf16)r.v[3]};
    else { const __bf16 h = (__bf16)r.v[0]; p[0] = *reinterpret_cast<const nc_bf16*>(&h); }
}

// Block (x = chunk, y = sample) handles vectors [chunk * per, ...) of the sample; `per` is a multiple of NC_THREADS and CV (vectors per pixel) a power of
// two <= NC_THREADS, so a thread's channel vector (t % CV) never changes and its pixel advances by NC_THREADS / CV per iteration.  No integer division.
struct NcChunk { int n, i0, i1, cv, pix, pstep; };
__device__ __forceinline__ NcChunk nc_chunk(int nv, int CV, int per) {
    NcChunk c;
    c.n = blockIdx.y;
    const int ch = blockIdx.x, sh = __ffs(CV) - 1;
    c.i0 = ch * per + threadIdx.x;
    c.i1 = (ch + 1) * per < nv ? (ch + 1) * per : nv;
    c.cv = threadIdx.x & (CV - 1);
    c.pix = c.i0 >> sh;
    c.pstep = NC_THREADS >> sh;
    return c;
}

// Derivative of the activation from its OUTPUT y, without a branch in the streaming loops (a `switch` per element splits the loop into basic blocks, and the
// compiler then waits for every load pair before issuing the next one): ReLU family  y > 0 ? 1 : ns;  smooth family  k0 + y * (k1 + k2 * y)
// (identity 1, sigmoid y (1 - y), tanh 1 - y^2);  wr selects the family (1 or 0, exact blend).
struct NcAct { float k0, k1, k2, wr, ns; };
__device__ __forceinline__ float nc_act_grad(float y, const NcAct& a) {
    const float gs = a.k0 + y * (a.k1 + a.k2 * y), gr = y > 0.f ? 1.f : a.ns;
    return a.wr * gr + (1.f - a.wr) * gs;
}

// 1 / sqrt(v) to double precision without the fp64 sqrt / division sequences (hundreds of cycles each): fp32 rsq estimate + two Newton steps in fp64
__device__ __forceinline__ double nc_rsqrt(double v) {
    double r = (double)__frsqrt_rn((float)v);
    r = r * (1.5 - 0.5 * v * r * r);
    return r * (1.5 - 0.5 * v * r * r);
}

// ------------------------------------------------------------------------------------------------ block-wide sums of NV doubles per thread
// red: NV * 4 doubles of LDS.  Result on every thread.  Fixed order (butterfly, then waves 0..3).
template <int NV>
__device__ __forceinline__ void nc_block_sums(double (&v)[NV], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] += __shfl_xor(v[k], off, 64);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[k * 4 + wave] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = (red[k * 4] + red[k * 4 + 1]) + (red[k * 4 + 2] + red[k * 4 + 3]);
}

#define NC_MAXC 1024              // channels: at most NC_THREADS channel vectors of 4

// GroupNorm side of the chain (G == 0: none).  Channels per group are a power of two.
struct NcGn {
    int G, cpg_shift;
    const float* gamma; const float* beta; float eps;
    float* mean; float* rstd;      // [N][G]: written by the forward apply kernel, read by the backward
};

// ------------------------------------------------------------------------------------------------ per-sample sums
// DOTS == false: q0 = sum x, q1 = sum x^2 -> part[2][C][rows], row = sample * blocks per sample + chunk.
// DOTS == true: q0 = sum d, q1 = sum d * u, q2 = sum u, q3 = sum u^2 with d the gradient w.r.t. the chain's output (read through the pooling map when pooling
// is fused: every pixel of a window receives 1 / pk^2 of the pooled gradient) and u = (x - mean[c]) * rstd[c].  Without GroupNorm these go to part[4][C][rows];
// with GroupNorm (one block per sample) the block finishes its sample: with w the normalised GroupNorm input (affine in u) and dv the gradient w.r.t. the
// BatchNorm output, S1 = sum d, S2 = sum d * w (GroupNorm parameter gradients), T1 = sum dv, T2 = sum dv * u (BatchNorm backward) -> part[4][C][N], and the
// per-sample coefficients P, Q0, R0 of dz = act'(x) * (P d + (Q0 - a dd) u + (R0 - a b)) -> pre[N][3][C]  (a dd, a b: batch-level terms, nc_param_finalize).
struct NcBwdCtx { const float* bn_gamma; const float* bn_beta; const float* bn_rstd; NcGn gn; float* pre; };

template <int VEC, bool DOTS, typename AT = float>
__global__ __launch_bounds__(NC_THREADS) void nc_sample_sums_kernel(const AT* __restrict__ x, const AT* __restrict__ dout, int HW, int C, int per,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd, NcPool pm,
                                                                     double* __restrict__ part, int rows, NcBwdCtx bc) {
    constexpr int NQ = DOTS ? 4 : 2;
    __shared__ float sm[NQ * VEC][NC_THREADS];
    const int CV = C / VEC, nv = HW * CV;
    const NcChunk ck = nc_chunk(nv, CV, per);
    const AT* xs = x + (size_t)ck.n * HW * C;
    float acc[NQ][VEC];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[q][v] = 0.f;
    NcVec<VEC> mu, rs;
    if constexpr (DOTS) { mu = nc_load<VEC>(mean + ck.cv * VEC); rs = nc_load<VEC>(rstd + ck.cv * VEC); }
    // GroupNorm epilogue: thread c < C owns channel c; its parameters are requested now, their latency hides behind the streaming loop
    float e_gr = 0.f, e_gm = 0.f, e_gbn = 1.f, e_bbn = 0.f, e_ggn = 1.f, e_brs = 0.f;
    if (DOTS && bc.gn.G > 0 && (int)threadIdx.x < C) {
        const int c = threadIdx.x, g = c >> bc.gn.cpg_shift;
        e_gr = bc.gn.rstd[(size_t)blockIdx.y * bc.gn.G + g]; e_gm = bc.gn.mean[(size_t)blockIdx.y * bc.gn.G + g];
        if (bc.bn_gamma) e_gbn = bc.bn_gamma[c];
        if (bc.bn_beta) e_bbn = bc.bn_beta[c];
        if (bc.gn.gamma) e_ggn = bc.gn.gamma[c];
        e_brs = bc.bn_rstd[c];
    }
    constexpr int B = 4;                                        // vectors per thread in flight: all loads of a batch are issued before the first use
    if constexpr (!DOTS) {
        for (int i = ck.i0; i < ck.i1; i += B * NC_THREADS) {
            NcVec<VEC> xv[B];
#pragma unroll
            for (int k = 0; k < B; ++k) xv[k] = nc_load<VEC>(xs + (size_t)(i + k * NC_THREADS < ck.i1 ? i + k * NC_THREADS : i) * VEC);
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const float ok = i + k * NC_THREADS < ck.i1 ? 1.f : 0.f;
#pragma unroll
                for (int v = 0; v < VEC; ++v) { const float a = xv[k].v[v] * ok; acc[0][v] += a; acc[1][v] += a * a; }
            }
        }
    } else if (pm.pk == 0) {
        const AT* ds = dout + (size_t)ck.n * HW * C;
        for (int i = ck.i0; i < ck.i1; i += B * NC_THREADS) {
            NcVec<VEC> xv[B], dv[B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int j = i + k * NC_THREADS < ck.i1 ? i + k * NC_THREADS : i;
                xv[k] = nc_load<VEC>(xs + (size_t)j * VEC); dv[k] = nc_load<VEC>(ds + (size_t)j * VEC);
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const float ok = i + k * NC_THREADS < ck.i1 ? 1.f : 0.f;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float u = (xv[k].v[v] - mu.v[v]) * rs.v[v] * ok, d = dv[k].v[v] * ok;
                    acc[0][v] += d; acc[1][v] += d * u; acc[2][v] += u; acc[3][v] += u * u;
                }
            }
        }
    } else {
        const int H = HW / pm.W;
        const AT* ds = dout + (size_t)ck.n * (H / pm.pk) * pm.Wo * C + ck.cv * VEC;
        int h = ck.pix / pm.W, w = ck.pix - h * pm.W;                      // one division per thread; the loop only adds
        const int hstep = ck.pstep / pm.W, wstep = ck.pstep - hstep * pm.W;
        for (int i = ck.i0; i < ck.i1; i += B * NC_THREADS) {
            NcVec<VEC> xv[B], dv[B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const bool ok = i + k * NC_THREADS < ck.i1;
                xv[k] = nc_load<VEC>(xs + (size_t)(ok ? i + k * NC_THREADS : i) * VEC);
                dv[k] = nc_load<VEC>(ds + (ok ? ((size_t)nc_pdiv(h, pm) * pm.Wo + nc_pdiv(w, pm)) * C : 0));
                h += hstep; w += wstep;
                if (w >= pm.W) { w -= pm.W; ++h; }
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const float ok = i + k * NC_THREADS < ck.i1 ? 1.f : 0.f;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float u = (xv[k].v[v] - mu.v[v]) * rs.v[v] * ok, d = dv[k].v[v] * pm.inv * ok;
                    acc[0][v] += d; acc[1][v] += d * u; acc[2][v] += u; acc[3][v] += u * u;
                }
            }
        }
    }
    // lanes sharing a channel vector (lane % CV): butterfly over the offsets >= CV; then the 256 / max(CV, 64) holders of each channel vector through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        if (off >= CV) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[q][v] += __shfl_xor(acc[q][v], off, 64);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int v = 0; v < VEC; ++v) sm[q * VEC + v][threadIdx.x] = acc[q][v];
    __syncthreads();
    const int stride = CV > 64 ? CV : 64, holders = NC_THREADS / stride;
    if (!DOTS || bc.gn.G == 0) {
        for (int o = threadIdx.x; o < NQ * C; o += NC_THREADS) {
            const int q = o / C, c = o - q * C;
            const int cvi = c / VEC, v = c - cvi * VEC;
            float s = 0.f;
            for (int k = 0; k < holders; ++k) s += sm[q * VEC + v][cvi + k * stride];
            part[((size_t)q * C + c) * rows + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = (double)s;
        }
        return;
    }
    if constexpr (DOTS) {
        // the sample's algebra (one block per sample, C <= NC_THREADS): thread c finishes channel c; the group sums A, B go through LDS
        __shared__ float mom[4 * NC_THREADS];
        __shared__ double grp[2][NC_THREADS];
        for (int o = threadIdx.x; o < 4 * C; o += NC_THREADS) {
            const int q = o / C, c = o - q * C;
            const int cvi = c / VEC, v = c - cvi * VEC;
            float s = 0.f;
            for (int k = 0; k < holders; ++k) s += sm[q * VEC + v][cvi + k * stride];
            mom[o] = s;
        }
        __syncthreads();
        const int c = threadIdx.x, n = blockIdx.y, N = gridDim.y, sh = bc.gn.cpg_shift, cpg = 1 << sh;
        const bool own = c < C;
        const double gr = (double)e_gr, al = (double)e_gbn * gr, be = ((double)e_bbn - (double)e_gm) * gr;
        const double m0 = own ? (double)mom[c] : 0.0, m1 = own ? (double)mom[C + c] : 0.0, u1 = own ? (double)mom[2 * C + c] : 0.0, u2 = own ? (double)mom[3 * C + c] : 0.0;
        const double s2 = al * m1 + be * m0;                                                       // sum d * w
        if (own) { grp[0][c] = (double)e_ggn * m0; grp[1][c] = (double)e_ggn * s2; }
        __syncthreads();
        if (own) {
            const int g0 = (c >> sh) << sh;
            double A = 0.0, Bs = 0.0;                                                               // group sums of gn_gamma * S1, gn_gamma * S2
            for (int k = g0; k < g0 + cpg; ++k) { A += grp[0][k]; Bs += grp[1][k]; }
            const double hw = (double)HW, inv_m = 1.0 / ((double)cpg * hw);
            const double ca = (double)e_ggn * gr, cb = A * inv_m * gr, cd = Bs * inv_m * gr;
            part[(size_t)c * N + n] = m0;                                                          // S1 = sum d
            part[((size_t)C + c) * N + n] = s2;                                                    // S2 = sum d * w
            part[(2 * (size_t)C + c) * N + n] = ca * m0 - cb * hw - cd * (al * u1 + be * hw);      // T1 = sum dv
            part[(3 * (size_t)C + c) * N + n] = ca * m1 - cb * u1 - cd * (al * u2 + be * u1);      // T2 = sum dv * u
            const double a = (double)e_gbn * (double)e_brs;
            float* pf = bc.pre + (size_t)n * 3 * C;
            pf[c] = (float)(a * ca); pf[C + c] = (float)(-a * al * cd); pf[2 * C + c] = (float)(-a * (cb + be * cd));
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
// One block per channel: batch sums of x, x^2 over every row of the producer's partials -> BatchNorm statistics, running statistics and the affine
// v = a x + b (a = gamma * rstd, b = beta - mean * a) -> ab[2][C].
__global__ __launch_bounds__(NC_THREADS) void nc_bn_fwd_finalize_kernel(const double* __restrict__ part, int rows, double count, int C,
                                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                         float* running_mean, float* running_var, long long* nbt, float momentum, float eps,
                                                                         float* __restrict__ save_mean, float* __restrict__ save_rstd, float* __restrict__ ab) {
    __shared__ double red[2 * 4];
    const int t = threadIdx.x, c = blockIdx.x;
    // the scalars of thread 0 are requested before the reduction: their latency hides behind it
    float g = 1.f, b = 0.f, rm = 0.f, rv = 0.f;
    if (t == 0) { g = gamma ? gamma[c] : 1.f; b = beta ? beta[c] : 0.f; rm = running_mean ? running_mean[c] : 0.f; rv = running_var ? running_var[c] : 0.f; }
    double s[2] = {0.0, 0.0};
    const double* p0 = part + (size_t)c * rows; const double* p1 = part + ((size_t)C + c) * rows;
    // (four rows per thread in flight, four partial sums combined in a fixed order: with one dependent load per iteration a channel of 12 544 rows — a 56 x 56 map at
    //  512 images — took 14 us on its one block)
    {
        double q0[4] = {0.0, 0.0, 0.0, 0.0}, q1[4] = {0.0, 0.0, 0.0, 0.0};
        int j = t;
        for (; j + 3 * NC_THREADS < rows; j += 4 * NC_THREADS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { q0[k] += p0[j + k * NC_THREADS]; q1[k] += p1[j + k * NC_THREADS]; }
        }
        for (int k = 0; j < rows; j += NC_THREADS, ++k) { q0[k] += p0[j]; q1[k] += p1[j]; }
        s[0] = (q0[0] + q0[1]) + (q0[2] + q0[3]); s[1] = (q1[0] + q1[1]) + (q1[2] + q1[3]);
    }
    nc_block_sums<2>(s, red);
    if (t != 0) return;
    if (c == 0 && nbt) nbt[0] += 1;
    const double mean = s[0] / count;
    double var = s[1] / count - mean * mean; if (var < 0.0) var = 0.0;
    const float rstd = (float)nc_rsqrt(var + (double)eps);
    save_mean[c] = (float)mean; save_rstd[c] = rstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * rm + momentum * (float)mean;
    if (running_var) { const double unb = count > 1.0 ? var * count / (count - 1.0) : var; running_var[c] = (1.f - momentum) * rv + momentum * (float)unb; }
    const float sc = g * rstd;
    ab[c] = sc; ab[C + c] = b - (float)mean * sc;
}

// out = S x + T.  Prologue (GroupNorm): the block's sample -> statistics of v = a x + b over each group from the sample's sums of x, x^2 (`part`, rps rows per
// sample), S = a * rstd_g * gn_gamma, T = (b - mean_g) * rstd_g * gn_gamma + gn_beta, shared through LDS; the first loads of x are already in flight.
// Pooled: one thread per pooled (pixel, channel vector), the mean of the affine over its window (same per-element rounding as affine + AvgPool2d).
// ADD (round 4): out = S x + T + addend — a residual link's sum with a same-shape tensor rides in the layer's apply pass (unpooled layers only)
template <int VEC, typename AT = float, bool ADD = false>
__global__ __launch_bounds__(NC_THREADS) void nc_apply_kernel(const AT* __restrict__ x, AT* __restrict__ out, int HW, int C, int per,
                                                               const float* __restrict__ ab, const double* __restrict__ part, int rows, int rps, NcGn gn, NcPool pm,
                                                               const AT* __restrict__ addend = nullptr, AT* __restrict__ pooled_x = nullptr) {
    __shared__ float sS[NC_THREADS], sT[NC_THREADS];           // GroupNorm chains: C <= NC_THREADS
    const int CV = C / VEC;
    const int n = blockIdx.y;
    constexpr int PF = 4;                                       // vectors per thread requested before the prologue
    const AT* xs = x + (size_t)n * HW * C;
    NcVec<VEC> pre[PF], prea[ADD ? PF : 1];
    const AT* as = ADD ? addend + (size_t)n * HW * C : nullptr;
    NcChunk ck;
    if (pm.pk == 0) {
        ck = nc_chunk(HW * CV, CV, per);
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int i = ck.i0 + k * NC_THREADS;
            const size_t at = (size_t)(i < ck.i1 ? i : ck.i0 < ck.i1 ? ck.i0 : 0) * VEC;
            pre[k] = nc_load<VEC>(xs + at);
            if constexpr (ADD) prea[k] = nc_load<VEC>(as + at);
        }
    }
    NcVec<VEC> S, T;
    if (gn.G > 0) {
        // thread c < C owns channel c (C <= NC_THREADS): its sums of v = a x + b and v^2 from the sample's sums of x, x^2; group totals through LDS
        __shared__ double grp[2][NC_THREADS];
        const int c = threadIdx.x, sh = gn.cpg_shift, cpg = 1 << sh;
        const bool own = c < C;
        float a = 0.f, b = 0.f, gg = 1.f, gb = 0.f;
        double s1 = 0.0, s2 = 0.0;
        if (own) {
            a = ab[c]; b = ab[C + c];
            if (gn.gamma) gg = gn.gamma[c];
            if (gn.beta) gb = gn.beta[c];
            for (int j = 0; j < rps; ++j) { s1 += part[(size_t)c * rows + (size_t)n * rps + j]; s2 += part[((size_t)C + c) * rows + (size_t)n * rps + j]; }
            const double hw = (double)HW, ad = (double)a, bd = (double)b;
            grp[0][c] = ad * s1 + hw * bd;
            grp[1][c] = ad * ad * s2 + 2.0 * ad * bd * s1 + hw * bd * bd;
        }
        __syncthreads();
        if (own) {
            const int g = c >> sh, g0 = g << sh;
            double sv = 0.0, svv = 0.0;
            for (int k = g0; k < g0 + cpg; ++k) { sv += grp[0][k]; svv += grp[1][k]; }
            const double inv_m = 1.0 / ((double)cpg * (double)HW);
            const double mean = sv * inv_m;
            double var = svv * inv_m - mean * mean; if (var < 0.0) var = 0.0;
            const float rstd = (float)nc_rsqrt(var + (double)gn.eps);
            if (blockIdx.x == 0 && c == g0) { gn.mean[(size_t)n * gn.G + g] = (float)mean; gn.rstd[(size_t)n * gn.G + g] = rstd; }
            sS[c] = a * rstd * gg;
            sT[c] = (b - (float)mean) * rstd * gg + gb;
        }
        __syncthreads();
        const int cv = threadIdx.x & (CV - 1);
#pragma unroll
        for (int v = 0; v < VEC; ++v) { S.v[v] = sS[cv * VEC + v]; T.v[v] = sT[cv * VEC + v]; }
    } else {
        const int cv = threadIdx.x & (CV - 1);
        S = nc_load<VEC>(ab + cv * VEC); T = nc_load<VEC>(ab + C + cv * VEC);
    }
    if (pm.pk == 0) {
        AT* os = out + (size_t)n * HW * C;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int i = ck.i0 + k * NC_THREADS;
            if (i < ck.i1) {
                NcVec<VEC> o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o.v[v] = pre[k].v[v] * S.v[v] + T.v[v] + (ADD ? prea[ADD ? k : 0].v[v] : 0.f);
                nc_store<VEC>(os + (size_t)i * VEC, o);
            }
        }
        for (int i = ck.i0 + PF * NC_THREADS; i < ck.i1; i += PF * NC_THREADS) {
            NcVec<VEC> xv[PF], av[ADD ? PF : 1];
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const size_t at = (size_t)(i + k * NC_THREADS < ck.i1 ? i + k * NC_THREADS : i) * VEC;
                xv[k] = nc_load<VEC>(xs + at);
                if constexpr (ADD) av[k] = nc_load<VEC>(as + at);
            }
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                NcVec<VEC> o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) o.v[v] = xv[k].v[v] * S.v[v] + T.v[v] + (ADD ? av[ADD ? k : 0].v[v] : 0.f);
                if (i + k * NC_THREADS < ck.i1) nc_store<VEC>(os + (size_t)(i + k * NC_THREADS) * VEC, o);
            }
        }
        return;
    }
    const int H = HW / pm.W, HoWo = (H / pm.pk) * pm.Wo;
    ck = nc_chunk(HoWo * CV, CV, per);
    const AT* xc = xs + ck.cv * VEC; AT* os = out + (size_t)n * HoWo * C;
    // `pooled_x` (dcv_norm_chain_desc::pool_x): the window means of x itself, for the moments epilogue of the consumer's data-gradient kernel (the mean of
    // u = (x - mean) rstd over a window is (mean of x - mean) rstd: what a pooled gradient multiplies in the sum of d u)
    AT* px = pooled_x ? pooled_x + (size_t)n * HoWo * C : nullptr;
    int ho = ck.pix / pm.Wo, wo = ck.pix - ho * pm.Wo;
    const int hstep = ck.pstep / pm.Wo, wstep = ck.pstep - hstep * pm.Wo;
    if (pm.pk == 2) {
        // 2 x 2 windows (round 4: the pooled layers of the wide specs stream 67 MB each): the four vectors of a window are requested together, two windows per
        // thread in flight; same sums in the same order as the general loop below
        const size_t rowb = (size_t)pm.W * C;
        for (int i = ck.i0; i < ck.i1; i += 2 * NC_THREADS) {
            NcVec<VEC> xv[2][4];
            int hh[2], ww[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                hh[k] = ho; ww[k] = wo;
                const bool ok = i + k * NC_THREADS < ck.i1;
                const AT* b = xc + (ok ? ((size_t)(2 * ho) * pm.W + 2 * wo) * C : 0);
                xv[k][0] = nc_load<VEC>(b); xv[k][1] = nc_load<VEC>(b + C); xv[k][2] = nc_load<VEC>(b + rowb); xv[k][3] = nc_load<VEC>(b + rowb + C);
                ho += hstep; wo += wstep;
                if (wo >= pm.Wo) { wo -= pm.Wo; ++ho; }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                NcVec<VEC> o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float a = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) a += xv[k][q].v[v] * S.v[v] + T.v[v];
                    o.v[v] = a * pm.inv;
                }
                if (i + k * NC_THREADS < ck.i1) nc_store<VEC>(os + (size_t)(i + k * NC_THREADS) * VEC, o);
                if (px != nullptr && i + k * NC_THREADS < ck.i1) {
                    NcVec<VEC> m;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) m.v[v] = (xv[k][0].v[v] + xv[k][1].v[v] + xv[k][2].v[v] + xv[k][3].v[v]) * pm.inv;
                    nc_store<VEC>(px + (size_t)(i + k * NC_THREADS) * VEC, m);
                }
            }
        }
        return;
    }
    for (int i = ck.i0; i < ck.i1; i += NC_THREADS) {
        float acc[VEC], accx[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { acc[v] = 0.f; accx[v] = 0.f; }
        for (int kh = 0; kh < pm.pk; ++kh)
            for (int kw = 0; kw < pm.pk; ++kw) {
                const NcVec<VEC> xv = nc_load<VEC>(xc + ((size_t)(ho * pm.pk + kh) * pm.W + wo * pm.pk + kw) * C);
#pragma unroll
                for (int v = 0; v < VEC; ++v) { acc[v] += xv.v[v] * S.v[v] + T.v[v]; accx[v] += xv.v[v]; }
            }
        NcVec<VEC> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.v[v] = acc[v] * pm.inv;
        nc_store<VEC>(os + (size_t)i * VEC, o);
        if (px != nullptr) {
            NcVec<VEC> m;
#pragma unroll
            for (int v = 0; v < VEC; ++v) m.v[v] = accx[v] * pm.inv;
            nc_store<VEC>(px + (size_t)i * VEC, m);
        }
        ho += hstep; wo += wstep;
        if (wo >= pm.Wo) { wo -= pm.Wo; ++ho; }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// One block per channel.  GroupNorm chain (gn = 1): ordered sums over the samples of S1, S2, T1, T2 (part[4][C][N]) -> d_gn_beta, d_gn_gamma, d_bn_beta,
// d_bn_gamma and the batch-level BatchNorm-backward terms a * T2 / rows, a * T1 / rows -> chan[2][C] (the apply kernel subtracts them from Q0, R0).
// BatchNorm only (gn = 0): sums of d, d * u over the rows of part -> d_bn_beta, d_bn_gamma and the channel's P = a, Q = -a dd, R = -a b -> coef[3][C].
__global__ __launch_bounds__(NC_THREADS) void nc_param_finalize_kernel(const double* __restrict__ part, int rows, double count, int C, int gn,
                                                                        const float* __restrict__ bn_gamma, const float* __restrict__ bn_rstd,
                                                                        float* d_bn_gamma, float* d_bn_beta, float* d_gn_gamma, float* d_gn_beta,
                                                                        float* __restrict__ chan, float* __restrict__ coef) {
    __shared__ double red[4 * 4];
    const int t = threadIdx.x, c = blockIdx.x;
    float g = 1.f, rs = 0.f;
    if (t == 0) { g = bn_gamma ? bn_gamma[c] : 1.f; rs = bn_rstd[c]; }
    const size_t cr = (size_t)C * rows;
    const double* p0 = part + (size_t)c * rows;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (gn) { for (int j = t; j < rows; j += NC_THREADS) { s[0] += p0[j]; s[1] += p0[cr + j]; s[2] += p0[2 * cr + j]; s[3] += p0[3 * cr + j]; } }
    else {
        // (as nc_bn_fwd_finalize_kernel: four rows per thread in flight)
        double q0[4] = {0.0, 0.0, 0.0, 0.0}, q1[4] = {0.0, 0.0, 0.0, 0.0};
        int j = t;
        for (; j + 3 * NC_THREADS < rows; j += 4 * NC_THREADS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { q0[k] += p0[j + k * NC_THREADS]; q1[k] += p0[cr + j + k * NC_THREADS]; }
        }
        for (int k = 0; j < rows; j += NC_THREADS, ++k) { q0[k] += p0[j]; q1[k] += p0[cr + j]; }
        s[2] = (q0[0] + q0[1]) + (q0[2] + q0[3]); s[3] = (q1[0] + q1[1]) + (q1[2] + q1[3]);
    }
    nc_block_sums<4>(s, red);
    if (t != 0) return;
    if (gn) { if (d_gn_beta) d_gn_beta[c] = (float)s[0]; if (d_gn_gamma) d_gn_gamma[c] = (float)s[1]; }
    if (d_bn_beta) d_bn_beta[c] = (float)s[2];
    if (d_bn_gamma) d_bn_gamma[c] = (float)s[3];
    const float a = g * rs, ad = a * (float)(s[3] / count), ab = a * (float)(s[2] / count);
    if (gn) { chan[c] = ad; chan[C + c] = ab; }
    else { coef[c] = a; coef[C + c] = -ad; coef[2 * C + c] = -ab; chan[c] = 0.f; chan[C + c] = 0.f; }
}

template <int VEC, typename AT = float>
__global__ __launch_bounds__(NC_THREADS) void nc_bwd_apply_kernel(const AT* __restrict__ dout, const AT* __restrict__ x, AT* __restrict__ dz, int HW, int C,
                                                                   int per, const float* __restrict__ coef, int coef_stride, const float* __restrict__ chan,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd, NcAct act, NcPool pm,
                                                                   double* __restrict__ dbp, int dbrows, float* __restrict__ dbf) {
    // dbp != nullptr: the per-channel sums of dz over this block's pixels (the convolution bias gradient) go to dbp[channel][block] — a thread's channel vector
    // never changes, so it sums its own outputs in registers; the block's threads of one channel vector meet in LDS in a fixed order (round 3: a separate
    // dcv_channel_sum pass over dz cost 8 + 11 us per layer of cifar100_deep at bs 512: 6.6 % of the bf16 step)
    __shared__ float dbred[NC_THREADS * VEC];
    float sdz[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) sdz[v] = 0.f;
    const int CV = C / VEC;
    const NcChunk ck = nc_chunk(HW * CV, CV, per);
    const float* cf = coef + (size_t)ck.n * coef_stride + ck.cv * VEC;
    const NcVec<VEC> P = nc_load<VEC>(cf);
    NcVec<VEC> Q = nc_load<VEC>(cf + C), R = nc_load<VEC>(cf + 2 * C);
    const NcVec<VEC> mu = nc_load<VEC>(mean + ck.cv * VEC), rs = nc_load<VEC>(rstd + ck.cv * VEC);
    {   // the batch-level BatchNorm-backward terms a * dd, a * b of the channel complete the per-sample Q0, R0 (zeros without GroupNorm); unconditional
        // loads: behind a branch the compiler waits for each of them separately
        const NcVec<VEC> ad = nc_load<VEC>(chan + ck.cv * VEC), ab = nc_load<VEC>(chan + C + ck.cv * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) { Q.v[v] -= ad.v[v]; R.v[v] -= ab.v[v]; }
    }
    const AT* xs = x + (size_t)ck.n * HW * C; AT* zs = dz + (size_t)ck.n * HW * C;
    constexpr int B = 4;                                        // vectors per thread in flight: all loads of a batch are issued before the first use
    if (pm.pk == 0) {
        const AT* ds = dout + (size_t)ck.n * HW * C;
        for (int i = ck.i0; i < ck.i1; i += B * NC_THREADS) {
            NcVec<VEC> xv[B], dv[B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int j = i + k * NC_THREADS < ck.i1 ? i + k * NC_THREADS : i;
                xv[k] = nc_load<VEC>(xs + (size_t)j * VEC); dv[k] = nc_load<VEC>(ds + (size_t)j * VEC);
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                NcVec<VEC> o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float u = (xv[k].v[v] - mu.v[v]) * rs.v[v];
                    o.v[v] = (P.v[v] * dv[k].v[v] + Q.v[v] * u + R.v[v]) * nc_act_grad(xv[k].v[v], act);
                }
                if (i + k * NC_THREADS < ck.i1) {
                    nc_store<VEC>(zs + (size_t)(i + k * NC_THREADS) * VEC, o);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) sdz[v] += o.v[v];
                }
            }
        }
    } else {
    const int H = HW / pm.W;
    const AT* ds = dout + (size_t)ck.n * (H / pm.pk) * pm.Wo * C + ck.cv * VEC;
    int h = ck.pix / pm.W, w = ck.pix - h * pm.W;
    const int hstep = ck.pstep / pm.W, wstep = ck.pstep - hstep * pm.W;
    for (int i = ck.i0; i < ck.i1; i += B * NC_THREADS) {
        NcVec<VEC> xv[B], dv[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const bool ok = i + k * NC_THREADS < ck.i1;
            xv[k] = nc_load<VEC>(xs + (size_t)(ok ? i + k * NC_THREADS : i) * VEC);
            dv[k] = nc_load<VEC>(ds + (ok ? ((size_t)nc_pdiv(h, pm) * pm.Wo + nc_pdiv(w, pm)) * C : 0));
            h += hstep; w += wstep;
            if (w >= pm.W) { w -= pm.W; ++h; }
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            NcVec<VEC> o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float u = (xv[k].v[v] - mu.v[v]) * rs.v[v];
                o.v[v] = (P.v[v] * (dv[k].v[v] * pm.inv) + Q.v[v] * u + R.v[v]) * nc_act_grad(xv[k].v[v], act);
            }
            if (i + k * NC_THREADS < ck.i1) {
                nc_store<VEC>(zs + (size_t)(i + k * NC_THREADS) * VEC, o);
#pragma unroll
                for (int v = 0; v < VEC; ++v) sdz[v] += o.v[v];
            }
        }
    }
    }
    if (dbp != nullptr || dbf != nullptr) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) dbred[threadIdx.x * VEC + v] = sdz[v];
        __syncthreads();
        const int block = blockIdx.y * gridDim.x + blockIdx.x;
        for (int c = threadIdx.x; c < C; c += NC_THREADS) {
            const int cv = c / VEC, v = c - cv * VEC;
            float s = 0.f;
            for (int tq = cv; tq < NC_THREADS; tq += CV) s += dbred[tq * VEC + v];       // threads cv, cv + CV, ...: fixed order
            if (dbf != nullptr) dbf[(size_t)block * C + c] = s;                         // (round 4) [block][channel] fp32: summed with every other layer's slabs by ONE dcv_slab_reduce
            else dbp[(size_t)c * dbrows + block] = (double)s;
        }
    }
}

// ------------------------------------------------------------------------------------------------ round 4: a following MAX pooling with overlapping windows
// Conv -> act -> BatchNorm -> MaxPool2d(k, stride, pad), k <= 2 stride (the 3x3 / 2 pooling behind the stem of a ResNet; BatchNorm only, bf16 tensors, eight
// channels per lane).  Unfused, the stem of resnet18_style at 512 images moved 0.82 GB four times forward (affine: read + write, pooling: read) and 5 GB backward
// (pooling backward writes the full-size gradient, the sums pass and the apply pass read it and x again): 1.9 ms of a 13.6 ms step.  Fused:
//   forward   one thread per POOLED (pixel, channel vector): out = max over the window of bf16(S x + T) — the values the unfused pair compares, first maximum wins,
//             a NaN always wins (torch) — and the window position r k + s of the winner as one byte per element; the full-size affine is never written;
//   backward  sums: per pooled element d = dout and u = (x - mean) rstd at the winner's pixel -> sum d, sum d u (the gradient of the full-size affine is zero
//             anywhere else); apply: per full-size pixel the gradients of the at most four windows it won, dz = act'(x) (P d + Q u + R) with nc_param_finalize's
//             coefficients — the full-size gradient is never written either.
struct NcMax { int H, W, Ho, Wo, k, st, pad; };

template <int K>                                                  // compile-time window (2, 3): its K x K loads are all requested before the first compare; 0: the descriptor's
__global__ __launch_bounds__(NC_THREADS) void nc_maxpool_fwd_kernel(const nc_bf16* __restrict__ x, nc_bf16* __restrict__ out, unsigned char* __restrict__ arg, int C,
                                                                     const float* __restrict__ ab, NcMax pm, unsigned total, nc_bf16* __restrict__ xwin) {
    const unsigned i = blockIdx.x * NC_THREADS + threadIdx.x;
    if (i >= total) return;
    const unsigned CV = (unsigned)C >> 3, cv = i % CV;
    unsigned p = i / CV;
    const int wo = (int)(p % (unsigned)pm.Wo); p /= (unsigned)pm.Wo;
    const int ho = (int)(p % (unsigned)pm.Ho), n = (int)(p / (unsigned)pm.Ho);
    const NcVec<8> S = nc_load<8>(ab + cv * 8), T = nc_load<8>(ab + C + cv * 8);
    const nc_bf16* xs = x + (size_t)n * pm.H * pm.W * C + cv * 8;
    const int h0 = ho * pm.st - pm.pad, w0 = wo * pm.st - pm.pad;
    float best[8], bx[8]; unsigned bi[8];                       // bx: x itself at the winner (`xwin`, dcv_norm_chain_desc::pool_x: what a pooled gradient multiplies in sum d u)
    const unsigned first = (unsigned)((h0 < 0 ? -h0 : 0) * pm.k + (w0 < 0 ? -w0 : 0));      // torch initialises the index to the window's first in-range element
#pragma unroll
    for (int v = 0; v < 8; ++v) { best[v] = -INFINITY; bi[v] = first; bx[v] = 0.f; }
    auto take = [&](const NcVec<8>& xv, unsigned pos) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const float a = (float)(__bf16)(xv.v[v] * S.v[v] + T.v[v]);          // the bf16 value the unfused layer stores and the pooling compares
            if (a > best[v] || a != a) { best[v] = a; bi[v] = pos; bx[v] = xv.v[v]; }
        }
    };
    if constexpr (K > 0) {
        NcVec<8> xv[K * K];
        bool ok[K * K];
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int sx = 0; sx < K; ++sx) {
                const int h = h0 + r, w = w0 + sx;
                ok[r * K + sx] = h >= 0 && h < pm.H && w >= 0 && w < pm.W;
                xv[r * K + sx] = nc_load<8>(xs + (ok[r * K + sx] ? ((size_t)h * pm.W + w) * C : 0));
            }
#pragma unroll
        for (int q = 0; q < K * K; ++q) if (ok[q]) take(xv[q], (unsigned)q);
    } else {
        for (int r = 0; r < pm.k; ++r) {
            const int h = h0 + r;
            if (h < 0 || h >= pm.H) continue;
            for (int sx = 0; sx < pm.k; ++sx) {
                const int w = w0 + sx;
                if (w < 0 || w >= pm.W) continue;
                take(nc_load<8>(xs + ((size_t)h * pm.W + w) * C), (unsigned)(r * pm.k + sx));
            }
        }
    }
    NcVec<8> o;
#pragma unroll
    for (int v = 0; v < 8; ++v) o.v[v] = best[v];
    nc_store<8>(out + (size_t)i * 8, o);
    if (xwin != nullptr) {
#pragma unroll
        for (int v = 0; v < 8; ++v) o.v[v] = bx[v];
        nc_store<8>(xwin + (size_t)i * 8, o);
    }
    *reinterpret_cast<uint2*>(arg + (size_t)i * 8) = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
}

// sums of d and d u over the pooled elements of this block's chunk of one sample -> part[2][C][rows] (row = sample * gridDim.x + chunk)
__global__ __launch_bounds__(NC_THREADS) void nc_maxpool_bwd_sums_kernel(const nc_bf16* __restrict__ x, const nc_bf16* __restrict__ dout, const unsigned char* __restrict__ arg,
                                                                          int C, int per, const float* __restrict__ mean, const float* __restrict__ rstd, NcMax pm,
                                                                          double* __restrict__ part, int rows) {
    __shared__ float sm[16][NC_THREADS];
    const int CV = C >> 3, nvp = pm.Ho * pm.Wo * CV;
    const NcChunk ck = nc_chunk(nvp, CV, per);
    const NcVec<8> mu = nc_load<8>(mean + ck.cv * 8), rs = nc_load<8>(rstd + ck.cv * 8);
    const nc_bf16* xs = x + (size_t)ck.n * pm.H * pm.W * C + ck.cv * 8;
    const nc_bf16* ds = dout + (size_t)ck.n * nvp * 8;
    const unsigned char* as = arg + (size_t)ck.n * nvp * 8;
    float acc[2][8];
#pragma unroll
    for (int v = 0; v < 8; ++v) { acc[0][v] = 0.f; acc[1][v] = 0.f; }
    int ho = ck.pix / pm.Wo, wo = ck.pix - ho * pm.Wo;
    const int hstep = ck.pstep / pm.Wo, wstep = ck.pstep - hstep * pm.Wo;
    for (int i = ck.i0; i < ck.i1; i += NC_THREADS) {
        const NcVec<8> d = nc_load<8>(ds + (size_t)i * 8);
        const uint2 aw = *reinterpret_cast<const uint2*>(as + (size_t)i * 8);
        const int h0 = ho * pm.st - pm.pad, w0 = wo * pm.st - pm.pad;
        // the winner's x per channel: eight two-byte loads, all requested before the first use (the winner is inside the image by construction)
        float xw[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const unsigned av = ((v < 4 ? aw.x : aw.y) >> (8 * (v & 3))) & 0xffu;
            const int r = (int)(av / (unsigned)pm.k), sx = (int)av - r * pm.k;
            xw[v] = __uint_as_float((unsigned)xs[((size_t)(h0 + r) * pm.W + (w0 + sx)) * C + v] << 16);
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) { acc[0][v] += d.v[v]; acc[1][v] += d.v[v] * ((xw[v] - mu.v[v]) * rs.v[v]); }
        ho += hstep; wo += wstep;
        if (wo >= pm.Wo) { wo -= pm.Wo; ++ho; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        if (off >= CV) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int v = 0; v < 8; ++v) acc[q][v] += __shfl_xor(acc[q][v], off, 64);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int v = 0; v < 8; ++v) sm[q * 8 + v][threadIdx.x] = acc[q][v];
    __syncthreads();
    const int stride = CV > 64 ? CV : 64, holders = NC_THREADS / stride;
    for (int o = threadIdx.x; o < 2 * C; o += NC_THREADS) {
        const int q = o / C, c = o - q * C, cvi = c >> 3, v = c & 7;
        float sacc = 0.f;
        for (int k = 0; k < holders; ++k) sacc += sm[q * 8 + v][cvi + k * stride];
        part[((size_t)q * C + c) * rows + (size_t)blockIdx.y * gridDim.x + blockIdx.x] = (double)sacc;
    }
}

// dz = act'(x) (P d + Q u + R), d = the gradients of the windows this pixel won; the chunks and the bias partials (dbp / dbf) as nc_bwd_apply_kernel's
template <int K, int ST>                                         // compile-time window and stride (0: the descriptor's): the per-pixel window ranges are divisions by the stride
__global__ __launch_bounds__(NC_THREADS) void nc_maxpool_bwd_apply_kernel(const nc_bf16* __restrict__ dout, const unsigned char* __restrict__ arg, const nc_bf16* __restrict__ x,
                                                                           nc_bf16* __restrict__ dz, int C, int per, const float* __restrict__ coef,
                                                                           const float* __restrict__ mean, const float* __restrict__ rstd, NcAct act, NcMax pm,
                                                                           double* __restrict__ dbp, int dbrows, float* __restrict__ dbf) {
    __shared__ float dbred[NC_THREADS * 8];
    float sdz[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) sdz[v] = 0.f;
    const int CV = C >> 3, HW = pm.H * pm.W;
    const NcChunk ck = nc_chunk(HW * CV, CV, per);
    const float* cf = coef + ck.cv * 8;
    const NcVec<8> P = nc_load<8>(cf), Q = nc_load<8>(cf + C), R = nc_load<8>(cf + 2 * C);
    const NcVec<8> mu = nc_load<8>(mean + ck.cv * 8), rs = nc_load<8>(rstd + ck.cv * 8);
    const nc_bf16* xs = x + (size_t)ck.n * HW * C; nc_bf16* zs = dz + (size_t)ck.n * HW * C;
    const size_t pooled = (size_t)ck.n * pm.Ho * pm.Wo * C + ck.cv * 8;
    const nc_bf16* ds = dout + pooled; const unsigned char* as = arg + pooled;
    int h = ck.pix / pm.W, w = ck.pix - h * pm.W;
    const int hstep = ck.pstep / pm.W, wstep = ck.pstep - hstep * pm.W;
    const int kk = K > 0 ? K : pm.k, ss = ST > 0 ? ST : pm.st;
    constexpr int B = 2;                                          // pixels per thread in flight: x, and the gradient + winner bytes of the 2 x 2 candidate windows of each
    for (int i = ck.i0; i < ck.i1; i += B * NC_THREADS) {
        NcVec<8> xv[B], g[B][4];
        uint2 aw[B][4];
        unsigned me[B][4];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const bool live = i + b * NC_THREADS < ck.i1;
            xv[b] = nc_load<8>(xs + (size_t)(live ? i + b * NC_THREADS : i) * 8);
            // the windows o with o st - pad <= i < o st - pad + k, per axis: o_hi = (i + pad) / st and o_hi - 1 (k <= 2 st), those inside [o_lo, Ho - 1]
            const int th = h + pm.pad, tw = w + pm.pad;
            const int hhi = th / ss, whi = tw / ss;
            const int hlo = th - kk + 1 <= 0 ? 0 : (th - kk + ss) / ss, wlo = tw - kk + 1 <= 0 ? 0 : (tw - kk + ss) / ss;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ho = hhi - (q >> 1), wo = whi - (q & 1);
                const bool ok = live && ho >= hlo && ho < pm.Ho && wo >= wlo && wo < pm.Wo;
                const size_t o = ok ? ((size_t)ho * pm.Wo + wo) * C : 0;
                g[b][q] = nc_load<8>(ds + o);
                aw[b][q] = *reinterpret_cast<const uint2*>(as + o);
                me[b][q] = ok ? (unsigned)((h - (ho * ss - pm.pad)) * kk + (w - (wo * ss - pm.pad))) : 0xffffu;
            }
            h += hstep; w += wstep;
            if (w >= pm.W) { w -= pm.W; ++h; }
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            float d[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) d[v] = 0.f;
            // (summed in the order ho ascending, wo ascending: the order of the plain loop this replaces)
#pragma unroll
            for (int q = 3; q >= 0; --q)
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    const unsigned av = ((v < 4 ? aw[b][q].x : aw[b][q].y) >> (8 * (v & 3))) & 0xffu;
                    d[v] += av == me[b][q] ? g[b][q].v[v] : 0.f;
                }
            NcVec<8> o8;
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const float u = (xv[b].v[v] - mu.v[v]) * rs.v[v];
                o8.v[v] = (P.v[v] * d[v] + Q.v[v] * u + R.v[v]) * nc_act_grad(xv[b].v[v], act);
            }
            if (i + b * NC_THREADS < ck.i1) {
                nc_store<8>(zs + (size_t)(i + b * NC_THREADS) * 8, o8);
#pragma unroll
                for (int v = 0; v < 8; ++v) sdz[v] += o8.v[v];
            }
        }
    }
    if (dbp != nullptr || dbf != nullptr) {
#pragma unroll
        for (int v = 0; v < 8; ++v) dbred[threadIdx.x * 8 + v] = sdz[v];
        __syncthreads();
        const int block = blockIdx.y * gridDim.x + blockIdx.x;
        for (int c = threadIdx.x; c < C; c += NC_THREADS) {
            const int cv = c >> 3, v = c & 7;
            float sacc = 0.f;
            for (int tq = cv; tq < NC_THREADS; tq += CV) sacc += dbred[tq * 8 + v];
            if (dbf != nullptr) dbf[(size_t)block * C + c] = sacc;
            else dbp[(size_t)c * dbrows + block] = (double)sacc;
        }
    }
}

// out[c] = sum of part[c][0 .. rows): one wave per channel, lanes stride over the contiguous row (coalesced), fixed-order butterfly
__global__ __launch_bounds__(64) void nc_rows_sum_kernel(const double* __restrict__ part, int rows, float* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const double* p = part + (size_t)c * rows;
    double s = 0.0;
    for (int b = lane; b < rows; b += DCV_WAVE) s += p[b];
    s = dcv_wave_sum(s);
    if (lane == 0) out[c] = (float)s;
}

// ------------------------------------------------------------------------------------------------ host side
inline int nc_vec(int C) { return (C % 4 == 0) ? 4 : 1; }
// channels per lane: 8 on bf16 tensors without GroupNorm (16-byte accesses) when the channel count allows it
inline int nc_vec(const dcv_norm_chain_desc* d) {
    if (d->io_dtype == DCV_DTYPE_BF16 && d->groups == 0 && d->channels % 8 == 0 && ((d->channels / 8) & (d->channels / 8 - 1)) == 0 && d->channels / 8 <= NC_THREADS) return 8;
    return nc_vec(d->channels);
}
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

// blocks per sample / vectors per block for `nv` vectors per sample (`per` a multiple of NC_THREADS).  With GroupNorm a sample is ONE block (its statistics
// and backward algebra are finished inside the block): up to 32 vectors per thread; without, <= 16 vectors per thread and as many blocks as that takes.
// n (round 4): the batch — n x bps is the number of rows every partial-sum array of the chain has (statistics, moments, bias partials), and whoever sums them does
// it with a few blocks per channel: at 512 x 3 x 224 x 224 the stem of resnet18_style had 12 800 rows, the step's slab sum waited 84 us for its eight blocks.  Past
// 4096 rows a block takes up to 64 vectors per thread instead of 16.
inline void nc_split(int nv, bool one_block, int* bps, int* per, int n = 1) {
    if (one_block) { *bps = 1; *per = (int)dcv_align_up((size_t)nv, NC_THREADS); return; }
    int b = (int)dcv_ceil_div((size_t)nv, (size_t)NC_THREADS * 16);
    const int bmin = (int)dcv_ceil_div((size_t)nv, (size_t)NC_THREADS * 64);
    while ((long)n * b > 4096 && b > bmin && b > 1) b = (b + 1) / 2 > bmin ? (b + 1) / 2 : bmin;
    if (b < 1) b = 1;
    const int p = (int)dcv_align_up(dcv_ceil_div((size_t)nv, (size_t)b), NC_THREADS);
    *bps = (int)dcv_ceil_div((size_t)nv, (size_t)p); *per = p;
}

inline bool nc_enabled() {
    static int e = -1;
    if (e < 0) { const char* s = getenv("DCV_NORM_STREAMING"); e = s ? atoi(s) : 1; }
    return e != 0;
}

// timing experiments only (tools/chain_bench.py): bit k set = do not launch stage k of a chain call (1: sums, 2: finalize, 4: apply)
inline int nc_skip() {
    static int e = -1;
    if (e < 0) { const char* s = getenv("DCV_NC_SKIP"); e = s ? atoi(s) : 0; }
    return e;
}

inline bool nc_shape_ok(const dcv_norm_chain_desc* d) {
    if (!d || d->n <= 0 || d->n > 65535 || d->hw <= 0 || d->channels <= 0 || d->channels > NC_MAXC || d->groups < 0) return false;
    const int vec = nc_vec(d), cv = d->channels / vec;
    if (!pow2(cv) || cv > NC_THREADS) return false;
    if (d->groups > 0) {
        if (d->channels % d->groups != 0 || !pow2(d->channels / d->groups) || d->channels > NC_THREADS) return false;   // one thread per channel
        if ((size_t)d->hw * cv > (size_t)NC_THREADS * 32) return false;         // one block per sample
    }
    if (d->pool < 0) return false;
    if (d->pool_mode == 1) {
        // a following max pooling with overlapping windows (nc_maxpool_*): BatchNorm only, bf16 tensors, eight channels per lane, every pixel in at most 2 x 2 windows
        if (d->pool != 0 || d->groups != 0 || d->io_dtype != DCV_DTYPE_BF16 || vec != 8 || d->width <= 0 || d->hw % d->width != 0) return false;
        if (d->pool_k < 2 || d->pool_k > 15 || d->pool_stride < 1 || d->pool_k > 2 * d->pool_stride || d->pool_pad < 0 || 2 * d->pool_pad > d->pool_k) return false;
        const int H = d->hw / d->width, ho = (H + 2 * d->pool_pad - d->pool_k) / d->pool_stride + 1, wo = (d->width + 2 * d->pool_pad - d->pool_k) / d->pool_stride + 1;
        if (H + 2 * d->pool_pad < d->pool_k || d->width + 2 * d->pool_pad < d->pool_k || ho <= 0 || wo <= 0) return false;
        if ((size_t)d->n * ho * wo * cv >= ((size_t)1 << 32) - NC_THREADS) return false;
        return true;
    }
    if (d->pool_mode != 0) return false;
    // (round 4: fused pooling also without the GroupNorm stage — the BatchNorm-only layers of the wide specs: out = mean over the window of a[c] x + b[c])
    if (d->pool > 0 && !(d->width > 0 && d->hw % d->width == 0 && d->width % d->pool == 0 && (d->hw / d->width) % d->pool == 0)) return false;
    return true;
}

inline NcPool nc_pool(const dcv_norm_chain_desc* d) {
    return NcPool{d->width > 0 ? d->width : 1, d->pool > 0 ? d->width / d->pool : 1, d->pool, d->pool > 0 ? 1.f / (float)(d->pool * d->pool) : 1.f,
                  (d->pool > 0 && pow2(d->pool)) ? ilog2(d->pool) : -1};
}
inline NcMax nc_max(const dcv_norm_chain_desc* d) {
    const int H = d->hw / d->width;
    return NcMax{H, d->width, (H + 2 * d->pool_pad - d->pool_k) / d->pool_stride + 1, (d->width + 2 * d->pool_pad - d->pool_k) / d->pool_stride + 1, d->pool_k, d->pool_stride, d->pool_pad};
}
inline NcAct nc_act(int act, float slope) {
    switch (act) {
        case DCV_ACT_RELU: return NcAct{0.f, 0.f, 0.f, 1.f, 0.f};
        case DCV_ACT_LEAKY_RELU: return NcAct{0.f, 0.f, 0.f, 1.f, slope};
        case DCV_ACT_SIGMOID: return NcAct{0.f, 1.f, -1.f, 0.f, 0.f};
        case DCV_ACT_TANH: return NcAct{1.f, 0.f, -1.f, 0.f, 0.f};
        default: return NcAct{1.f, 0.f, 0.f, 0.f, 0.f};
    }
}
inline NcGn nc_gn(const dcv_norm_chain_desc* d) {
    return NcGn{d->groups, d->groups > 0 ? ilog2(d->channels / d->groups) : 0, d->gn_gamma, d->gn_beta, d->gn_eps, d->gn_mean, d->gn_rstd};
}

inline size_t nc_part_bytes(const dcv_norm_chain_desc* d) {
    int bps, per; nc_split(d->hw * (d->channels / nc_vec(d)), d->groups > 0, &bps, &per, d->n);
    return dcv_align_up((size_t)4 * d->channels * (size_t)d->n * bps * sizeof(double), 256);
}
inline size_t nc_chan_bytes(const dcv_norm_chain_desc* d) { return dcv_align_up((size_t)2 * d->channels * sizeof(float), 256); }
inline size_t nc_db_bytes(const dcv_norm_chain_desc* d) {      // [channels][blocks] partial sums of dz (the bias gradient, dcv_norm_chain_desc::d_bias)
    int bps, per; nc_split(d->hw * (d->channels / nc_vec(d)), d->groups > 0, &bps, &per, d->n);
    return dcv_align_up((size_t)d->channels * (size_t)d->n * bps * sizeof(double), 256);
}

}  // namespace

extern "C" {

int dcv_norm_chain_supported(const dcv_norm_chain_desc* d) { return (nc_enabled() && nc_shape_ok(d)) ? 1 : 0; }

size_t dcv_norm_chain_coef_floats(const dcv_norm_chain_desc* d) {
    if (!nc_shape_ok(d)) return 0;
    return (size_t)3 * d->channels * (d->groups > 0 ? (size_t)d->n : 1);
}

int dcv_norm_chain_bias_rows(const dcv_norm_chain_desc* d) {
    if (!nc_shape_ok(d)) return 0;
    int bps, per; nc_split(d->hw * (d->channels / nc_vec(d)), d->groups > 0, &bps, &per, d->n);
    return d->n * bps;
}

size_t dcv_norm_chain_ws_bytes(const dcv_norm_chain_desc* d) {
    if (!nc_shape_ok(d)) return 0;
    return nc_part_bytes(d) + nc_chan_bytes(d) + nc_db_bytes(d);
}

int dcv_norm_chain_fwd(const dcv_norm_chain_desc* d, const void* xv, void* outv, const double* stats, int stats_rows, int stats_rows_per_sample,
                       void* ws, size_t ws_bytes, void* stream) {
    const float* x = reinterpret_cast<const float*>(xv); float* out = reinterpret_cast<float*>(outv);
    const nc_bf16* x16 = reinterpret_cast<const nc_bf16*>(xv); nc_bf16* out16 = reinterpret_cast<nc_bf16*>(outv);
    DCV_CHECK_ARG(nc_shape_ok(d) && x && out && d->bn_mean && d->bn_rstd && d->coef && ws && (stats == nullptr || stats_rows > 0));
    DCV_CHECK_ARG(d->groups == 0 || (d->gn_mean && d->gn_rstd));
    DCV_CHECK_ARG(nc_vec(d) != 8 || ((reinterpret_cast<uintptr_t>(xv) | reinterpret_cast<uintptr_t>(outv)) & 15) == 0);      // (eight bf16 channels per lane: 16-byte accesses)
    if (ws_bytes < nc_part_bytes(d) + nc_chan_bytes(d)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    const int C = d->channels, vec = nc_vec(d), CV = C / vec, G = d->groups;
    int bps, per; nc_split(d->hw * CV, G > 0, &bps, &per, d->n);
    const NcPool pm = nc_pool(d);
    const NcGn gn = nc_gn(d);
    const bool io16 = d->io_dtype == DCV_DTYPE_BF16;
    float* ab = d->coef;                                          // [2][C]: the BatchNorm affine
    const double* part = stats; int rows = stats_rows, rps = stats_rows_per_sample;
    if (part == nullptr || (G > 0 && (rps <= 0 || rows != d->n * rps))) {
        // the producer did not leave (per-sample) sums: one streaming pass over x
        double* pw = reinterpret_cast<double*>(ws);
        rows = d->n * bps; rps = bps;
        const dim3 sgrid((unsigned)bps, (unsigned)d->n);
        const NcBwdCtx none{};
        const nc_bf16* no16 = nullptr; const float* no32 = nullptr;
        if (io16 && vec == 8) hipLaunchKernelGGL((nc_sample_sums_kernel<8, false, nc_bf16>), sgrid, dim3(NC_THREADS), 0, st, x16, no16, d->hw, C, per, nullptr, nullptr, pm, pw, rows, none);
        else if (io16 && vec == 4) hipLaunchKernelGGL((nc_sample_sums_kernel<4, false, nc_bf16>), sgrid, dim3(NC_THREADS), 0, st, x16, no16, d->hw, C, per, nullptr, nullptr, pm, pw, rows, none);
        else if (io16) hipLaunchKernelGGL((nc_sample_sums_kernel<1, false, nc_bf16>), sgrid, dim3(NC_THREADS), 0, st, x16, no16, d->hw, C, per, nullptr, nullptr, pm, pw, rows, none);
        else if (vec == 4) hipLaunchKernelGGL((nc_sample_sums_kernel<4, false>), sgrid, dim3(NC_THREADS), 0, st, x, no32, d->hw, C, per, nullptr, nullptr, pm, pw, rows, none);
        else hipLaunchKernelGGL((nc_sample_sums_kernel<1, false>), sgrid, dim3(NC_THREADS), 0, st, x, no32, d->hw, C, per, nullptr, nullptr, pm, pw, rows, none);
        part = pw;
    }
    if (!(nc_skip() & 2)) hipLaunchKernelGGL(nc_bn_fwd_finalize_kernel, dim3((unsigned)C), dim3(NC_THREADS), 0, st, part, rows, (double)d->n * (double)d->hw, C, d->bn_gamma, d->bn_beta,
                       d->running_mean, d->running_var, reinterpret_cast<long long*>(d->num_batches_tracked), d->momentum, d->bn_eps, d->bn_mean, d->bn_rstd, ab);
    if (d->pool_mode == 1) {
        DCV_CHECK_ARG(d->pool_arg != nullptr && d->addend == nullptr && (reinterpret_cast<uintptr_t>(d->pool_arg) & 7) == 0 && (reinterpret_cast<uintptr_t>(d->pool_x) & 15) == 0);
        if (nc_skip() & 4) DCV_RETURN_LAUNCH();
        const NcMax mx = nc_max(d);
        const size_t total = (size_t)d->n * mx.Ho * mx.Wo * CV;
        const dim3 fgrid((unsigned)dcv_ceil_div(total, NC_THREADS));
        if (mx.k == 3) hipLaunchKernelGGL((nc_maxpool_fwd_kernel<3>), fgrid, dim3(NC_THREADS), 0, st, x16, out16, d->pool_arg, C, ab, mx, (unsigned)total, reinterpret_cast<nc_bf16*>(d->pool_x));
        else if (mx.k == 2) hipLaunchKernelGGL((nc_maxpool_fwd_kernel<2>), fgrid, dim3(NC_THREADS), 0, st, x16, out16, d->pool_arg, C, ab, mx, (unsigned)total, reinterpret_cast<nc_bf16*>(d->pool_x));
        else hipLaunchKernelGGL((nc_maxpool_fwd_kernel<0>), fgrid, dim3(NC_THREADS), 0, st, x16, out16, d->pool_arg, C, ab, mx, (unsigned)total, reinterpret_cast<nc_bf16*>(d->pool_x));
        DCV_RETURN_LAUNCH();
    }
    int abps = bps, aper = per;
    if (d->pool > 0) nc_split((d->hw / (d->pool * d->pool)) * CV, G > 0, &abps, &aper, d->n);      // (a GroupNorm sample is one block: its prologue; else as many as it takes)
    const dim3 agrid((unsigned)abps, (unsigned)d->n);
    if (nc_skip() & 4) DCV_RETURN_LAUNCH();
    DCV_CHECK_ARG(d->pool_x == nullptr || (d->pool > 0 && d->pool_mode == 0 && (nc_vec(d) != 8 || (reinterpret_cast<uintptr_t>(d->pool_x) & 15) == 0)));
    nc_bf16* px16 = reinterpret_cast<nc_bf16*>(d->pool_x); float* px32 = reinterpret_cast<float*>(d->pool_x);
    if (d->addend != nullptr) {
        // out = S x + T + addend (a residual link behind the layer): the streaming instances without pooling
        DCV_CHECK_ARG(d->pool == 0 && (nc_vec(d) != 8 || (reinterpret_cast<uintptr_t>(d->addend) & 15) == 0));
        const nc_bf16* a16 = reinterpret_cast<const nc_bf16*>(d->addend); const float* a32 = reinterpret_cast<const float*>(d->addend);
        if (io16 && vec == 8) hipLaunchKernelGGL((nc_apply_kernel<8, nc_bf16, true>), agrid, dim3(NC_THREADS), 0, st, x16, out16, d->hw, C, aper, ab, part, rows, rps, gn, pm, a16);
        else if (io16 && vec == 4) hipLaunchKernelGGL((nc_apply_kernel<4, nc_bf16, true>), agrid, dim3(NC_THREADS), 0, st, x16, out16, d->hw, C, aper, ab, part, rows, rps, gn, pm, a16);
        else if (io16) hipLaunchKernelGGL((nc_apply_kernel<1, nc_bf16, true>), agrid, dim3(NC_THREADS), 0, st, x16, out16, d->hw, C, aper, ab, part, rows, rps, gn, pm, a16);
        else if (vec == 4) hipLaunchKernelGGL((nc_apply_kernel<4, float, true>), agrid, dim3(NC_THREADS), 0, st, x, out, d->hw, C, aper, ab, part, rows, rps, gn, pm, a32);
        else hipLaunchKernelGGL((nc_apply_kernel<1, float, true>), agrid, dim3(NC_THREADS), 0, st, x, out, d->hw, C, aper, ab, part, rows, rps, gn, pm, a32);
        DCV_RETURN_LAUNCH();
    }
    if (io16 && vec == 8) hipLaunchKernelGGL((nc_apply_kernel<8, nc_bf16>), agrid, dim3(NC_THREADS), 0, st, x16, out16, d->hw, C, aper, ab, part, rows, rps, gn, pm, (const nc_bf16*)nullptr, px16);
    else if (io16 && vec == 4) hipLaunchKernelGGL((nc_apply_kernel<4, nc_bf16>), agrid, dim3(NC_THREADS), 0, st, x16, out16, d->hw, C, aper, ab, part, rows, rps, gn, pm, (const nc_bf16*)nullptr, px16);
    else if (io16) hipLaunchKernelGGL((nc_apply_kernel<1, nc_bf16>), agrid, dim3(NC_THREADS), 0, st, x16, out16, d->hw, C, aper, ab, part, rows, rps, gn, pm, (const nc_bf16*)nullptr, px16);
    else if (vec == 4) hipLaunchKernelGGL((nc_apply_kernel<4>), agrid, dim3(NC_THREADS), 0, st, x, out, d->hw, C, aper, ab, part, rows, rps, gn, pm, (const float*)nullptr, px32);
    else hipLaunchKernelGGL((nc_apply_kernel<1>), agrid, dim3(NC_THREADS), 0, st, x, out, d->hw, C, aper, ab, part, rows, rps, gn, pm, (const float*)nullptr, px32);
    DCV_RETURN_LAUNCH();
}

int dcv_norm_chain_bwd(const dcv_norm_chain_desc* d, const void* doutv, const void* xv, void* dzv,
                       float* d_bn_gamma, float* d_bn_beta, float* d_gn_gamma, float* d_gn_beta, void* ws, size_t ws_bytes, void* stream) {
    const float* dout = reinterpret_cast<const float*>(doutv); const float* x = reinterpret_cast<const float*>(xv); float* dz = reinterpret_cast<float*>(dzv);
    const nc_bf16* dout16 = reinterpret_cast<const nc_bf16*>(doutv); const nc_bf16* x16 = reinterpret_cast<const nc_bf16*>(xv); nc_bf16* dz16 = reinterpret_cast<nc_bf16*>(dzv);
    const bool io16 = d && d->io_dtype == DCV_DTYPE_BF16;
    DCV_CHECK_ARG(nc_shape_ok(d) && dout && x && dz && d->bn_mean && d->bn_rstd && d->coef && ws);
    DCV_CHECK_ARG(d->groups == 0 || (d->gn_mean && d->gn_rstd));
    DCV_CHECK_ARG(nc_vec(d) != 8 || ((reinterpret_cast<uintptr_t>(doutv) | reinterpret_cast<uintptr_t>(xv) | reinterpret_cast<uintptr_t>(dzv)) & 15) == 0);
    if (ws_bytes < nc_part_bytes(d) + nc_chan_bytes(d) + (d->d_bias ? nc_db_bytes(d) : 0)) return DCV_ERR_WORKSPACE;
    DCV_CHECK_ARG(!(d->d_bias && d->d_bias_part) && (d->moments == nullptr || (d->groups == 0 && d->moments_rows > 0)));
    hipStream_t st = dcv_stream(stream);
    const int C = d->channels, vec = nc_vec(d), CV = C / vec, G = d->groups;
    int bps, per; nc_split(d->hw * CV, G > 0, &bps, &per, d->n);
    const NcPool pm = nc_pool(d);
    double* part = reinterpret_cast<double*>(ws);
    float* chan = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + nc_part_bytes(d));
    const int rows = d->n * bps;
    double* dbp = d->d_bias ? reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + nc_part_bytes(d) + nc_chan_bytes(d)) : nullptr;
    float* dbf = d->d_bias_part;
    // `moments`: the producer of dout (dcv_conv2d_bf16_dgrad_moments) left the per-row sums of dout and dout * u: no pass over dout and x here
    const double* fpart = d->moments ? d->moments : part;
    const int frows = d->moments ? d->moments_rows : rows;
    const dim3 sgrid((unsigned)bps, (unsigned)d->n);
    if (d->pool_mode == 1) {
        DCV_CHECK_ARG(d->pool_arg != nullptr && (reinterpret_cast<uintptr_t>(d->pool_arg) & 7) == 0);
        const NcMax mx = nc_max(d);
        int pbps, pper; nc_split(mx.Ho * mx.Wo * CV, false, &pbps, &pper, d->n);        // (fewer vectors than the full-size map: the rows fit the same workspace)
        const int prows = d->n * pbps;
        // (`moments`: the consumer's data-gradient kernel left the sums of dout and dout u over the POOLED elements, u at the winners — `pool_x` of the forward)
        if (!(nc_skip() & 1) && d->moments == nullptr) hipLaunchKernelGGL(nc_maxpool_bwd_sums_kernel, dim3((unsigned)pbps, (unsigned)d->n), dim3(NC_THREADS), 0, st, x16, dout16, d->pool_arg, C, pper, d->bn_mean, d->bn_rstd, mx, part, prows);
        if (!(nc_skip() & 2)) hipLaunchKernelGGL(nc_param_finalize_kernel, dim3((unsigned)C), dim3(NC_THREADS), 0, st, d->moments ? d->moments : part, d->moments ? d->moments_rows : prows, (double)d->n * (double)d->hw, C, 0, d->bn_gamma, d->bn_rstd,
                           d_bn_gamma, d_bn_beta, d_gn_gamma, d_gn_beta, chan, d->coef);
        if (nc_skip() & 4) DCV_RETURN_LAUNCH();
        if (mx.k == 3 && mx.st == 2) hipLaunchKernelGGL((nc_maxpool_bwd_apply_kernel<3, 2>), sgrid, dim3(NC_THREADS), 0, st, dout16, d->pool_arg, x16, dz16, C, per, d->coef, d->bn_mean, d->bn_rstd, nc_act(d->act, d->act_slope), mx, dbp, rows, dbf);
        else if (mx.k == 2 && mx.st == 2) hipLaunchKernelGGL((nc_maxpool_bwd_apply_kernel<2, 2>), sgrid, dim3(NC_THREADS), 0, st, dout16, d->pool_arg, x16, dz16, C, per, d->coef, d->bn_mean, d->bn_rstd, nc_act(d->act, d->act_slope), mx, dbp, rows, dbf);
        else hipLaunchKernelGGL((nc_maxpool_bwd_apply_kernel<0, 0>), sgrid, dim3(NC_THREADS), 0, st, dout16, d->pool_arg, x16, dz16, C, per, d->coef, d->bn_mean, d->bn_rstd, nc_act(d->act, d->act_slope), mx, dbp, rows, dbf);
        if (dbp) hipLaunchKernelGGL(nc_rows_sum_kernel, dim3((unsigned)C), dim3(64), 0, st, dbp, rows, d->d_bias);
        DCV_RETURN_LAUNCH();
    }
    const NcBwdCtx bc{d->bn_gamma, d->bn_beta, d->bn_rstd, nc_gn(d), d->coef};
    if ((nc_skip() & 1) || d->moments != nullptr) {}
    else if (io16 && vec == 8) hipLaunchKernelGGL((nc_sample_sums_kernel<8, true, nc_bf16>), sgrid, dim3(NC_THREADS), 0, st, x16, dout16, d->hw, C, per, d->bn_mean, d->bn_rstd, pm, part, rows, bc);
    else if (io16 && vec == 4) hipLaunchKernelGGL((nc_sample_sums_kernel<4, true, nc_bf16>), sgrid, dim3(NC_THREADS), 0, st, x16, dout16, d->hw, C, per, d->bn_mean, d->bn_rstd, pm, part, rows, bc);
    else if (io16) hipLaunchKernelGGL((nc_sample_sums_kernel<1, true, nc_bf16>), sgrid, dim3(NC_THREADS), 0, st, x16, dout16, d->hw, C, per, d->bn_mean, d->bn_rstd, pm, part, rows, bc);
    else if (vec == 4) hipLaunchKernelGGL((nc_sample_sums_kernel<4, true>), sgrid, dim3(NC_THREADS), 0, st, x, dout, d->hw, C, per, d->bn_mean, d->bn_rstd, pm, part, rows, bc);
    else hipLaunchKernelGGL((nc_sample_sums_kernel<1, true>), sgrid, dim3(NC_THREADS), 0, st, x, dout, d->hw, C, per, d->bn_mean, d->bn_rstd, pm, part, rows, bc);
    if (!(nc_skip() & 2)) hipLaunchKernelGGL(nc_param_finalize_kernel, dim3((unsigned)C), dim3(NC_THREADS), 0, st, fpart, frows, (double)d->n * (double)d->hw, C, G > 0 ? 1 : 0, d->bn_gamma, d->bn_rstd,
                       d_bn_gamma, d_bn_beta, d_gn_gamma, d_gn_beta, chan, d->coef);
    const int cstride = G > 0 ? 3 * C : 0;
    const float* ch = chan;
    if (nc_skip() & 4) DCV_RETURN_LAUNCH();
    if (io16 && vec == 8) hipLaunchKernelGGL((nc_bwd_apply_kernel<8, nc_bf16>), sgrid, dim3(NC_THREADS), 0, st, dout16, x16, dz16, d->hw, C, per, d->coef, cstride, ch, d->bn_mean, d->bn_rstd, nc_act(d->act, d->act_slope), pm, dbp, rows, dbf);
    else if (io16 && vec == 4) hipLaunchKernelGGL((nc_bwd_apply_kernel<4, nc_bf16>), sgrid, dim3(NC_THREADS), 0, st, dout16, x16, dz16, d->hw, C, per, d->coef, cstride, ch, d->bn_mean, d->bn_rstd, nc_act(d->act, d->act_slope), pm, dbp, rows, dbf);
    else if (io16) hipLaunchKernelGGL((nc_bwd_apply_kernel<1, nc_bf16>), sgrid, dim3(NC_THREADS), 0, st, dout16, x16, dz16, d->hw, C, per, d->coef, cstride, ch, d->bn_mean, d->bn_rstd, nc_act(d->act, d->act_slope), pm, dbp, rows, dbf);
    else if (vec == 4) hipLaunchKernelGGL((nc_bwd_apply_kernel<4>), sgrid, dim3(NC_THREADS), 0, st, dout, x, dz, d->hw, C, per, d->coef, cstride, ch, d->bn_mean, d->bn_rstd, nc_act(d->act, d->act_slope), pm, dbp, rows, dbf);
    else hipLaunchKernelGGL((nc_bwd_apply_kernel<1>), sgrid, dim3(NC_THREADS), 0, st, dout, x, dz, d->hw, C, per, d->coef, cstride, ch, d->bn_mean, d->bn_rstd, nc_act(d->act, d->act_slope), pm, dbp, rows, dbf);
    if (dbp) hipLaunchKernelGGL(nc_rows_sum_kernel, dim3((unsigned)C), dim3(64), 0, st, dbp, rows, d->d_bias);
    DCV_RETURN_LAUNCH();
}

}  // extern "C"
