// conv2d_direct.hip — direct (non-GEMM) NHWC fp32 convolution: forward, data-gradient and weight-gradient.
//
// This is the path for the reference's stock `image_classifier` (conf/base/parameters.yml:12-17: C in {3,4,16}, 5x5 / 3x3, stride 1):
// with Cout <= 16 and K = Cin*R*S <= 144 an implicit GEMM cannot fill an MFMA tile, and the layers are HBM-bound
// (SURVEY.md §7.4 item 4), so the right tool is a register-blocked direct convolution:
//   * forward / dgrad ("s1" kernel, stride 1): one thread computes a strip of PW=4 output pixels x TCO output channels; the filter
//     slice [R][S][Cin][TCO] sits in LDS and is read as wave-broadcast ds_read_b128; each input row of the strip is loaded once
//     (16-B NHWC pixel vectors) and reused across the S taps from registers (sliding window).  dgrad is the same kernel with the
//     filter staged flipped / transposed and pad' = R-1-pad.
//   * generic kernel: any stride / dilation / kernel size, one output pixel per thread (also serves strided dgrad as a gather).
//   * wgrad: thread = (4 output channels x VEC input channels x one tap) accumulating over a chunk of pixels; per-chunk partial
//     slabs are summed by an ordered second kernel -> deterministic, no float atomics.
// Wide layers (C >= 32) are served by conv2d_mfma.hip; this file remains the any-shape fallback.
//
// Algorithmic bytes per launch (roofline "hbm"): 4 B * (N*H*W*Cin + Cout*Cin*R*S + N*Ho*Wo*Cout)   (SURVEY.md §8d).
// Replaces torch.nn.Conv2d forward (reference src/deepcv/meta/submodule_creators.py:251) and its autograd (ignite_training.py:253).
#include "common.h"

namespace {

struct DirectArgs {
    int N, Hin, Win, Cin;          // tensor read by the kernel   (fwd: x      dgrad: dy)
    int Hout, Wout, Cout;          // tensor written by the kernel (fwd: y      dgrad: dx)
    int R, S, sh, sw, ph, pw, dh, dw;
    int w_cout, w_cin;             // OIHW dims of the weight tensor as stored
    int dgrad;                     // 0: forward weight staging, 1: data-gradient staging (+ gather addressing in the generic kernel)
    int cin_chunk;                 // input channels staged in LDS per pass (multiple of VEC, <= Cin)
    int act; float slope;
    int in_nchw;                   // input tensor is NCHW (forward first layer only)
};

// Stage filter slice for output-channel tile `co0` and input channels [ci0, ci0+cc) into LDS as wl[r][s][ci][TCO].
template <int TCO>
__device__ __forceinline__ void stage_weights(const DirectArgs& a, const float* __restrict__ w, float* wl, int co0, int ci0, int cc, bool flip) {
    const int total = a.R * a.S * cc * TCO;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int co = i % TCO; int t = i / TCO;
        const int ci = t % cc; t /= cc;
        const int s = t % a.S; const int r = t / a.S;
        const int cog = co0 + co, cig = ci0 + ci;
        float v = 0.f;
        if (cog < a.Cout) {
            if (!a.dgrad) v = w[(((size_t)cog * a.w_cin + cig) * a.R + r) * a.S + s];
            else {
                // kernel-input channel = original Cout index (cig), kernel-output channel = original Cin index (cog)
                const int rr = flip ? (a.R - 1 - r) : r, ss = flip ? (a.S - 1 - s) : s;
                v = w[(((size_t)cig * a.w_cin + cog) * a.R + rr) * a.S + ss];
            }
        }
        wl[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ stride-1 strip kernel
// IN_MODE: 0 = NHWC with Cin % 4 == 0 (16-B loads), 1 = NHWC scalar, 2 = NCHW scalar
template <int TCO, int S, int IN_MODE>
__global__ __launch_bounds__(256) void conv_s1_kernel(DirectArgs a, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y) {
    constexpr int PW = 4;
    constexpr int VEC = (IN_MODE == 0) ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);

    const int strips_w = (a.Wout + PW - 1) / PW;
    const size_t total = (size_t)a.N * a.Hout * strips_w;
    const size_t sid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = sid < total;
    int n = 0, ho = 0, wo0 = 0;
    if (active) { size_t t = sid; wo0 = (int)(t % strips_w) * PW; t /= strips_w; ho = (int)(t % a.Hout); n = (int)(t / a.Hout); }
    const int co0 = blockIdx.y * TCO;

    float acc[PW][TCO];
#pragma unroll
    for (int p = 0; p < PW; ++p)
#pragma unroll
        for (int c = 0; c < TCO; ++c) acc[p][c] = 0.f;

    for (int ci0 = 0; ci0 < a.Cin; ci0 += a.cin_chunk) {
        const int cc = (a.Cin - ci0 < a.cin_chunk) ? (a.Cin - ci0) : a.cin_chunk;
        __syncthreads();
        stage_weights<TCO>(a, w, wl, co0, ci0, cc, true);
        __syncthreads();
        if (!active) continue;
        for (int r = 0; r < a.R; ++r) {
            const int hi = ho + r - a.ph;
            if (hi < 0 || hi >= a.Hin) continue;
            for (int ci = 0; ci < cc; ci += VEC) {
                float xin[PW + S - 1][VEC];
#pragma unroll
                for (int j = 0; j < PW + S - 1; ++j) {
                    const int wi = wo0 + j - a.pw;
                    const bool ok = (wi >= 0) && (wi < a.Win);
                    if constexpr (IN_MODE == 0) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok) v = *reinterpret_cast<const float4*>(x + (((size_t)n * a.Hin + hi) * a.Win + wi) * a.Cin + ci0 + ci);
                        xin[j][0] = v.x; xin[j][1] = v.y; xin[j][2] = v.z; xin[j][3] = v.w;
                    } else if constexpr (IN_MODE == 1) {
                        xin[j][0] = ok ? x[(((size_t)n * a.Hin + hi) * a.Win + wi) * a.Cin + ci0 + ci] : 0.f;
                    } else {
                        xin[j][0] = ok ? x[(((size_t)n * a.Cin + ci0 + ci) * a.Hin + hi) * a.Win + wi] : 0.f;
                    }
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float* wp = wl + ((size_t)(r * S + s) * cc + ci + v) * TCO;
                        float wv[TCO];
#pragma unroll
                        for (int c = 0; c < TCO; c += 4) {
                            const float4 t = *reinterpret_cast<const float4*>(wp + c);
                            wv[c] = t.x; wv[c + 1] = t.y; wv[c + 2] = t.z; wv[c + 3] = t.w;
                        }
#pragma unroll
                        for (int p = 0; p < PW; ++p)
#pragma unroll
                            for (int c = 0; c < TCO; ++c) acc[p][c] = fmaf(xin[p + s][v], wv[c], acc[p][c]);
                    }
                }
            }
        }
    }
    if (!active) return;
    float bv[TCO];
#pragma unroll
    for (int c = 0; c < TCO; ++c) bv[c] = (bias != nullptr && co0 + c < a.Cout) ? bias[co0 + c] : 0.f;
    const bool vec_store = (a.Cout % 4 == 0);
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int wo = wo0 + p;
        if (wo >= a.Wout) break;
        float* yp = y + (((size_t)n * a.Hout + ho) * a.Wout + wo) * a.Cout + co0;
        if (vec_store) {
#pragma unroll
            for (int c = 0; c < TCO; c += 4) {
                if (co0 + c < a.Cout) {
                    float4 o;
                    o.x = dcv_act_apply(acc[p][c] + bv[c], a.act, a.slope); o.y = dcv_act_apply(acc[p][c + 1] + bv[c + 1], a.act, a.slope);
                    o.z = dcv_act_apply(acc[p][c + 2] + bv[c + 2], a.act, a.slope); o.w = dcv_act_apply(acc[p][c + 3] + bv[c + 3], a.act, a.slope);
                    *reinterpret_cast<float4*>(yp + c) = o;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < TCO; ++c)
                if (co0 + c < a.Cout) yp[c] = dcv_act_apply(acc[p][c] + bv[c], a.act, a.slope);
        }
    }
}

// ------------------------------------------------------------------------------------------------ generic kernel (any stride / dilation / taps)
template <int TCO, int IN_MODE>
__global__ __launch_bounds__(256) void conv_generic_kernel(DirectArgs a, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y) {
    constexpr int VEC = (IN_MODE == 0) ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);
    const size_t total = (size_t)a.N * a.Hout * a.Wout;
    const size_t pid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = pid < total;
    int n = 0, ho = 0, wo = 0;
    if (active) { size_t t = pid; wo = (int)(t % a.Wout); t /= a.Wout; ho = (int)(t % a.Hout); n = (int)(t / a.Hout); }
    const int co0 = blockIdx.y * TCO;
    float acc[TCO];
#pragma unroll
    for (int c = 0; c < TCO; ++c) acc[c] = 0.f;

    for (int ci0 = 0; ci0 < a.Cin; ci0 += a.cin_chunk) {
        const int cc = (a.Cin - ci0 < a.cin_chunk) ? (a.Cin - ci0) : a.cin_chunk;
        __syncthreads();
        stage_weights<TCO>(a, w, wl, co0, ci0, cc, false);
        __syncthreads();
        if (!active) continue;
        for (int r = 0; r < a.R; ++r) {
            int hi;
            if (!a.dgrad) hi = ho * a.sh - a.ph + r * a.dh;
            else { const int t = ho + a.ph - r * a.dh; if (t < 0 || (t % a.sh) != 0) continue; hi = t / a.sh; }
            if (hi < 0 || hi >= a.Hin) continue;
            for (int s = 0; s < a.S; ++s) {
                int wi;
                if (!a.dgrad) wi = wo * a.sw - a.pw + s * a.dw;
                else { const int t = wo + a.pw - s * a.dw; if (t < 0 || (t % a.sw) != 0) continue; wi = t / a.sw; }
                if (wi < 0 || wi >= a.Win) continue;
                for (int ci = 0; ci < cc; ci += VEC) {
                    float xin[VEC];
                    if constexpr (IN_MODE == 0) {
                        const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * a.Hin + hi) * a.Win + wi) * a.Cin + ci0 + ci);
                        xin[0] = v.x; xin[1] = v.y; xin[2] = v.z; xin[3] = v.w;
                    } else if constexpr (IN_MODE == 1) {
                        xin[0] = x[(((size_t)n * a.Hin + hi) * a.Win + wi) * a.Cin + ci0 + ci];
                    } else {
                        xin[0] = x[(((size_t)n * a.Cin + ci0 + ci) * a.Hin + hi) * a.Win + wi];
                    }
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float* wp = wl + ((size_t)(r * a.S + s) * cc + ci + v) * TCO;
#pragma unroll
                        for (int c = 0; c < TCO; c += 4) {
                            const float4 t = *reinterpret_cast<const float4*>(wp + c);
                            acc[c] = fmaf(xin[v], t.x, acc[c]); acc[c + 1] = fmaf(xin[v], t.y, acc[c + 1]);
                            acc[c + 2] = fmaf(xin[v], t.z, acc[c + 2]); acc[c + 3] = fmaf(xin[v], t.w, acc[c + 3]);
                        }
                    }
                }
            }
        }
    }
    if (!active) return;
    float* yp = y + (((size_t)n * a.Hout + ho) * a.Wout + wo) * a.Cout + co0;
#pragma unroll
    for (int c = 0; c < TCO; ++c)
        if (co0 + c < a.Cout) yp[c] = dcv_act_apply(acc[c] + (bias != nullptr ? bias[co0 + c] : 0.f), a.act, a.slope);
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgradArgs {
    int N, H, W, Cin, Ho, Wo, Cout, R, S, sh, sw, ph, pw, dh, dw;
    int x_nchw;
    int units;                     // CB * CIV * R * S
    int cb, civ;                   // ceil(Cout/4), Cin/VEC
    int ub;                        // units per block (= min(units, 256))
    int nsub;                      // pixel sub-streams per block (= 256 / ub)
    size_t pixels, pix_per_chunk;
    size_t wtotal;                 // Cout*Cin*R*S
};

template <int VEC>
__global__ __launch_bounds__(256) void conv_wgrad_partial_kernel(WgradArgs a, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial) {
    __shared__ float red[256 * 4 * VEC];
    const int t = threadIdx.x;
    const int ul = t % a.ub, sub = t / a.ub;
    const int unit = blockIdx.y * a.ub + ul;
    const bool active = (sub < a.nsub) && (unit < a.units);
    // unit -> (cob, civ, r, s), s fastest
    int u = unit;
    const int s = u % a.S; u /= a.S;
    const int r = u % a.R; u /= a.R;
    const int civ = u % a.civ; const int cob = u / a.civ;
    const int co0 = cob * 4, ci0 = civ * VEC;

    float acc[4][VEC];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[j][v] = 0.f;

    const size_t p0 = (size_t)blockIdx.x * a.pix_per_chunk;
    size_t p1 = p0 + a.pix_per_chunk; if (p1 > a.pixels) p1 = a.pixels;
    if (active && p0 + sub < p1) {
        size_t p = p0 + sub;
        int wo = (int)(p % a.Wo); size_t q = p / a.Wo; int ho = (int)(q % a.Ho); int n = (int)(q / a.Ho);
        const bool co_full = (co0 + 3 < a.Cout) && (a.Cout % 4 == 0);
        for (; p < p1; p += a.nsub) {
            const int hi = ho * a.sh - a.ph + r * a.dh, wi = wo * a.sw - a.pw + s * a.dw;
            if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W) {
                float g[4];
                if (co_full) { const float4 gv = *reinterpret_cast<const float4*>(dy + p * a.Cout + co0); g[0] = gv.x; g[1] = gv.y; g[2] = gv.z; g[3] = gv.w; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = (co0 + j < a.Cout) ? dy[p * a.Cout + co0 + j] : 0.f;
                }
                float xv[VEC];
                if constexpr (VEC == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * a.H + hi) * a.W + wi) * a.Cin + ci0);
                    xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
                } else {
                    xv[0] = a.x_nchw ? x[(((size_t)n * a.Cin + ci0) * a.H + hi) * a.W + wi] : x[(((size_t)n * a.H + hi) * a.W + wi) * a.Cin + ci0];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[j][v] = fmaf(g[j], xv[v], acc[j][v]);
            }
            // advance (n, ho, wo) by nsub pixels
            wo += a.nsub;
            while (wo >= a.Wo) { wo -= a.Wo; if (++ho >= a.Ho) { ho = 0; ++n; } }
        }
    }
    // ordered merge of the sub-streams through LDS, then one partial slab per (chunk, weight)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[(j * VEC + v) * 256 + t] = acc[j][v];
    __syncthreads();
    if (sub == 0 && unit < a.units) {
        float* dst = partial + (size_t)blockIdx.x * a.wtotal;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float sum = 0.f;
                for (int k = 0; k < a.nsub; ++k) sum += red[(j * VEC + v) * 256 + k * a.ub + ul];
                const int co = co0 + j, ci = ci0 + v;
                if (co < a.Cout) dst[(((size_t)co * a.Cin + ci) * a.R + r) * a.S + s] = sum;
            }
    }
}

// dw[i] = sum over chunks of partial[chunk][i]: block = 32 consecutive weights x 8 chunk lanes (coalesced 128-B rows, chunks/8 serial loads per thread),
// merged through LDS in a fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int chunks, size_t wtotal) {
    __shared__ float red[8][32];
    const int wl = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const size_t wi = (size_t)blockIdx.x * 32 + wl;
    float s = 0.f;
    if (wi < wtotal)
        for (int c = cl; c < chunks; c += 8) s += partial[(size_t)c * wtotal + wi];
    red[cl][wl] = s;
    __syncthreads();
    if (cl == 0 && wi < wtotal) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][wl];
        dw[wi] = t;
    }
}

// ------------------------------------------------------------------------------------------------ host helpers
inline bool conv_desc_ok(const dcv_conv2d_desc* d) {
    if (!d) return false;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cout <= 0 || d->r <= 0 || d->s <= 0) return false;
    if (d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h <= 0 || d->dil_w <= 0) return false;
    const int ho = (d->h + 2 * d->pad_h - d->dil_h * (d->r - 1) - 1) / d->stride_h + 1;
    const int wo = (d->w + 2 * d->pad_w - d->dil_w * (d->s - 1) - 1) / d->stride_w + 1;
    if (ho != d->ho || wo != d->wo || ho <= 0 || wo <= 0) return false;
    if (d->act < DCV_ACT_NONE || d->act > DCV_ACT_TANH) return false;
    if (d->x_layout != DCV_LAYOUT_NHWC && d->x_layout != DCV_LAYOUT_NCHW) return false;
    return true;
}

#define DIRECT_LDS_FLOATS 12288     // 48 KiB of filter slice per block

inline int pick_tco(int cout) { return cout <= 4 ? 4 : (cout <= 8 ? 8 : 16); }

// chooses cin_chunk (multiple of vec) such that R*S*chunk*tco <= DIRECT_LDS_FLOATS ; returns 0 if even one vector does not fit
inline int pick_cin_chunk(int cin, int r, int s, int tco, int vec) {
    long per_c = (long)r * s * tco;
    long max_c = DIRECT_LDS_FLOATS / per_c;
    if (max_c >= cin) return cin;
    max_c = (max_c / vec) * vec;
    return (int)(max_c > 0 ? max_c : 0);
}

template <int TCO, int IN_MODE>
int launch_s1(const DirectArgs& a, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    const int strips_w = (a.Wout + 3) / 4;
    const size_t total = (size_t)a.N * a.Hout * strips_w;
    dim3 grid((unsigned)dcv_ceil_div(total, 256), (unsigned)dcv_ceil_div(a.Cout, TCO));
    const size_t smem = (size_t)a.R * a.S * a.cin_chunk * TCO * sizeof(float);
    switch (a.S) {
        case 1: hipLaunchKernelGGL((conv_s1_kernel<TCO, 1, IN_MODE>), grid, dim3(256), smem, st, a, x, w, bias, y); break;
        case 3: hipLaunchKernelGGL((conv_s1_kernel<TCO, 3, IN_MODE>), grid, dim3(256), smem, st, a, x, w, bias, y); break;
        case 5: hipLaunchKernelGGL((conv_s1_kernel<TCO, 5, IN_MODE>), grid, dim3(256), smem, st, a, x, w, bias, y); break;
        case 7: hipLaunchKernelGGL((conv_s1_kernel<TCO, 7, IN_MODE>), grid, dim3(256), smem, st, a, x, w, bias, y); break;
        default: return DCV_ERR_UNSUPPORTED;
    }
    return DCV_OK;
}

template <int TCO, int IN_MODE>
int launch_generic(const DirectArgs& a, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    const size_t total = (size_t)a.N * a.Hout * a.Wout;
    dim3 grid((unsigned)dcv_ceil_div(total, 256), (unsigned)dcv_ceil_div(a.Cout, TCO));
    const size_t smem = (size_t)a.R * a.S * a.cin_chunk * TCO * sizeof(float);
    hipLaunchKernelGGL((conv_generic_kernel<TCO, IN_MODE>), grid, dim3(256), smem, st, a, x, w, bias, y);
    return DCV_OK;
}

template <int IN_MODE>
int dispatch_tco(bool s1, int tco, const DirectArgs& a, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    if (s1) {
        switch (tco) {
            case 4: return launch_s1<4, IN_MODE>(a, x, w, bias, y, st);
            case 8: return launch_s1<8, IN_MODE>(a, x, w, bias, y, st);
            default: return launch_s1<16, IN_MODE>(a, x, w, bias, y, st);
        }
    }
    switch (tco) {
        case 4: return launch_generic<4, IN_MODE>(a, x, w, bias, y, st);
        case 8: return launch_generic<8, IN_MODE>(a, x, w, bias, y, st);
        default: return launch_generic<16, IN_MODE>(a, x, w, bias, y, st);
    }
}

int run_direct(DirectArgs a, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    const bool s1 = (a.sh == 1 && a.sw == 1 && a.dh == 1 && a.dw == 1) && (a.S == 1 || a.S == 3 || a.S == 5 || a.S == 7);
    const int in_mode = a.in_nchw ? 2 : ((a.Cin % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? 0 : 1);
    const int vec = in_mode == 0 ? 4 : 1;
    int tco = pick_tco(a.Cout);
    int chunk = pick_cin_chunk(a.Cin, a.R, a.S, tco, vec);
    while (chunk == 0 && tco > 4) { tco /= 2; chunk = pick_cin_chunk(a.Cin, a.R, a.S, tco, vec); }
    if (chunk == 0) return DCV_ERR_UNSUPPORTED;
    a.cin_chunk = chunk;
    int rc;
    if (in_mode == 0) rc = dispatch_tco<0>(s1, tco, a, x, w, bias, y, st);
    else if (in_mode == 1) rc = dispatch_tco<1>(s1, tco, a, x, w, bias, y, st);
    else rc = dispatch_tco<2>(s1, tco, a, x, w, bias, y, st);
    if (rc != DCV_OK) return rc;
    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
}

inline void wgrad_plan(const dcv_conv2d_desc* d, WgradArgs* a, int* chunks, int* unit_batches, int* vec_out) {
    const int vec = (d->x_layout == DCV_LAYOUT_NHWC && d->cin % 4 == 0) ? 4 : 1;
    a->N = d->n; a->H = d->h; a->W = d->w; a->Cin = d->cin; a->Ho = d->ho; a->Wo = d->wo; a->Cout = d->cout; a->R = d->r; a->S = d->s;
    a->sh = d->stride_h; a->sw = d->stride_w; a->ph = d->pad_h; a->pw = d->pad_w; a->dh = d->dil_h; a->dw = d->dil_w;
    a->x_nchw = d->x_layout == DCV_LAYOUT_NCHW;
    a->cb = (d->cout + 3) / 4; a->civ = d->cin / vec;
    a->units = a->cb * a->civ * d->r * d->s;
    a->ub = a->units < 256 ? a->units : 256;
    a->nsub = 256 / a->ub;
    a->pixels = (size_t)d->n * d->ho * d->wo;
    a->wtotal = (size_t)d->cout * d->cin * d->r * d->s;
    *unit_batches = (int)dcv_ceil_div(a->units, a->ub);
    // ~64 pixels per sub-stream per chunk, at most 512 chunks, and keep the partial slabs under 64 MiB
    size_t ch = dcv_ceil_div(a->pixels, (size_t)a->nsub * 64);
    if (ch > 512) ch = 512;
    const size_t max_by_bytes = ((size_t)64 << 20) / (a->wtotal * sizeof(float));
    if (ch > max_by_bytes) ch = max_by_bytes;
    if (ch < 1) ch = 1;
    a->pix_per_chunk = dcv_ceil_div(a->pixels, ch);
    *chunks = (int)dcv_ceil_div(a->pixels, a->pix_per_chunk);
    *vec_out = vec;
}

}  // namespace

// entry points used by conv2d.hip's dispatcher
int dcv_direct_conv2d_fwd(const dcv_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    DirectArgs a{};
    a.N = d->n; a.Hin = d->h; a.Win = d->w; a.Cin = d->cin; a.Hout = d->ho; a.Wout = d->wo; a.Cout = d->cout;
    a.R = d->r; a.S = d->s; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    a.w_cout = d->cout; a.w_cin = d->cin; a.dgrad = 0; a.act = d->act; a.slope = d->act_slope; a.in_nchw = d->x_layout == DCV_LAYOUT_NCHW;
    return run_direct(a, x, w, bias, y, st);
}

int dcv_direct_conv2d_dgrad(const dcv_conv2d_desc* d, const float* dy, const float* w, float* dx, hipStream_t st) {
    DirectArgs a{};
    a.N = d->n; a.Hin = d->ho; a.Win = d->wo; a.Cin = d->cout; a.Hout = d->h; a.Wout = d->w; a.Cout = d->cin;
    a.R = d->r; a.S = d->s; a.dh = d->dil_h; a.dw = d->dil_w;
    a.w_cout = d->cout; a.w_cin = d->cin; a.dgrad = 1; a.act = DCV_ACT_NONE; a.slope = 0.f; a.in_nchw = 0;
    const bool s1 = d->stride_h == 1 && d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1 && (d->s == 1 || d->s == 3 || d->s == 5 || d->s == 7);
    if (s1) {
        // stride-1 data gradient == forward convolution of dy with the flipped, transposed filter and pad' = R-1-pad
        a.sh = 1; a.sw = 1; a.ph = d->r - 1 - d->pad_h; a.pw = d->s - 1 - d->pad_w;
        if (a.ph < 0 || a.pw < 0) return DCV_ERR_UNSUPPORTED;
    } else {
        // gather form handled by the generic kernel (dgrad addressing: ho = (hi + pad - r*dil) / stride when divisible)
        a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w;
    }
    return run_direct(a, dy, w, nullptr, dx, st);
}

size_t dcv_direct_conv2d_wgrad_ws(const dcv_conv2d_desc* d) {
    WgradArgs a{}; int chunks, ub, vec;
    wgrad_plan(d, &a, &chunks, &ub, &vec);
    return dcv_align_up((size_t)chunks * a.wtotal * sizeof(float), 256);
}

int dcv_direct_conv2d_wgrad(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
    WgradArgs a{}; int chunks, ubatches, vec;
    wgrad_plan(d, &a, &chunks, &ubatches, &vec);
    if (!ws || ws_bytes < (size_t)chunks * a.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
    if (ubatches > 65535) return DCV_ERR_UNSUPPORTED;
    float* partial = reinterpret_cast<float*>(ws);
    dim3 grid((unsigned)chunks, (unsigned)ubatches);
    if (vec == 4 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) hipLaunchKernelGGL((conv_wgrad_partial_kernel<4>), grid, dim3(256), 0, st, a, x, dy, partial);
    else if (vec == 4) return DCV_ERR_INVALID_ARG;   // NHWC activations are always 16-B aligned (torch allocator); anything else is a caller bug
    else hipLaunchKernelGGL((conv_wgrad_partial_kernel<1>), grid, dim3(256), 0, st, a, x, dy, partial);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)dcv_ceil_div(a.wtotal, 32)), dim3(256), 0, st, partial, dw, chunks, a.wtotal);
    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
}
