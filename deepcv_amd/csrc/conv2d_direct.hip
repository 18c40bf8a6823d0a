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
#include "conv2d_internal.h"
#include <stdlib.h>

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

// Re-lays the OIHW filter out as wt[cout tile][R][S][Cin][TCO] (zero-padded output channels) for the scalar-cache weight path of conv_tile_kernel.
// dgrad: kernel-input channel = original Cout, kernel-output channel = original Cin, taps flipped.
__global__ __launch_bounds__(256) void prep_weights_kernel(DirectArgs a, const float* __restrict__ w, float* __restrict__ wt, int tco, int tiles, int cin_pad, int flip) {
    const int total = tiles * a.R * a.S * cin_pad * tco;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % tco; int t = i / tco;
        const int ci = t % cin_pad; t /= cin_pad;
        const int s = t % a.S; t /= a.S;
        const int r = t % a.R; const int tile = t / a.R;
        const int cog = tile * tco + co;
        float v = 0.f;
        if (cog < a.Cout && ci < a.Cin) {
            if (!a.dgrad) v = w[(((size_t)cog * a.w_cin + ci) * a.R + r) * a.S + s];
            else {
                const int rr = flip ? a.R - 1 - r : r, ss = flip ? a.S - 1 - s : s;
                v = w[(((size_t)ci * a.w_cin + cog) * a.R + rr) * a.S + ss];
            }
        }
        wt[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ LDS-tiled forward / stride-1 data gradient
// One block = 256 output pixels (TH rows x Wout columns of one image; several whole images when they are small).  The zero-padded input halo tile is
// staged ONCE into LDS with fully coalesced 16-B loads (lanes walk consecutive NHWC pixels: 1 KiB per wave-instruction) in a channel-vector-planar
// layout [Cin/4][rows][cols] of float4, so that the compute phase — one output pixel per lane — reads it with conflict-free ds_read_b128 (consecutive
// lanes = consecutive 16-B slots).  A strip-per-thread kernel reading straight from global memory was texture-addresser bound: 64-B lane strides make
// every load instruction touch 64 separate segments.  Filter weights come from the SCALAR cache: prep_weights_kernel re-lays them out as
// [cout tile][R][S][Cin_pad][TCO], their address is wave-uniform (s_load_dwordx4..x16 -> SGPR operands of v_fma), so they cost neither LDS bandwidth
// nor VGPRs.  Per lane and LDS read: 4 * TCO FMAs.  Bias + activation are fused in the epilogue.
struct TileArgs {
    int N, Hin, Win, Cin;          // tensor read by the kernel   (fwd: x      dgrad: dy)
    int Hout, Wout, Cout;          // tensor written by the kernel (fwd: y      dgrad: dx)
    int R, S, sh, sw, ph, pw, dh, dw;
    int in_nchw, act; float slope;
    int cin_pad;                   // Cin rounded up to a multiple of 4 (padding channels are zero in LDS and in the re-laid filter)
    int CC;                        // channels staged per LDS pass (multiple of 4)
    int TH, XH, XW;                // output rows per sub-tile, halo tile rows / cols
    int nsub;                      // sub-tiles (image row bands) per block
    int tiles_per_img, total_subtiles;
    int stat_per_sub;              // statistics epilogue: one row of partials per SUB-TILE (whole images sharing a block: per-sample sums) instead of per block
};

// Can the statistics epilogue keep the sums of the PPT pixel slots of a thread apart (one sub-tile = one whole image per slot)?  Register budget: PPT * TCO * 2.
#define DCV_TILE_PERK(PPT_, TCO_) ((PPT_) > 1 && (PPT_) * (TCO_) <= 16)

// ST: compile-time filter width (the taps of one filter row are unrolled so that their LDS / scalar loads are issued together), 0 = runtime.
// PPT: output pixels per thread (lane l owns pixels l, l+256, ... of the block's PPT*256-pixel tile): the per-thread setup, the scalar weight loads and
// the loop overhead are amortised over PPT independent FMA streams.
template <int TCO, int ST, int PPT>
__global__ __launch_bounds__(256) void conv_tile_kernel(TileArgs a, const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                         float* __restrict__ y, double* __restrict__ stats, const float* __restrict__ addend) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* xs = reinterpret_cast<float4*>(smem);            // [CC/4][nsub][XH][XW]
    const int t = threadIdx.x;
    const int px_per_sub = a.TH * a.Wout;
    const int co0 = blockIdx.y * TCO;
    const int plane = a.nsub * a.XH * a.XW;                  // float4 elements per channel-vector plane
    const float* __restrict__ wtile = wt + (size_t)blockIdx.y * a.R * a.S * a.cin_pad * TCO;

    int lds_off[PPT];                                        // float4 offset of the pixel's window origin inside plane 0
    size_t out_off[PPT];
    bool active[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int q = t + 256 * k;
        const int sub = q / px_per_sub, lp = q - sub * px_per_sub;
        const int hl = lp / a.Wout, wl = lp - hl * a.Wout;
        const int g = blockIdx.x * a.nsub + sub;
        const int n = g / a.tiles_per_img, ho = (g - n * a.tiles_per_img) * a.TH + hl;
        active[k] = (sub < a.nsub) && (g < a.total_subtiles) && (ho < a.Hout);
        lds_off[k] = active[k] ? (sub * a.XH + hl * a.sh) * a.XW + wl * a.sw : 0;
        out_off[k] = (((size_t)n * a.Hout + ho) * a.Wout + wl) * a.Cout + co0;
    }

    float acc[PPT][TCO];
#pragma unroll
    for (int k = 0; k < PPT; ++k)
#pragma unroll
        for (int c = 0; c < TCO; ++c) acc[k][c] = 0.f;

    for (int ci0 = 0; ci0 < a.cin_pad; ci0 += a.CC) {
        const int cc = (a.cin_pad - ci0 < a.CC) ? (a.cin_pad - ci0) : a.CC;
        const int cv = cc >> 2;
        __syncthreads();
        // ---- stage the halo tiles of this block (all sub-tiles) for channels [ci0, ci0 + cc)
        // Integer division is ~40 VALU instructions on this ISA: every index decode below is hoisted out of the copy loops (two divisions per thread),
        // the loops themselves only add strides.
        if (!a.in_nchw && (a.Cin & 3) == 0) {
            const int rowlen = a.XW * cv;                                  // float4 elements per halo row: (column, channel-vector) pairs, contiguous in HBM
            if (rowlen <= 256) {
                const int rows_per_iter = 256 / rowlen;
                const int r0 = t / rowlen, e = t - r0 * rowlen;
                const int xc = e / cv, c4 = e - xc * cv;
                const int wi = xc - a.pw;
                const bool col_ok = wi >= 0 && wi < a.Win;
                const int wic = wi < 0 ? 0 : (wi >= a.Win ? a.Win - 1 : wi);
                if (r0 < rows_per_iter) {
                    for (int sb = 0; sb < a.nsub; ++sb) {
                        const int gs = blockIdx.x * a.nsub + sb;
                        const int ns = gs / a.tiles_per_img, h0 = (gs - ns * a.tiles_per_img) * a.TH;   // wave-uniform: scalar ALU
                        const int nc = ns < a.N ? ns : a.N - 1;
                        const bool sub_ok = gs < a.total_subtiles;
                        for (int xr = r0; xr < a.XH; xr += rows_per_iter) {
                            const int hi = h0 * a.sh - a.ph + xr;
                            const bool ok = sub_ok && col_ok && hi >= 0 && hi < a.Hin;
                            const int hic = hi < 0 ? 0 : (hi >= a.Hin ? a.Hin - 1 : hi);
                            float4 v = *reinterpret_cast<const float4*>(x + (((size_t)nc * a.Hin + hic) * a.Win + wic) * a.Cin + ci0 + c4 * 4);   // unconditional (clamped) load
                            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                            xs[(size_t)c4 * plane + ((size_t)sb * a.XH + xr) * a.XW + xc] = v;
                        }
                    }
                }
            } else {
                // very wide rows: walk (column, channel-vector) pairs with a stride of 256 inside each row
                for (int sb = 0; sb < a.nsub; ++sb) {
                    const int gs = blockIdx.x * a.nsub + sb;
                    const int ns = gs / a.tiles_per_img, h0 = (gs - ns * a.tiles_per_img) * a.TH;
                    const int nc = ns < a.N ? ns : a.N - 1;
                    for (int xr = 0; xr < a.XH; ++xr) {
                        const int hi = h0 * a.sh - a.ph + xr;
                        const bool row_ok = gs < a.total_subtiles && hi >= 0 && hi < a.Hin;
                        const int hic = hi < 0 ? 0 : (hi >= a.Hin ? a.Hin - 1 : hi);
                        for (int e = t; e < rowlen; e += 256) {
                            const int xc = e / cv, c4 = e - xc * cv;
                            const int wi = xc - a.pw;
                            const int wic = wi < 0 ? 0 : (wi >= a.Win ? a.Win - 1 : wi);
                            float4 v = *reinterpret_cast<const float4*>(x + (((size_t)nc * a.Hin + hic) * a.Win + wic) * a.Cin + ci0 + c4 * 4);
                            if (!(row_ok && wi >= 0 && wi < a.Win)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                            xs[(size_t)c4 * plane + ((size_t)sb * a.XH + xr) * a.XW + xc] = v;
                        }
                    }
                }
            }
        } else {
            // scalar staging: NCHW planes (column index fastest -> coalesced) or NHWC with Cin % 4 != 0; padding channels are written as zeros.
            // Thread = fixed (column, channel) pair chosen once; loops walk rows.
            float* xf = reinterpret_cast<float*>(smem);
            const int rowlen = a.XW * cc;
            for (int e0 = 0; e0 < rowlen; e0 += 256) {
                const int e = e0 + t;
                if (e < rowlen) {
                    int xc, c;
                    if (a.in_nchw) { c = e / a.XW; xc = e - c * a.XW; } else { xc = e / cc; c = e - xc * cc; }
                    const int wi = xc - a.pw, cg = ci0 + c;
                    const bool col_ok = wi >= 0 && wi < a.Win && cg < a.Cin;
                    const int wic = wi < 0 ? 0 : (wi >= a.Win ? a.Win - 1 : wi), cgc = cg < a.Cin ? cg : a.Cin - 1;
                    for (int sb = 0; sb < a.nsub; ++sb) {
                        const int gs = blockIdx.x * a.nsub + sb;
                        const int ns = gs / a.tiles_per_img, h0 = (gs - ns * a.tiles_per_img) * a.TH;
                        const int nc = ns < a.N ? ns : a.N - 1;
                        const bool sub_ok = gs < a.total_subtiles;
                        // row stride / base of this thread's (column, channel) in the source; eight rows in flight per batch
                        const size_t rstride = a.in_nchw ? (size_t)a.Win : (size_t)a.Win * a.Cin;
                        const float* colp = a.in_nchw ? x + ((size_t)nc * a.Cin + cgc) * a.Hin * a.Win + wic : x + ((size_t)nc * a.Hin * a.Win + wic) * a.Cin + cgc;
                        float* dcol = xf + ((size_t)(c >> 2) * plane + (size_t)sb * a.XH * a.XW + xc) * 4 + (c & 3);
                        for (int xr0 = 0; xr0 < a.XH; xr0 += 8) {
                            float v[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int hi = h0 * a.sh - a.ph + xr0 + u;
                                const int hic = hi < 0 ? 0 : (hi >= a.Hin ? a.Hin - 1 : hi);
                                v[u] = colp[(size_t)hic * rstride];
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int xr = xr0 + u, hi = h0 * a.sh - a.ph + xr;
                                const bool ok = sub_ok && col_ok && hi >= 0 && hi < a.Hin;
                                if (xr < a.XH) dcol[(size_t)xr * a.XW * 4] = ok ? v[u] : 0.f;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        {
            const int SW = ST > 0 ? ST : a.S;
            for (int r = 0; r < a.R; ++r) {
                const int rowoff = (r * a.dh) * a.XW;
                const float* __restrict__ wrow = wtile + ((size_t)(r * SW) * a.cin_pad + ci0) * TCO;   // wave-uniform -> scalar loads
                for (int c4 = 0; c4 < cv; ++c4) {
                    const float4* __restrict__ pl = xs + (size_t)c4 * plane + rowoff;
                    if constexpr (ST > 0) {
#pragma unroll
                        for (int s = 0; s < ST; ++s) {
                            const float* __restrict__ w4 = wrow + ((size_t)s * a.cin_pad + c4 * 4) * TCO;
#pragma unroll
                            for (int k = 0; k < PPT; ++k) {
                                const float4 xv = pl[lds_off[k] + s * a.dw];
#pragma unroll
                                for (int c = 0; c < TCO; ++c) {
                                    acc[k][c] = fmaf(xv.x, w4[c], acc[k][c]);
                                    acc[k][c] = fmaf(xv.y, w4[TCO + c], acc[k][c]);
                                    acc[k][c] = fmaf(xv.z, w4[2 * TCO + c], acc[k][c]);
                                    acc[k][c] = fmaf(xv.w, w4[3 * TCO + c], acc[k][c]);
                                }
                            }
                        }
                    } else {
                        for (int s = 0; s < SW; ++s) {
                            const float* __restrict__ w4 = wrow + ((size_t)s * a.cin_pad + c4 * 4) * TCO;
#pragma unroll
                            for (int k = 0; k < PPT; ++k) {
                                const float4 xv = pl[lds_off[k] + s * a.dw];
#pragma unroll
                                for (int c = 0; c < TCO; ++c) {
                                    acc[k][c] = fmaf(xv.x, w4[c], acc[k][c]);
                                    acc[k][c] = fmaf(xv.y, w4[TCO + c], acc[k][c]);
                                    acc[k][c] = fmaf(xv.z, w4[2 * TCO + c], acc[k][c]);
                                    acc[k][c] = fmaf(xv.w, w4[3 * TCO + c], acc[k][c]);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    float bv[TCO];
#pragma unroll
    for (int c = 0; c < TCO; ++c) bv[c] = (bias != nullptr && co0 + c < a.Cout) ? bias[co0 + c] : 0.f;
    const bool vec_store = (a.Cout & 3) == 0;
    // per-thread sums of the activated outputs (BatchNorm statistics partials), kept per pixel slot where slots may belong to different images
    constexpr bool PERK = DCV_TILE_PERK(PPT, TCO);
    constexpr int NS = PERK ? PPT : 1;
    float ssum[NS][TCO], ssq[NS][TCO];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int c = 0; c < TCO; ++c) { ssum[j][c] = 0.f; ssq[j][c] = 0.f; }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        if (!active[k]) continue;
        float* yp = y + out_off[k];
        float o[TCO];
#pragma unroll
        for (int c = 0; c < TCO; ++c) o[c] = dcv_act_apply(acc[k][c] + bv[c], a.act, a.slope);
        if (addend != nullptr) {                                   // y = conv + addend (same layout as y): a gradient arriving over another edge of the graph
            const float* ap = addend + out_off[k];
#pragma unroll
            for (int c = 0; c < TCO; ++c)
                if (co0 + c < a.Cout) o[c] += ap[c];
        }
#pragma unroll
        for (int c = 0; c < TCO; ++c) { ssum[PERK ? k : 0][c] += o[c]; ssq[PERK ? k : 0][c] += o[c] * o[c]; }
        if (vec_store) {
#pragma unroll
            for (int c = 0; c < TCO; c += 4)
                if (co0 + c < a.Cout) *reinterpret_cast<float4*>(yp + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
        } else {
#pragma unroll
            for (int c = 0; c < TCO; ++c)
                if (co0 + c < a.Cout) yp[c] = o[c];
        }
    }
    if (stats != nullptr) {
        // sums of y and y^2 per output channel -> stats[2][Cout][rows], one row per block (rows = gridDim.x) or, with stat_per_sub (every pixel slot of a
        // thread is a different whole image), one row per sub-tile (rows = total_subtiles: per-sample sums).  All per-thread sums go through LDS
        // ([quantity][256], conflict free); quantity q is then summed by wave q % 4: 4 values per lane + one butterfly (fixed order throughout).
        float* redf = reinterpret_cast<float*>(smem);              // [sets * 2 * TCO][256]; the input tile is dead
        const int lane = t & 63, wave = t >> 6;
        const bool per_sub = PERK && a.stat_per_sub;
        if (PERK && !per_sub) {
#pragma unroll
            for (int j = 1; j < NS; ++j)
#pragma unroll
                for (int c = 0; c < TCO; ++c) { ssum[0][c] += ssum[j][c]; ssq[0][c] += ssq[j][c]; }
        }
        const int sets = per_sub ? NS : 1;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (j < sets) {
#pragma unroll
                for (int c = 0; c < TCO; ++c) { redf[((j * 2) * TCO + c) * 256 + t] = ssum[j][c]; redf[((j * 2 + 1) * TCO + c) * 256 + t] = ssq[j][c]; }
            }
        __syncthreads();
        const int rows = per_sub ? a.total_subtiles : (int)gridDim.x;
        for (int q = wave; q < sets * 2 * TCO; q += 4) {
            const float* rq = redf + q * 256 + lane;
            const float v = dcv_wave_sum((rq[0] + rq[64]) + (rq[128] + rq[192]));
            const int j = q / (2 * TCO), qq = q - j * 2 * TCO;
            const int c = qq < TCO ? qq : qq - TCO;
            const int row = per_sub ? (int)blockIdx.x * a.nsub + j : (int)blockIdx.x;
            if (lane == 0 && co0 + c < a.Cout && row < rows) stats[((size_t)(qq < TCO ? 0 : 1) * a.Cout + co0 + c) * rows + row] = (double)v;   // [2][Cout][rows]
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

// ------------------------------------------------------------------------------------------------ LDS-tiled weight gradient
// One block walks tiles of TH output rows x Wo columns of one image: the zero-padded input halo tile [XH][XW][Cin] and the dy tile [TH*Wo][TCO]
// are staged in LDS with coalesced 16-B loads (padding resolved while staging: no bounds checks in the inner loop), then every thread owns one
// "unit" = (VEC input channels, tap r, tap s) x ALL TCO output channels of the tile and sweeps the tile's pixels from LDS: per pixel one
// ds_read_b128 of x, TCO/4 wave-broadcast ds_read_b128 of dy and TCO*VEC FMAs.  Units fewer than 256 are replicated over `nsub` pixel sub-streams.
// Accumulators stay in registers across all tiles of the block; sub-streams are merged through LDS in a fixed order and ONE partial slab per block
// goes to the ordered second stage (wgrad_reduce_kernel) -> deterministic.
struct WgTiledArgs {
    int N, H, W, Cin, Ho, Wo, Cout, R, S, sh, sw, ph, pw, dh, dw;
    int x_nchw;
    int TH, XH, XW;                // tile height (output rows), halo tile dims (input rows / cols incl. padding)
    int units, civ, nsub;          // units = civ * R * S ; civ = Cin / VEC ; nsub = 256 / units
    int tiles_per_img, total_tiles;
    size_t wtotal;
};

template <int TCO, int VEC>
__global__ __launch_bounds__(256) void conv_wgrad_tiled_kernel(WgTiledArgs a, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);                       // [XH][XW][Cin]
    float* dys = xs + (size_t)a.XH * a.XW * a.Cin;                    // [TH*Wo][TCO]
    const int t = threadIdx.x;
    const int ul = t % a.units, sub = t / a.units;
    const bool active = sub < a.nsub;
    const int s = ul % a.S, r = (ul / a.S) % a.R, civ = ul / (a.S * a.R);
    const int ci0 = civ * VEC;
    const int xbase = ((r * a.dh) * a.XW + s * a.dw) * a.Cin + ci0;
    const int co0 = blockIdx.y * TCO;

    float acc[TCO][VEC];
#pragma unroll
    for (int j = 0; j < TCO; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[j][v] = 0.f;

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int n = tile / a.tiles_per_img, ho0 = (tile % a.tiles_per_img) * a.TH;
        const int th = (a.Ho - ho0 < a.TH) ? (a.Ho - ho0) : a.TH;
        const int hi0 = ho0 * a.sh - a.ph, wi0 = -a.pw;
        __syncthreads();
        if (!a.x_nchw) {
            if constexpr (VEC == 4) {
                const int cv = a.Cin / 4, tot = a.XH * a.XW * cv;
                for (int i = t; i < tot; i += 256) {
                    const int c4 = i % cv; int q = i / cv; const int xc = q % a.XW, xr = q / a.XW;
                    const int hi = hi0 + xr, wi = wi0 + xc;
                    const bool ok = hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                    const int hic = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi), wic = wi < 0 ? 0 : (wi >= a.W ? a.W - 1 : wi);
                    float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * a.H + hic) * a.W + wic) * a.Cin + c4 * 4);   // unconditional, zeroed below
                    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(xs + ((size_t)xr * a.XW + xc) * a.Cin + c4 * 4) = v;
                }
            } else {
                const int tot = a.XH * a.XW * a.Cin;
                for (int i = t; i < tot; i += 256) {
                    const int c = i % a.Cin; int q = i / a.Cin; const int xc = q % a.XW, xr = q / a.XW;
                    const int hi = hi0 + xr, wi = wi0 + xc;
                    const bool ok = hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                    const int hic = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi), wic = wi < 0 ? 0 : (wi >= a.W ? a.W - 1 : wi);
                    const float v = x[(((size_t)n * a.H + hic) * a.W + wic) * a.Cin + c];
                    xs[i] = ok ? v : 0.f;
                }
            }
        } else {
            const int tot = a.Cin * a.XH * a.XW;                      // column index fastest: coalesced reads of the NCHW planes
            for (int i = t; i < tot; i += 256) {
                const int xc = i % a.XW; int q = i / a.XW; const int xr = q % a.XH, c = q / a.XH;
                const int hi = hi0 + xr, wi = wi0 + xc;
                const bool ok = hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                const int hic = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi), wic = wi < 0 ? 0 : (wi >= a.W ? a.W - 1 : wi);
                const float v = x[(((size_t)n * a.Cin + c) * a.H + hic) * a.W + wic];
                xs[((size_t)xr * a.XW + xc) * a.Cin + c] = ok ? v : 0.f;
            }
        }
        {
            const int npx = a.TH * a.Wo, tot = npx * (TCO / 4);
            const bool full = (a.Cout % 4 == 0);
            for (int i = t; i < tot; i += 256) {
                const int c4 = i % (TCO / 4), p = i / (TCO / 4);
                const int hl = p / a.Wo, wl = p - hl * a.Wo;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int co = co0 + c4 * 4;
                if (hl < th && co < a.Cout) {
                    const float* src = dy + (((size_t)n * a.Ho + ho0 + hl) * a.Wo + wl) * a.Cout + co;
                    if (full) v = *reinterpret_cast<const float4*>(src);
                    else { v.x = src[0]; if (co + 1 < a.Cout) v.y = src[1]; if (co + 2 < a.Cout) v.z = src[2]; if (co + 3 < a.Cout) v.w = src[3]; }
                }
                *reinterpret_cast<float4*>(dys + (size_t)p * TCO + c4 * 4) = v;
            }
        }
        __syncthreads();
        if (active) {
            const int npx = th * a.Wo;
            for (int p = sub; p < npx; p += a.nsub) {
                const int hl = p / a.Wo, wl = p - hl * a.Wo;
                const float* xp = xs + xbase + ((hl * a.sh) * a.XW + wl * a.sw) * a.Cin;
                float xv[VEC];
                if constexpr (VEC == 4) { const float4 v = *reinterpret_cast<const float4*>(xp); xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w; }
                else xv[0] = xp[0];
                const float* gp = dys + (size_t)p * TCO;
#pragma unroll
                for (int j = 0; j < TCO; j += 4) {
                    const float4 g = *reinterpret_cast<const float4*>(gp + j);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        acc[j][v] = fmaf(g.x, xv[v], acc[j][v]); acc[j + 1][v] = fmaf(g.y, xv[v], acc[j + 1][v]);
                        acc[j + 2][v] = fmaf(g.z, xv[v], acc[j + 2][v]); acc[j + 3][v] = fmaf(g.w, xv[v], acc[j + 3][v]);
                    }
                }
            }
        }
    }
    // merge the pixel sub-streams (fixed order) one output channel at a time through LDS, then write this block's partial slab
    float* red = reinterpret_cast<float*>(smem);                      // 256 * VEC floats (tile buffers are dead now)
    float* dst = partial + (size_t)blockIdx.x * a.wtotal;
#pragma unroll
    for (int j = 0; j < TCO; ++j) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[v * 256 + t] = acc[j][v];
        __syncthreads();
        const int co = co0 + j;
        if (sub == 0 && co < a.Cout) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float sum = 0.f;
                for (int k = 0; k < a.nsub; ++k) sum += red[v * 256 + k * a.units + ul];
                dst[(((size_t)co * a.Cin + ci0 + v) * a.R + r) * a.S + s] = sum;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ sliding-window weight gradient (stride_w in {1, 2}, dil_w = 1)
// Same tiling as above (zero-padded x halo tile + dy tile of TH output rows in LDS, one partial slab per block, ordered second stage), but a thread owns
// a unit = (VEC input channels, filter ROW r, 4 output channels) with ALL S taps of that row, and walks SEG consecutive output pixels of one image row at
// a time: the SEG+S-1 input pixels under the window and the SEG dy vectors are fetched with one ds_read each and feed SEG*S*4*VEC FMAs from registers
// (4->4 5x5: 12 LDS reads per 320 FMAs, against 2 per 16 for the one-tap-per-thread kernel, whose loop was LDS-latency and address-arithmetic bound).
// Units fewer than 256 are replicated over `nsub` segment sub-streams; consecutive sub-streams work on consecutive tile ROWS and both tiles have a row
// stride of an odd number of 16-B slots, so the ds_read_b128 of neighbouring lanes fall into different banks.  Sub-streams are merged through LDS in a
// fixed order (row stride congruent to `units` mod 32: conflict-free for the writers and for the readers) -> deterministic.
// filter columns merged per pass of the sub-stream merge: 2 -> 36 rows (37.6 KB of LDS, three resident blocks per CU); 3 -> 52 rows, one pass less but only
// two resident blocks (measured: 0.4388 vs 0.4524 ms per stock step)
#ifndef DCV_SLIDE_SG
#define DCV_SLIDE_SG 2
#endif
struct WgSlideArgs {
    int N, H, W, Cin, Ho, Wo, Cout, R, sh, ph, pw, dh;
    int x_nchw;
    int TH, XH, XW, WoP;           // output rows per tile, halo tile rows / cols, tile width rounded up to a multiple of SEG
    int xrs, drs;                  // LDS row strides (floats) of the x and dy tiles
    int civ, cogb, units, nsub;    // input-channel vectors, 4-wide output-channel groups per block, units = civ * R * cogb, sub-streams = 256 / units
    int spr;                       // segments per tile row
    int xs_floats;                 // size of the x tile
    int cv_shift, cogb_shift;      // log2 of Cin/VEC and of cogb when they are powers of two, else -1
    int red_rs;                    // row stride of the merge buffer
    int tiles_per_img, total_tiles;
    int nt;                        // threads per block
    int fuse_db;                   // also accumulate the bias gradient (column sums of dy) into slab entries [wtotal, wtotal + Cout)
    size_t wtotal, slab;           // weight-gradient entries, entries per partial slab (wtotal [+ Cout])
};

__device__ __forceinline__ void dcv_divmod(int j, int d, int shift, int& q, int& r) {
    if (shift >= 0) { q = j >> shift; r = j & (d - 1); } else { q = j / d; r = j - q * d; }
}

// `bx`, `by`, `nbx`: the block's coordinates inside ITS problem (blockIdx / gridDim of a single-problem launch; a range of a multi-problem launch)
template <int S, int VEC, int SEG, int SW>
__device__ __forceinline__ void conv_wgrad_slide_body(const WgSlideArgs& a, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
                                                      const int bx, const int by, const int nbx, char* smem) {
    float* xs = reinterpret_cast<float*>(smem);                       // [XH][xrs]   (pixel-major, channel fastest)
    float* dys = xs + a.xs_floats;                                    // [TH][drs]   (pixel-major, 4 * cogb channels)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
    const int ul = t % a.units, sub = t / a.units;
    const bool active = sub < a.nsub;
    const int cog = ul % a.cogb, r = (ul / a.cogb) % a.R, civ = ul / (a.cogb * a.R);   // cog fastest: neighbouring lanes share the x address (broadcast)
    const int ci0 = civ * VEC;
    const int tcob = 4 * a.cogb;
    const int co_base = by * tcob;

    float acc[S][4][VEC];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[s][j][v] = 0.f;
    float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool db_owner = a.fuse_db && ul < a.cogb;               // the (r = 0, first input-channel vector) unit of each output-channel group

    for (int tile = bx; tile < a.total_tiles; tile += nbx) {
        const int n = tile / a.tiles_per_img, ho0 = (tile % a.tiles_per_img) * a.TH;
        const int th = (a.Ho - ho0 < a.TH) ? (a.Ho - ho0) : a.TH;
        const int hi0 = ho0 * a.sh - a.ph, wi0 = -a.pw;
        __syncthreads();
        // ---- stage the x halo tile: a wave takes tile rows wave, wave + nw, ...; lanes run along the row (coalesced); four rows are loaded before any
        //      of them is written to LDS so that their global-memory latencies overlap; padding is resolved here (clamped address + select)
        if (!a.x_nchw) {
            const int cv = a.Cin / VEC, per_row = a.XW * cv;
            for (int j = lane; j < per_row; j += 64) {
                int xc, c; dcv_divmod(j, cv, a.cv_shift, xc, c);
                const int wi = wi0 + xc;
                const bool wok = wi >= 0 && wi < a.W;
                const int wic = wi < 0 ? 0 : (wi >= a.W ? a.W - 1 : wi);
                const float* col = x + (size_t)n * a.H * a.W * a.Cin + (size_t)wic * a.Cin + c * VEC;
                for (int xr0 = wave; xr0 < a.XH; xr0 += 4 * nw) {
                    float vv[4][VEC];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int hi = hi0 + xr0 + u * nw;
                        const int hic = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi);
                        const float* sp = col + (size_t)hic * a.W * a.Cin;
                        if constexpr (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(sp); vv[u][0] = q.x; vv[u][1] = q.y; vv[u][2] = q.z; vv[u][3] = q.w; }
                        else vv[u][0] = sp[0];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int xr = xr0 + u * nw, hi = hi0 + xr;
                        const bool ok = wok && hi >= 0 && hi < a.H;
                        if (xr < a.XH) {
                            float* dp = xs + (size_t)xr * a.xrs + j * VEC;
                            if constexpr (VEC == 4) *reinterpret_cast<float4*>(dp) = ok ? make_float4(vv[u][0], vv[u][1], vv[u][2], vv[u][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                            else dp[0] = ok ? vv[u][0] : 0.f;
                        }
                    }
                }
            }
        } else {
            const int pairs = a.Cin * a.XH;                           // (channel, tile row) pairs; lanes along the NCHW row
            for (int xc = lane; xc < a.XW; xc += 64) {
                const int wi = wi0 + xc;
                const bool wok = wi >= 0 && wi < a.W;
                const int wic = wi < 0 ? 0 : (wi >= a.W ? a.W - 1 : wi);
                for (int q0 = wave; q0 < pairs; q0 += 4 * nw) {
                    float vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int q = q0 + u * nw; q = q < pairs ? q : pairs - 1;
                        const int c = q / a.XH, xr = q - c * a.XH;
                        const int hi = hi0 + xr;
                        const int hic = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi);
                        vv[u] = x[(((size_t)n * a.Cin + c) * a.H + hic) * a.W + wic];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = q0 + u * nw;
                        if (q < pairs) {
                            const int c = q / a.XH, xr = q - c * a.XH;
                            const int hi = hi0 + xr;
                            const bool ok = wok && hi >= 0 && hi < a.H;
                            xs[(size_t)xr * a.xrs + xc * a.Cin + c] = ok ? vv[u] : 0.f;
                        }
                    }
                }
            }
        }
        // ---- stage the dy tile (zero beyond Wo, beyond the last row of the image and beyond Cout), same four-rows-in-flight scheme
        {
            const int per_row = a.WoP * a.cogb;
            const bool full = (a.Cout % 4 == 0);
            for (int j = lane; j < per_row; j += 64) {
                int wl, c4; dcv_divmod(j, a.cogb, a.cogb_shift, wl, c4);
                const int co = co_base + c4 * 4;
                const bool cok = wl < a.Wo && co < a.Cout;
                const float* col = dy + ((size_t)n * a.Ho + ho0) * a.Wo * a.Cout + (size_t)(wl < a.Wo ? wl : 0) * a.Cout + (co < a.Cout ? co : 0);
                for (int h0 = wave; h0 < a.TH; h0 += 4 * nw) {
                    float4 vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int hl = h0 + u * nw;
                        const float* sp = col + (size_t)(hl < th ? hl : 0) * a.Wo * a.Cout;
                        if (full) vv[u] = *reinterpret_cast<const float4*>(sp);
                        else {
                            vv[u] = make_float4(sp[0], 0.f, 0.f, 0.f);
                            if (co + 1 < a.Cout) vv[u].y = sp[1];
                            if (co + 2 < a.Cout) vv[u].z = sp[2];
                            if (co + 3 < a.Cout) vv[u].w = sp[3];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int hl = h0 + u * nw;
                        if (hl < a.TH) *reinterpret_cast<float4*>(dys + (size_t)hl * a.drs + j * 4) = (cok && hl < th) ? vv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
        __syncthreads();
        if (active) {
            // segment index = sg * th + hl (row fastest): consecutive sub-streams sit on consecutive rows; rows beyond th hold zero dy and are skipped
            const int nsegs = th * a.spr;
            int sg = sub / th, hl = sub - sg * th;
            const int dsg = a.nsub / th, dhl = a.nsub - dsg * th;
            for (int seg = sub; seg < nsegs; seg += a.nsub) {
                const int w0 = sg * SEG;
                const float* xp = xs + (size_t)(hl * a.sh + r * a.dh) * a.xrs + w0 * SW * a.Cin + ci0;
                const float* gp = dys + (size_t)hl * a.drs + w0 * tcob + 4 * cog;
                constexpr int XWN = (SEG - 1) * SW + S;                // input pixels under the window of SEG outputs
                float xw[XWN][VEC];
                float4 g[SEG];
#pragma unroll
                for (int k = 0; k < XWN; ++k) {
                    if constexpr (VEC == 4) {
                        const float4 v = *reinterpret_cast<const float4*>(xp + k * a.Cin);
                        xw[k][0] = v.x; xw[k][1] = v.y; xw[k][2] = v.z; xw[k][3] = v.w;
                    } else {
                        xw[k][0] = xp[k * a.Cin];
                    }
                }
#pragma unroll
                for (int p = 0; p < SEG; ++p) g[p] = *reinterpret_cast<const float4*>(gp + p * tcob);
                if (db_owner) {
#pragma unroll
                    for (int p = 0; p < SEG; ++p) { dbacc.x += g[p].x; dbacc.y += g[p].y; dbacc.z += g[p].z; dbacc.w += g[p].w; }
                }
#pragma unroll
                for (int p = 0; p < SEG; ++p)
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            acc[s][0][v] = fmaf(g[p].x, xw[p * SW + s][v], acc[s][0][v]);
                            acc[s][1][v] = fmaf(g[p].y, xw[p * SW + s][v], acc[s][1][v]);
                            acc[s][2][v] = fmaf(g[p].z, xw[p * SW + s][v], acc[s][2][v]);
                            acc[s][3][v] = fmaf(g[p].w, xw[p * SW + s][v], acc[s][3][v]);
                        }
                sg += dsg; hl += dhl;
                if (hl >= th) { hl -= th; ++sg; }
            }
        }
    }
    // ---- merge the sub-streams in a fixed order through red[rows][red_rs], SG filter columns (SG * 4 * VEC rows) per pass plus, in the last pass, the 4
    //      bias-gradient rows: thread (sub, unit) sums rows e = sub, sub + nsub, ... of ITS unit over all sub-streams (no index arithmetic: the unit's
    //      coordinates are already in registers).  With few units (4->4 5x5: 5 units, 51 sub-streams) every thread owns at most one row per pass; the
    //      sub-stream sum runs as four interleaved chains so that its LDS reads are in flight together (the merge used to cost more than the FMA loop).
    float* red = reinterpret_cast<float*>(smem);
    float* dst = partial + (size_t)bx * a.slab;
    constexpr int E = 4 * VEC;
    constexpr int SG = (VEC == 4) ? (S < DCV_SLIDE_SG ? S : DCV_SLIDE_SG) : S;   // filter columns per pass (rows <= 16 * DCV_SLIDE_SG + 4: see wgrad_slide_plan's LDS size)
    constexpr int PASSES = (S + SG - 1) / SG;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int s0 = ps * SG;
        const int ns = (S - s0 < SG) ? (S - s0) : SG;
        const int rows_w = ns * E;
        const bool with_db = a.fuse_db && ps == PASSES - 1;
        const int rows = rows_w + (with_db ? 4 : 0);
        __syncthreads();
        if (active) {
#pragma unroll
            for (int sl = 0; sl < SG; ++sl) {
                if (s0 + sl < S) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) red[(sl * E + j * VEC + v) * a.red_rs + t] = acc[(s0 + sl) < S ? (s0 + sl) : 0][j][v];
                }
            }
            if (with_db) { red[(rows_w + 0) * a.red_rs + t] = dbacc.x; red[(rows_w + 1) * a.red_rs + t] = dbacc.y; red[(rows_w + 2) * a.red_rs + t] = dbacc.z; red[(rows_w + 3) * a.red_rs + t] = dbacc.w; }
        }
        __syncthreads();
        if (active) {
            for (int e = sub; e < rows; e += a.nsub) {
                if (e >= rows_w && ul >= a.cogb) continue;                // bias rows: only the owner units hold column sums of dy
                const float* q = red + e * a.red_rs + ul;                 // bank(e * red_rs + ul) = (e * units + ul) mod 32: conflict-free across lanes
                float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
                int k = 0;
                for (; k + 3 < a.nsub; k += 4) { p0 += q[0]; p1 += q[a.units]; p2 += q[2 * a.units]; p3 += q[3 * a.units]; q += 4 * a.units; }
                for (; k < a.nsub; ++k) { p0 += *q; q += a.units; }
                const float sum = (p0 + p1) + (p2 + p3);
                if (e < rows_w) {
                    const int sl = e / E, rem = e - sl * E;
                    const int co = co_base + 4 * cog + rem / VEC, ci = ci0 + rem % VEC;
                    if (co < a.Cout) dst[(((size_t)co * a.Cin + ci) * a.R + r) * S + s0 + sl] = sum;
                } else {
                    const int co = co_base + 4 * cog + (e - rows_w);
                    if (co < a.Cout) dst[a.wtotal + co] = sum;
                }
            }
        }
    }
}

template <int S, int VEC, int SEG, int SW>
__global__ __launch_bounds__(256) void conv_wgrad_slide_kernel(WgSlideArgs a, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_wgrad_slide_body<S, VEC, SEG, SW>(a, x, dy, partial, blockIdx.x, blockIdx.y, gridDim.x, smem);
}

// Several layers' weight gradients in ONE launch (the training step defers them to the end of its backward pass, deepcv_amd/slabs.py): block ranges
// [start[i], start[i + 1]) run problem i.  One launch floor instead of one per layer, and the staging / FMA / merge phases of different layers' blocks
// overlap on the CUs.  Variants: stride-1 kernels of 3 and 5 taps (the reference's stock layers), NHWC vector or scalar input.
#define WG_MULTI_MAX 8
struct WgMultiTable {
    int count;
    int start[WG_MULTI_MAX + 1];
    int gx[WG_MULTI_MAX];
    int variant[WG_MULTI_MAX];     // 0: <3, 4>, 1: <5, 4>, 2: <3, 1>, 3: <5, 1>
    const float* x[WG_MULTI_MAX];
    const float* dy[WG_MULTI_MAX];
    float* partial[WG_MULTI_MAX];
    WgSlideArgs a[WG_MULTI_MAX];
};

__global__ __launch_bounds__(256) void conv_wgrad_slide_multi_kernel(WgMultiTable t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int i = 0;
    while (i + 1 < t.count && (int)blockIdx.x >= t.start[i + 1]) ++i;                  // block-uniform: scalar loop
    const int local = blockIdx.x - t.start[i];
    const int gx = t.gx[i], by = local / gx, bx = local - by * gx;
    const WgSlideArgs& a = t.a[i];
    switch (t.variant[i]) {
        case 0: conv_wgrad_slide_body<3, 4, 4, 1>(a, t.x[i], t.dy[i], t.partial[i], bx, by, gx, smem); break;
        case 1: conv_wgrad_slide_body<5, 4, 4, 1>(a, t.x[i], t.dy[i], t.partial[i], bx, by, gx, smem); break;
        case 2: conv_wgrad_slide_body<3, 1, 4, 1>(a, t.x[i], t.dy[i], t.partial[i], bx, by, gx, smem); break;
        default: conv_wgrad_slide_body<5, 1, 4, 1>(a, t.x[i], t.dy[i], t.partial[i], bx, by, gx, smem); break;
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
int launch_generic(const DirectArgs& a, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    const size_t total = (size_t)a.N * a.Hout * a.Wout;
    dim3 grid((unsigned)dcv_ceil_div(total, 256), (unsigned)dcv_ceil_div(a.Cout, TCO));
    const size_t smem = (size_t)a.R * a.S * a.cin_chunk * TCO * sizeof(float);
    hipLaunchKernelGGL((conv_generic_kernel<TCO, IN_MODE>), grid, dim3(256), smem, st, a, x, w, bias, y);
    return DCV_OK;
}

template <int IN_MODE>
int dispatch_generic(int tco, const DirectArgs& a, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    switch (tco) {
        case 4: return launch_generic<4, IN_MODE>(a, x, w, bias, y, st);
        case 8: return launch_generic<8, IN_MODE>(a, x, w, bias, y, st);
        default: return launch_generic<16, IN_MODE>(a, x, w, bias, y, st);
    }
}

// Blocks the tiled launch should have at least (4 per CU): the kernel alternates a staging phase and a compute phase, so a CU needs several resident
// blocks to overlap them; with fewer, the output channels are split over more, narrower tiles (blockIdx.y).  DCV_CONV_TARGET_BLOCKS overrides.
inline long conv_target_blocks() {
    static long v = -1;
    if (v < 0) { const char* e = getenv("DCV_CONV_TARGET_BLOCKS"); v = e ? atol(e) : 1024; }
    return v;
}

#define TILE_LDS_BYTES (48 * 1024)

bool tile_plan_ppt(const DirectArgs& a, int ppt, TileArgs* t) {
    const int px = 256 * ppt;
    t->N = a.N; t->Hin = a.Hin; t->Win = a.Win; t->Cin = a.Cin; t->Hout = a.Hout; t->Wout = a.Wout; t->Cout = a.Cout;
    t->R = a.R; t->S = a.S; t->sh = a.sh; t->sw = a.sw; t->ph = a.ph; t->pw = a.pw; t->dh = a.dh; t->dw = a.dw;
    t->in_nchw = a.in_nchw; t->act = a.act; t->slope = a.slope;
    t->cin_pad = (int)dcv_align_up(a.Cin, 4);
    int th = px / a.Wout; if (th > a.Hout) th = a.Hout;
    if (th < 1) return false;
    t->TH = th;
    t->XH = (th - 1) * a.sh + (a.R - 1) * a.dh + 1;
    t->XW = (a.Wout - 1) * a.sw + (a.S - 1) * a.dw + 1;
    t->tiles_per_img = (a.Hout + th - 1) / th;
    t->nsub = (th == a.Hout) ? px / (th * a.Wout) : 1;
    if (t->nsub < 1) t->nsub = 1;
    if ((long)t->nsub > (long)a.N) t->nsub = a.N;
    t->total_subtiles = a.N * t->tiles_per_img;
    const size_t per_c4 = (size_t)t->nsub * t->XH * t->XW * 16;     // bytes per channel-vector plane
    if (per_c4 > TILE_LDS_BYTES) return false;
    int cc = (int)(TILE_LDS_BYTES / per_c4) * 4;
    if (cc > t->cin_pad) cc = t->cin_pad;
    t->CC = cc;
    return true;
}

// Plans conv_tile_kernel for the kernel-view problem `a` (forward: as is; stride-1 dgrad: flipped filter).  False -> use the generic kernel.
// Among (pixels per thread, output-channel tile) it keeps the largest per-thread work which still yields DCV_CONV_TARGET_BLOCKS blocks.
bool tile_plan(const DirectArgs& a, TileArgs* t, int* tco_out, int* ppt_out) {
    if (a.dgrad == 2) return false;
    const int tco_max = pick_tco(a.Cout);
    static const int cand[9][2] = {{4, 16}, {4, 8}, {2, 16}, {4, 4}, {2, 8}, {1, 16}, {2, 4}, {1, 8}, {1, 4}};   // {PPT, TCO}, by PPT*TCO then PPT
    int best = -1; TileArgs bt{};
    for (int i = 0; i < 9; ++i) {
        const int ppt = cand[i][0], tco = cand[i][1];
        if (tco > tco_max) continue;
        TileArgs c{};
        if (!tile_plan_ppt(a, ppt, &c)) continue;
        const long blocks = (long)dcv_ceil_div(c.total_subtiles, c.nsub) * (long)dcv_ceil_div(a.Cout, tco);
        best = i; bt = c;                                        // keep shrinking work per thread while the grid is too small
        if (blocks >= conv_target_blocks()) break;
    }
    if (best < 0) return false;
    *t = bt; *tco_out = cand[best][1]; *ppt_out = cand[best][0];
    // several whole images per block, one per pixel slot of a thread: the statistics epilogue can emit one row per image
    t->stat_per_sub = (t->nsub > 1 && t->nsub == *ppt_out && t->tiles_per_img == 1 && DCV_TILE_PERK(*ppt_out, *tco_out)) ? 1 : 0;
    return true;
}

int run_direct(DirectArgs a, const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, hipStream_t st, bool prepared, double* stats = nullptr,
               const float* addend = nullptr) {
    TileArgs t{}; int tco, ppt;
    if (tile_plan(a, &t, &tco, &ppt)) {
        const int tiles = (int)dcv_ceil_div(a.Cout, tco);
        const size_t need = (size_t)tiles * a.R * a.S * t.cin_pad * tco * sizeof(float);
        if (!ws || ws_bytes < need) return DCV_ERR_WORKSPACE;
        float* wt = reinterpret_cast<float*>(ws);
        const int total = tiles * a.R * a.S * t.cin_pad * tco;
        const int pblocks = (total + 255) / 256 > 128 ? 128 : (total + 255) / 256;
        if (!prepared) hipLaunchKernelGGL(prep_weights_kernel, dim3((unsigned)pblocks), dim3(256), 0, st, a, w, wt, tco, tiles, t.cin_pad, 1);
        dim3 grid((unsigned)dcv_ceil_div(t.total_subtiles, t.nsub), (unsigned)tiles);
        size_t smem = (size_t)t.nsub * t.XH * t.XW * t.CC * sizeof(float);
        const size_t stat_smem = (size_t)(t.stat_per_sub ? ppt : 1) * 2 * tco * 1024;
        if (smem < stat_smem) smem = stat_smem;                        // the statistics epilogue reuses the tile buffer: [sets * 2 * TCO][256] floats
#define DCV_TILE_LAUNCH(TCO_, ST_, PPT_) hipLaunchKernelGGL((conv_tile_kernel<TCO_, ST_, PPT_>), grid, dim3(256), smem, st, t, x, wt, bias, y, stats, addend)
#define DCV_TILE_PPT(TCO_, ST_) switch (ppt) { case 4: DCV_TILE_LAUNCH(TCO_, ST_, 4); break; case 2: DCV_TILE_LAUNCH(TCO_, ST_, 2); break; default: DCV_TILE_LAUNCH(TCO_, ST_, 1); break; }
#define DCV_TILE_SWITCH(TCO_) switch (a.S) { case 1: DCV_TILE_PPT(TCO_, 1); break; case 3: DCV_TILE_PPT(TCO_, 3); break; case 5: DCV_TILE_PPT(TCO_, 5); break; \
                                             case 7: DCV_TILE_PPT(TCO_, 7); break; default: DCV_TILE_PPT(TCO_, 0); break; }
        switch (tco) {
            case 4: DCV_TILE_SWITCH(4); break;
            case 8: DCV_TILE_SWITCH(8); break;
            default: DCV_TILE_SWITCH(16); break;
        }
#undef DCV_TILE_PPT
#undef DCV_TILE_SWITCH
#undef DCV_TILE_LAUNCH
        return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
    }
    // generic kernel: LDS-resident filter slice, one output pixel per thread (strided dgrad gather, very wide rows)
    if (stats != nullptr || addend != nullptr) return DCV_ERR_UNSUPPORTED;   // callers ask dcv_direct_conv2d_fwd_stats_rows / fall back to a separate addition
    if (a.dgrad == 2) a.dgrad = 1;
    const int in_mode = a.in_nchw ? 2 : ((a.Cin % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? 0 : 1);
    const int vec = in_mode == 0 ? 4 : 1;
    int gtco = pick_tco(a.Cout);
    int chunk = pick_cin_chunk(a.Cin, a.R, a.S, gtco, vec);
    while (chunk == 0 && gtco > 4) { gtco /= 2; chunk = pick_cin_chunk(a.Cin, a.R, a.S, gtco, vec); }
    if (chunk == 0) return DCV_ERR_UNSUPPORTED;
    a.cin_chunk = chunk;
    int rc;
    if (in_mode == 0) rc = dispatch_generic<0>(gtco, a, x, w, bias, y, st);
    else if (in_mode == 1) rc = dispatch_generic<1>(gtco, a, x, w, bias, y, st);
    else rc = dispatch_generic<2>(gtco, a, x, w, bias, y, st);
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
size_t dcv_direct_conv2d_fwd_ws(const dcv_conv2d_desc* d) { return dcv_align_up((size_t)d->r * d->s * dcv_align_up(d->cin, 4) * dcv_align_up(d->cout, 16) * sizeof(float), 256); }
size_t dcv_direct_conv2d_dgrad_ws(const dcv_conv2d_desc* d) { return dcv_align_up((size_t)d->r * d->s * dcv_align_up(d->cout, 4) * dcv_align_up(d->cin, 16) * sizeof(float), 256); }

namespace {
inline DirectArgs fwd_args(const dcv_conv2d_desc* d) {
    DirectArgs a{};
    a.N = d->n; a.Hin = d->h; a.Win = d->w; a.Cin = d->cin; a.Hout = d->ho; a.Wout = d->wo; a.Cout = d->cout;
    a.R = d->r; a.S = d->s; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    a.w_cout = d->cout; a.w_cin = d->cin; a.dgrad = 0; a.act = d->act; a.slope = d->act_slope; a.in_nchw = d->x_layout == DCV_LAYOUT_NCHW;
    return a;
}
inline DirectArgs dgrad_args(const dcv_conv2d_desc* d) {
    DirectArgs a{};
    a.N = d->n; a.Hin = d->ho; a.Win = d->wo; a.Cin = d->cout; a.Hout = d->h; a.Wout = d->w; a.Cout = d->cin;
    a.R = d->r; a.S = d->s; a.dh = d->dil_h; a.dw = d->dil_w;
    a.w_cout = d->cout; a.w_cin = d->cin; a.dgrad = 1; a.act = DCV_ACT_NONE; a.slope = 0.f; a.in_nchw = 0;
    if (d->stride_h == 1 && d->stride_w == 1 && d->r - 1 - d->pad_h >= 0 && d->s - 1 - d->pad_w >= 0 && (d->r - 1) * d->dil_h >= d->pad_h && (d->s - 1) * d->dil_w >= d->pad_w) {
        // stride-1 data gradient == forward convolution of dy with the flipped, transposed filter and pad' = dil*(R-1) - pad
        a.sh = 1; a.sw = 1; a.ph = d->dil_h * (d->r - 1) - d->pad_h; a.pw = d->dil_w * (d->s - 1) - d->pad_w;
    } else {
        // gather form handled by the generic kernel (dgrad addressing: ho = (hi + pad - r*dil) / stride when divisible)
        a.dgrad = 2;
        a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w;
    }
    return a;
}
}  // namespace

int dcv_direct_conv2d_fwd(const dcv_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, hipStream_t st, bool prepared, double* stats) {
    return run_direct(fwd_args(d), x, w, bias, y, ws, ws_bytes, st, prepared, stats);
}

int dcv_direct_conv2d_fwd_stats_rows(const dcv_conv2d_desc* d, int* rows_per_sample) {
    const DirectArgs a = fwd_args(d);
    TileArgs t{}; int tco, ppt;
    if (rows_per_sample) *rows_per_sample = 0;
    if (!tile_plan(a, &t, &tco, &ppt)) return 0;
    if (t.stat_per_sub) { if (rows_per_sample) *rows_per_sample = 1; return t.total_subtiles; }
    if (t.nsub == 1 && rows_per_sample) *rows_per_sample = t.tiles_per_img;     // blocks never straddle images
    return (int)dcv_ceil_div(t.total_subtiles, t.nsub);
}

int dcv_direct_conv2d_dgrad(const dcv_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes, hipStream_t st, bool prepared, const float* addend) {
    return run_direct(dgrad_args(d), dy, w, nullptr, dx, ws, ws_bytes, st, prepared, nullptr, addend);
}

int dcv_direct_conv2d_prep_entries(const dcv_conv2d_desc* d, int op, dcv_prep_entry* e) {
    const DirectArgs a = op == 0 ? fwd_args(d) : dgrad_args(d);
    TileArgs t{}; int tco, ppt;
    if (!tile_plan(a, &t, &tco, &ppt)) return 0;                  // generic kernel: stages the OIHW filter itself
    *e = dcv_prep_entry{};
    e->kind = 0; e->kcout = a.Cout; e->kcin = a.Cin; e->R = a.R; e->S = a.S; e->w_cin = a.w_cin; e->dgrad = a.dgrad ? 1 : 0;
    e->tco = tco; e->tiles = (int)dcv_ceil_div(a.Cout, tco); e->cin_pad = t.cin_pad;
    e->total = e->tiles * a.R * a.S * t.cin_pad * tco;
    return 1;
}

namespace {

// Plans the LDS-tiled weight gradient; returns false when the layer does not fit it (too many units / tile larger than 48 KiB of LDS).
bool wgrad_tiled_plan(const dcv_conv2d_desc* d, WgTiledArgs* a, int* tco, int* vec, int* blocks, size_t* smem) {
    const int v = (d->x_layout == DCV_LAYOUT_NHWC && d->cin % 4 == 0) ? 4 : 1;
    const int civ = d->cin / v;
    const long units = (long)civ * d->r * d->s;
    if (units > 256) return false;
    const int tc = pick_tco(d->cout);
    a->N = d->n; a->H = d->h; a->W = d->w; a->Cin = d->cin; a->Ho = d->ho; a->Wo = d->wo; a->Cout = d->cout; a->R = d->r; a->S = d->s;
    a->sh = d->stride_h; a->sw = d->stride_w; a->ph = d->pad_h; a->pw = d->pad_w; a->dh = d->dil_h; a->dw = d->dil_w;
    a->x_nchw = d->x_layout == DCV_LAYOUT_NCHW;
    a->XW = (d->wo - 1) * d->stride_w + (d->s - 1) * d->dil_w + 1;
    int th = 256 / d->wo; if (th < 1) th = 1; if (th > d->ho) th = d->ho;
    size_t bytes = 0;
    for (;;) {
        a->TH = th; a->XH = (th - 1) * d->stride_h + (d->r - 1) * d->dil_h + 1;
        bytes = ((size_t)a->XH * a->XW * d->cin + (size_t)th * d->wo * tc) * sizeof(float);
        if (bytes <= 48 * 1024 || th == 1) break;
        th = th / 2;
    }
    if (bytes > 48 * 1024) return false;
    if (bytes < (size_t)256 * v * sizeof(float)) bytes = (size_t)256 * v * sizeof(float);
    a->units = (int)units; a->civ = civ; a->nsub = 256 / (int)units;
    a->tiles_per_img = (d->ho + th - 1) / th;
    a->total_tiles = d->n * a->tiles_per_img;
    a->wtotal = (size_t)d->cout * d->cin * d->r * d->s;
    int nb = a->total_tiles < 512 ? a->total_tiles : 512;
    const size_t max_by_bytes = ((size_t)64 << 20) / (a->wtotal * sizeof(float));
    if ((size_t)nb > max_by_bytes) nb = (int)(max_by_bytes > 0 ? max_by_bytes : 1);
    *tco = tc; *vec = v; *blocks = nb; *smem = bytes;
    return true;
}

// Plans the sliding-window weight gradient; false when the layer does not fit it (strided / dilated columns, unusual filter width, too many units).
// `shared_launch`: the problem runs as a block range of a multi-layer launch, where other layers' blocks fill the CUs: half as many (twice as long)
// blocks mean half as many merges and partial slabs (0.4097 vs 0.4135 ms per stock step); alone, a layer wants two blocks per CU
bool wgrad_slide_plan(const dcv_conv2d_desc* d, bool fuse_db, WgSlideArgs* a, int* vec, int* seg, int* blocks, int* ygrid, size_t* smem, bool shared_launch = false) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_WGRAD_SLIDE"); enabled = e ? atoi(e) : 1; }
    if (!enabled) return false;
    if ((d->stride_w != 1 && d->stride_w != 2) || d->dil_w != 1) return false;
    if (!(d->s == 1 || d->s == 3 || d->s == 5 || d->s == 7)) return false;
    const int v = (d->x_layout == DCV_LAYOUT_NHWC && d->cin % 4 == 0) ? 4 : 1;
    const int civ = d->cin / v;
    static int nt = -1;
    if (nt < 0) { const char* e = getenv("DCV_WGRAD_THREADS"); nt = e ? atoi(e) : 256; }
    if ((long)civ * d->r > nt) return false;
    const int cogs = (d->cout + 3) / 4;
    int cogb = nt / (civ * d->r); if (cogb > cogs) cogb = cogs;
    a->nt = nt;
    const int sg = (d->s == 7 && v == 4) ? 2 : 4;
    a->N = d->n; a->H = d->h; a->W = d->w; a->Cin = d->cin; a->Ho = d->ho; a->Wo = d->wo; a->Cout = d->cout; a->R = d->r;
    a->sh = d->stride_h; a->ph = d->pad_h; a->pw = d->pad_w; a->dh = d->dil_h;
    a->x_nchw = d->x_layout == DCV_LAYOUT_NCHW;
    a->WoP = (d->wo + sg - 1) / sg * sg;
    a->XW = (a->WoP - 1) * d->stride_w + d->s;
    a->spr = a->WoP / sg;
    a->civ = civ; a->cogb = cogb; a->units = civ * d->r * cogb; a->nsub = nt / a->units;
    auto log2_or_neg = [](int n) { int sft = 0; while ((1 << sft) < n) ++sft; return (1 << sft) == n ? sft : -1; };
    a->cv_shift = log2_or_neg(civ); a->cogb_shift = log2_or_neg(cogb);
    // LDS row strides: an odd number of 16-B slots (VEC = 4) / an odd number of words (VEC = 1)
    a->xrs = a->XW * d->cin;
    if (v == 4) { if ((a->xrs / 4) % 2 == 0) a->xrs += 4; } else { a->xrs = (a->xrs + 3) / 4 * 4; if ((a->xrs / 4) % 2 == 0) a->xrs += 4; }
    a->drs = a->WoP * 4 * cogb; if ((a->drs / 4) % 2 == 0) a->drs += 4;
    a->red_rs = nt + (a->units % 32);                                   // red_rs = units (mod 32)
    static int target_tiles = -1, target_blocks = -1;
    if (target_tiles < 0) { const char* e = getenv("DCV_WGRAD_TILES"); target_tiles = e ? atoi(e) : 512; }
    if (target_blocks < 0) { const char* e = getenv("DCV_WGRAD_BLOCKS"); target_blocks = e ? atoi(e) : 512; }
    int th = d->ho;
    size_t bytes = 0;
    for (;;) {
        a->TH = th; a->XH = (th - 1) * d->stride_h + (d->r - 1) * d->dil_h + 1;
        a->xs_floats = a->XH * a->xrs;
        bytes = ((size_t)a->xs_floats + (size_t)th * a->drs) * sizeof(float);
        const long tiles = (long)d->n * ((d->ho + th - 1) / th);
        if (th == 1 || (bytes <= 40 * 1024 && (tiles >= target_tiles || th <= 4))) break;
        th = (th + 1) / 2;
    }
    if (bytes > 64 * 1024) return false;
    const int sgrp = (v == 4) ? (d->s < DCV_SLIDE_SG ? d->s : DCV_SLIDE_SG) : d->s;             // filter columns merged per pass (SG of the kernel)
    const size_t red_bytes = ((size_t)sgrp * 4 * v + 4) * a->red_rs * sizeof(float);
    if (bytes < red_bytes) bytes = red_bytes;
    a->tiles_per_img = (d->ho + th - 1) / th;
    a->total_tiles = d->n * a->tiles_per_img;
    a->wtotal = (size_t)d->cout * d->cin * d->r * d->s;
    a->fuse_db = fuse_db ? 1 : 0;
    a->slab = a->wtotal + (fuse_db ? (size_t)d->cout : 0);
    const int want_blocks = shared_launch ? (target_blocks + 1) / 2 : target_blocks;
    int nb = a->total_tiles < want_blocks ? a->total_tiles : want_blocks;
    const size_t max_by_bytes = ((size_t)64 << 20) / (a->slab * sizeof(float));
    if ((size_t)nb > max_by_bytes) nb = (int)(max_by_bytes > 0 ? max_by_bytes : 1);
    *vec = v; *seg = sg; *blocks = nb; *ygrid = (cogs + cogb - 1) / cogb; *smem = bytes;
    return true;
}

template <int VEC, int SW>
void launch_wgrad_slide(int s, dim3 grid, size_t smem, hipStream_t st, const WgSlideArgs& a, const float* x, const float* dy, float* partial) {
    switch (s) {
        case 1: hipLaunchKernelGGL((conv_wgrad_slide_kernel<1, VEC, 4, SW>), grid, dim3(a.nt), smem, st, a, x, dy, partial); break;
        case 3: hipLaunchKernelGGL((conv_wgrad_slide_kernel<3, VEC, 4, SW>), grid, dim3(a.nt), smem, st, a, x, dy, partial); break;
        case 5: hipLaunchKernelGGL((conv_wgrad_slide_kernel<5, VEC, 4, SW>), grid, dim3(a.nt), smem, st, a, x, dy, partial); break;
        default: hipLaunchKernelGGL((conv_wgrad_slide_kernel<7, VEC, (VEC == 4 ? 2 : 4), SW>), grid, dim3(a.nt), smem, st, a, x, dy, partial); break;
    }
}

template <int VEC>
void launch_wgrad_tiled(int tco, dim3 grid, size_t smem, hipStream_t st, const WgTiledArgs& a, const float* x, const float* dy, float* partial) {
    switch (tco) {
        case 4: hipLaunchKernelGGL((conv_wgrad_tiled_kernel<4, VEC>), grid, dim3(256), smem, st, a, x, dy, partial); break;
        case 8: hipLaunchKernelGGL((conv_wgrad_tiled_kernel<8, VEC>), grid, dim3(256), smem, st, a, x, dy, partial); break;
        default: hipLaunchKernelGGL((conv_wgrad_tiled_kernel<16, VEC>), grid, dim3(256), smem, st, a, x, dy, partial); break;
    }
}

}  // namespace

size_t dcv_direct_conv2d_wgrad_ws(const dcv_conv2d_desc* d) {
    WgTiledArgs ta{}; int tco, tvec, blocks; size_t smem;
    { WgSlideArgs sa{}; int sv, sseg, sb, sy; size_t ssm;
      if (wgrad_slide_plan(d, true, &sa, &sv, &sseg, &sb, &sy, &ssm)) return dcv_align_up((size_t)sb * sa.slab * sizeof(float), 256); }
    if (wgrad_tiled_plan(d, &ta, &tco, &tvec, &blocks, &smem)) return dcv_align_up((size_t)blocks * ta.wtotal * sizeof(float), 256);
    WgradArgs a{}; int chunks, ub, vec;
    wgrad_plan(d, &a, &chunks, &ub, &vec);
    return dcv_align_up((size_t)chunks * a.wtotal * sizeof(float), 256);
}

bool dcv_direct_conv2d_wgrad_fuses_bias(const dcv_conv2d_desc* d) {
    WgSlideArgs sa{}; int sv, sseg, sb, sy; size_t ssm;
    return wgrad_slide_plan(d, true, &sa, &sv, &sseg, &sb, &sy, &ssm);
}

// `db` (nullable): bias gradient; *db_done tells the caller whether this back-end produced it
// Stage one of every job the multi-problem slide kernel serves (taken[i] = true), WG_MULTI_MAX problems per launch; their pending sums are appended to
// `segs` (the caller reduces them, dcv_slab_reduce).  Jobs it leaves (taken[i] = false) go through dcv_direct_conv2d_wgrad one by one.
int dcv_direct_conv2d_wgrad_multi(int count, const dcv_wgrad_job* jobs, const bool* eligible, bool* taken, dcv_slab_segment* segs, int* nsegs, hipStream_t st) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_WGRAD_MULTI"); enabled = e ? atoi(e) : 1; }
    int candidates = 0;                                                   // layers the slide kernel serves: two or more share a launch
    for (int i = 0; i < count; ++i) {
        if (!enabled || !eligible[i]) continue;
        WgSlideArgs sa{}; int sv, sseg, sb, sy; size_t ssm;
        const dcv_conv2d_desc* d = &jobs[i].desc;
        if (wgrad_slide_plan(d, jobs[i].db != nullptr, &sa, &sv, &sseg, &sb, &sy, &ssm) && d->stride_w == 1 && (d->s == 3 || d->s == 5) && sa.nt == 256 && sseg == 4) ++candidates;
    }
    const bool shared = candidates >= 2;
    WgMultiTable t{};
    size_t smem_max = 0;
    int blocks = 0;
    auto launch = [&]() {
        if (t.count == 0) return;
        t.start[t.count] = blocks;
        hipLaunchKernelGGL(conv_wgrad_slide_multi_kernel, dim3((unsigned)blocks), dim3(256), smem_max, st, t);
        t = WgMultiTable{}; smem_max = 0; blocks = 0;
    };
    for (int i = 0; i < count; ++i) {
        taken[i] = false;
        if (!enabled || !eligible[i]) continue;
        const dcv_wgrad_job& j = jobs[i];
        const dcv_conv2d_desc* d = &j.desc;
        WgSlideArgs sa{}; int sv, sseg, sb, sy; size_t ssm;
        if (!wgrad_slide_plan(d, j.db != nullptr, &sa, &sv, &sseg, &sb, &sy, &ssm, shared)) continue;
        if (d->stride_w != 1 || (d->s != 3 && d->s != 5) || sa.nt != 256 || sseg != 4) continue;
        if ((sv == 4 && (reinterpret_cast<uintptr_t>(j.x) & 15) != 0) || (reinterpret_cast<uintptr_t>(j.dy) & 15) != 0) continue;
        if (!j.ws || j.ws_bytes < (size_t)sb * sa.slab * sizeof(float)) return DCV_ERR_WORKSPACE;
        if (t.count == WG_MULTI_MAX) launch();
        const int k = t.count++;
        t.start[k] = blocks; t.gx[k] = sb; t.variant[k] = (sv == 4 ? 0 : 2) + (d->s == 5 ? 1 : 0);
        t.x[k] = j.x; t.dy[k] = j.dy; t.partial[k] = reinterpret_cast<float*>(j.ws); t.a[k] = sa;
        blocks += sb * sy;
        if (ssm > smem_max) smem_max = ssm;
        float* partial = reinterpret_cast<float*>(j.ws);
        segs[(*nsegs)++] = dcv_slab_segment{partial, j.dw, sa.wtotal, sa.slab, sb, 0};
        if (j.db) segs[(*nsegs)++] = dcv_slab_segment{partial + sa.wtotal, j.db, (unsigned long long)d->cout, sa.slab, sb, 0};
        taken[i] = true;
    }
    launch();
    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
}

// second stage: immediate (one launch of the shared slab reducer) or handed back to the caller as segments (`defer` != nullptr)
static int wgrad_second_stage(const float* partial, float* dw, float* db, int chunks, size_t wtotal, size_t slab, int cout, dcv_slab_segment* defer, int* ndefer, hipStream_t st) {
    dcv_slab_segment seg[2] = {{partial, dw, wtotal, slab, chunks, 0}, {partial + wtotal, db, (unsigned long long)(db ? cout : 0), slab, chunks, 0}};
    const int n = db ? 2 : 1;
    if (defer) { for (int i = 0; i < n; ++i) defer[i] = seg[i]; *ndefer = n; return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH; }
    return dcv_launch_slab_reduce(n, seg, st);
}

int dcv_direct_conv2d_wgrad(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, float* db, bool* db_done, void* ws, size_t ws_bytes, hipStream_t st,
                            dcv_slab_segment* defer, int* ndefer) {
    if (db_done) *db_done = false;
    if (ndefer) *ndefer = 0;
    float* partial = reinterpret_cast<float*>(ws);
    WgTiledArgs ta{}; int tco, tvec, blocks; size_t smem;
    {
        WgSlideArgs sa{}; int sv, sseg, sb, sy; size_t ssm;
        if (wgrad_slide_plan(d, db != nullptr, &sa, &sv, &sseg, &sb, &sy, &ssm) && (sv == 1 || (reinterpret_cast<uintptr_t>(x) & 15) == 0) && (reinterpret_cast<uintptr_t>(dy) & 15) == 0) {
            if (!ws || ws_bytes < (size_t)sb * sa.slab * sizeof(float)) return DCV_ERR_WORKSPACE;
            dim3 grid((unsigned)sb, (unsigned)sy);
            if (sv == 4 && d->stride_w == 1) launch_wgrad_slide<4, 1>(d->s, grid, ssm, st, sa, x, dy, partial);
            else if (sv == 4) launch_wgrad_slide<4, 2>(d->s, grid, ssm, st, sa, x, dy, partial);
            else if (d->stride_w == 1) launch_wgrad_slide<1, 1>(d->s, grid, ssm, st, sa, x, dy, partial);
            else launch_wgrad_slide<1, 2>(d->s, grid, ssm, st, sa, x, dy, partial);
            if (db_done) *db_done = db != nullptr;
            return wgrad_second_stage(partial, dw, db, sb, sa.wtotal, sa.slab, d->cout, defer, ndefer, st);
        }
    }
    if (wgrad_tiled_plan(d, &ta, &tco, &tvec, &blocks, &smem) && (tvec == 1 || (reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
        if (!ws || ws_bytes < (size_t)blocks * ta.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
        dim3 grid((unsigned)blocks, (unsigned)dcv_ceil_div(d->cout, tco));
        if (tvec == 4) launch_wgrad_tiled<4>(tco, grid, smem, st, ta, x, dy, partial);
        else launch_wgrad_tiled<1>(tco, grid, smem, st, ta, x, dy, partial);
        return wgrad_second_stage(partial, dw, nullptr, blocks, ta.wtotal, ta.wtotal, d->cout, defer, ndefer, st);
    }
    WgradArgs a{}; int chunks, ubatches, vec;
    wgrad_plan(d, &a, &chunks, &ubatches, &vec);
    if (!ws || ws_bytes < (size_t)chunks * a.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
    if (ubatches > 65535) return DCV_ERR_UNSUPPORTED;
    dim3 grid((unsigned)chunks, (unsigned)ubatches);
    if (vec == 4 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) hipLaunchKernelGGL((conv_wgrad_partial_kernel<4>), grid, dim3(256), 0, st, a, x, dy, partial);
    else if (vec == 4) return DCV_ERR_INVALID_ARG;   // NHWC activations are always 16-B aligned (torch allocator); anything else is a caller bug
    else hipLaunchKernelGGL((conv_wgrad_partial_kernel<1>), grid, dim3(256), 0, st, a, x, dy, partial);
    return wgrad_second_stage(partial, dw, nullptr, chunks, a.wtotal, a.wtotal, d->cout, defer, ndefer, st);
}
