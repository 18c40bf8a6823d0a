// conv2d_internal.h — internal (non-exported) entry points of the convolution back-ends, dispatched by conv2d.hip.
#pragma once
#include "common.h"

// direct (register-blocked, LDS-resident filter) back-end: any shape, tuned for the small-channel stock layers
size_t dcv_direct_conv2d_fwd_ws(const dcv_conv2d_desc* d);
size_t dcv_direct_conv2d_dgrad_ws(const dcv_conv2d_desc* d);
// `prepared`: ws already holds the prepared filter of this (d, op) (dcv_conv2d_prepare_filters); otherwise it is scratch and the filter is prepared first
// `stats` (nullable): per-block sums of y and y^2 per output channel, [2][Cout][rows] doubles with rows = dcv_*_conv2d_fwd_stats_rows(d) (0: unsupported);
// *rows_per_sample (nullable) > 0: no row mixes samples, sample n owns rows [n * rps, (n + 1) * rps)
int dcv_direct_conv2d_fwd(const dcv_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, hipStream_t st, bool prepared = false, double* stats = nullptr);
int dcv_direct_conv2d_fwd_stats_rows(const dcv_conv2d_desc* d, int* rows_per_sample = nullptr);
// `addend` (nullable): dx = data gradient + addend (a tensor shaped like dx); DCV_ERR_UNSUPPORTED when the kernel serving the layer cannot (nothing is launched then)
int dcv_direct_conv2d_dgrad(const dcv_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes, hipStream_t st, bool prepared = false,
                            const float* addend = nullptr);
size_t dcv_direct_conv2d_wgrad_ws(const dcv_conv2d_desc* d);
bool dcv_direct_conv2d_wgrad_fuses_bias(const dcv_conv2d_desc* d);
int dcv_direct_conv2d_wgrad_multi(int count, const dcv_wgrad_job* jobs, const bool* eligible, bool* taken, dcv_slab_segment* segs, int* nsegs, hipStream_t st);
int dcv_direct_conv2d_wgrad(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, float* db, bool* db_done, void* ws, size_t ws_bytes, hipStream_t st,
                            dcv_slab_segment* defer = nullptr, int* ndefer = nullptr);

// MFMA implicit-GEMM back-end (fp32 inputs, v_mfma_f32_16x16x4_f32): wide layers (conv2d_mfma.hip)
bool dcv_mfma_conv2d_supported(const dcv_conv2d_desc* d, int op /*0 fwd, 1 dgrad, 2 wgrad*/);
size_t dcv_mfma_conv2d_ws(const dcv_conv2d_desc* d, int op);
int dcv_mfma_conv2d_fwd(const dcv_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, hipStream_t st, bool prepared = false, double* stats = nullptr);
int dcv_mfma_conv2d_fwd_stats_rows(const dcv_conv2d_desc* d, int* rows_per_sample = nullptr);
int dcv_mfma_conv2d_dgrad(const dcv_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes, hipStream_t st, bool prepared = false);
int dcv_mfma_conv2d_wgrad(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st);
// bf16 x / dy (bf16 storage path): the all-taps kernel only; workspace = dcv_mfma_conv2d_ws(d, 2)
int dcv_mfma_wgrad_reduce_plain(const float* partial, float* dw, int chunks, size_t wtotal, hipStream_t st);
bool dcv_mfma_conv2d_wgrad_in16_supported(const dcv_conv2d_desc* d);
int dcv_mfma_conv2d_wgrad_in16(const dcv_conv2d_desc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st);
// dw[co][ci][tap] = ordered sum over chunks of partial[chunk][tap][co][ci] (3x3 filters)
int dcv_mfma_wgrad9_reduce(const float* partial, float* dw, int chunks, int cout, int cin, hipStream_t st);

// Prepared filters: one entry describes the re-layout of one OIHW filter for the kernel which will consume it (see dcv_conv2d_prepare_filters).
struct dcv_prep_entry {
    const float* w; float* dst;
    int kind;                      // 0: direct tiled kernel  wt[cout tile][R][S][cin_pad][tco]      1: MFMA kernel  wb[cout][tap][cin]      2: resident MFMA data gradient
    int kcout, kcin;               // kernel-view channel counts (data gradient: kcout = original Cin, kcin = original Cout)
    int R, S, w_cin;               // taps, OIHW row length (original Cin)
    int dgrad;                     // 0: forward filter, 1: transposed + flipped filter of the stride-1 data gradient
    int tco, tiles, cin_pad;       // kind 0 only
    int r0, rstep, Rw, s0, sstep, Sw;   // kind 1 data gradient: tap (r, s) of this (phase) filter is tap (r0 + rstep*(R-1-r), s0 + sstep*(S-1-s)) of the Rw x Sw filter
    int dst_off;                   // element offset of this entry inside the layer's prepared-filter buffer
    int total;                     // elements of dst
};
#define DCV_PREP_MAX 24
struct dcv_prep_list { int count; dcv_prep_entry e[DCV_PREP_MAX]; };

// value of element i of the prepared filter described by e
static __device__ __forceinline__ float dcv_prep_element(const dcv_prep_entry& e, int i) {
    if (e.kind == 0) {
        const int co = i % e.tco; int t = i / e.tco;
        const int ci = t % e.cin_pad; t /= e.cin_pad;
        const int s = t % e.S; t /= e.S;
        const int r = t % e.R; const int tile = t / e.R;
        const int cog = tile * e.tco + co;
        if (cog >= e.kcout || ci >= e.kcin) return 0.f;
        if (!e.dgrad) return e.w[(((size_t)cog * e.w_cin + ci) * e.R + r) * e.S + s];
        return e.w[(((size_t)ci * e.w_cin + cog) * e.R + (e.R - 1 - r)) * e.S + (e.S - 1 - s)];
    }
    if (e.kind == 2) {
        // B fragments of the LDS-resident network's matrix-core DATA gradient (resident.hip): [n 16][kq 4][j 4][tap R*S], the value lane (n, kq) feeds to the
        // MFMA which contracts output channels {4 kq + j}: w[co = 4 kq + j][ci = n][flipped tap], zero for n >= Cin (e.kcout = original Cin, e.kcin = 16)
        const int taps = e.R * e.S;
        const int tap = i % taps; int t = i / taps;
        const int j = t & 3; t >>= 2;
        const int kq = t & 3; const int n = t >> 2;
        const int co = 4 * kq + j;
        if (n >= e.kcout || co >= e.kcin) return 0.f;
        return e.w[((size_t)co * e.w_cin + n) * taps + (taps - 1 - tap)];
    }
    const int ci = i % e.kcin; int t = i / e.kcin;
    const int s = t % e.S; t /= e.S;
    const int r = t % e.R; const int co = t / e.R;
    if (!e.dgrad) return e.w[(((size_t)co * e.w_cin + ci) * e.R + r) * e.S + s];
    return e.w[(((size_t)ci * e.w_cin + co) * e.Rw + (e.r0 + e.rstep * (e.R - 1 - r))) * e.Sw + (e.s0 + e.sstep * (e.S - 1 - s))];
}
// The inverse map (round 4: the optimizer's tail launch SCATTERS an updated weight into its prepared layouts instead of gathering whole layouts behind a barrier):
// index in e.dst of element `off` of e.w (OIHW: [first][w_cin][R][S]), or -1 when the layout has no such element.  Pad positions of a layout (zeros) are written by
// the gather which creates it and never change.  Supported: kinds 0 and 2, kind 1 forward filters (dcv_prep_scatterable).
static __host__ __device__ __forceinline__ bool dcv_prep_scatterable(const dcv_prep_entry& e) { return e.kind == 0 || e.kind == 2 || (e.kind == 1 && !e.dgrad); }
static __device__ __forceinline__ int dcv_prep_scatter_index(const dcv_prep_entry& e, int off) {
    const int taps = e.R * e.S;
    const int tap_w = off % taps; int t = off / taps;
    const int b = t % e.w_cin, a = t / e.w_cin;                    // w[a][b][tap_w]
    if (e.kind == 0) {
        const int cog = e.dgrad ? b : a, ci = e.dgrad ? a : b, tap = e.dgrad ? taps - 1 - tap_w : tap_w;
        if (cog >= e.kcout || ci >= e.kcin) return -1;
        const int tile = cog / e.tco, co = cog - tile * e.tco;
        return ((tile * taps + tap) * e.cin_pad + ci) * e.tco + co;                        // [tile][R][S][cin_pad][tco]
    }
    if (e.kind == 2) {
        const int co = a, n = b;                                                           // value(n, kq, j, tap) = w[co = 4 kq + j][n][taps - 1 - tap]
        if (n >= e.kcout || co >= e.kcin) return -1;
        return ((n * 4 + (co >> 2)) * 4 + (co & 3)) * taps + (taps - 1 - tap_w);
    }
    if (a >= e.kcout || b >= e.kcin) return -1;
    return (a * taps + tap_w) * e.kcin + b;                                                // [co][r][s][ci]
}
// number of prepared-filter pieces of (d, op) written to out[] (<= DCV_PREP_PIECES); 0 when the kernel serving it reads the OIHW filter directly.
// op: 0 forward, 1 data gradient (a strided data gradient on the MFMA kernels has one piece per output phase)
#define DCV_PREP_PIECES 16
int dcv_direct_conv2d_prep_entries(const dcv_conv2d_desc* d, int op, dcv_prep_entry* out);
int dcv_mfma_conv2d_prep_entries(const dcv_conv2d_desc* d, int op, dcv_prep_entry* out);

// shared with elementwise.hip
extern "C" size_t dcv_act_bwd_workspace_bytes(size_t rows, int channels);
extern "C" int dcv_channel_sum(const float* x, float* out, size_t rows, int channels, void* ws, size_t ws_bytes, void* stream);
