// conv2d_mfma.hip — MFMA implicit-GEMM convolution back-end for wide layers (Cin % 16 == 0, Cout % 16 == 0): forward, stride-1 data gradient and
// weight gradient on `v_mfma_f32_16x16x4_f32` (fp32 in / fp32 accumulate: bit-wise an fmaf chain, 155 TF measured peak on MI355X).
//
// GEMM view (forward):  D[m][n] = sum_k A[m][k] * B[k][n],  m = output pixel (n, ho, wo), n = output channel, k = (r, s, ci) with ci fastest.
//   * A is never materialised: every K chunk (16 consecutive input channels of one filter tap) of a 128-pixel tile is gathered from the NHWC activation
//     with 16-B loads (4 lanes cover the 64 contiguous bytes of one pixel), zero-filled outside the image, and staged in LDS as [pixel][16 + 4 pad].
//   * B is the filter re-laid out once per launch as wb[cout][tap][cin] (K contiguous), staged as [cout][16 + 4 pad].
//   * K permutation: inside a 16-wide chunk lane group g = lane >> 4 owns k = 4g..4g+3 and uses element t in MFMA step t, for A and B alike, so each
//     lane fetches its four k-steps with ONE conflict-free ds_read_b128 (row stride 80 B spreads 16 rows over 16 distinct 16-B slots).
//   * block = 256 threads = 4 waves, tile 128 pixels x BN channels (BN in {16, 32, 64}); wave w owns pixels [32w, 32w+32) x BN: 2 x BN/16 accumulator
//     tiles; per chunk and wave: 2 + BN/16 ds_read_b128 for 8 * BN/16 MFMAs.  Global loads of chunk c+1 are issued before the MFMAs of chunk c and
//     written to the other LDS buffer afterwards (one barrier per chunk).
//   * epilogue: bias + activation, accumulator layout row = 4*(lane>>4) + reg (pixel), col = lane & 15 (channel) -> 64-B channel runs per pixel.
// dgrad (stride 1) = the same kernel on dy with the filter flipped / transposed by the prep kernel and pad' = dil*(R-1) - pad.
// wgrad: D[co][(tap, ci)] = sum over pixels; split over pixel chunks, per-chunk slabs + ordered reduce (deterministic).
//
// Roofline: MFMA fp32 (157.3 TF spec).  FLOPs per launch = 2 * N*Ho*Wo * Cout * Cin*R*S.
// Replaces torch.nn.Conv2d forward/backward (reference src/deepcv/meta/submodule_creators.py:251, ignite_training.py:253) for wide layers.
#include "conv2d_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA issued from inline assembly.  The builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) makes the compiler track the load as a store to LDS: it then puts
// `s_waitcnt vmcnt(0)` in front of the next LDS access it cannot prove disjoint — the step's first fragment read waited for the NEXT step's tiles, just requested
// (conv_mfma_wgrad9_kernel: 13 % of its time) — and, counting in order, every wait for an older register load also waits for the requests behind it.  Issued like
// this the compiler does not know about it; the kernels wait themselves (__builtin_amdgcn_s_waitcnt, which the compiler's own bookkeeping understands) before the
// barrier that publishes the tiles.
__device__ __forceinline__ i32x4 mf_buffer(const void* p, unsigned bytes) {      // raw buffer resource (what __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000) builds)
    const unsigned long long a = (unsigned long long)p;
    return (i32x4){(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// 64 lanes x 16 bytes from rsrc[voffset + soffset] (voffset range-checked against the resource: beyond it the lanes fetch zeros) to LDS bytes [lds_addr, lds_addr + 1024)
__device__ __forceinline__ void mf_dma16(i32x4 rsrc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voffset), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soffset)) : "memory");
}
#define MF_WAIT_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)      // s_waitcnt vmcnt(0) (expcnt, lgkmcnt untouched)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// four consecutive K elements of a lane -> the bf16 operand of v_mfma_f32_16x16x16_bf16 (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16x4 pack_bf16(const f32x4& v) { return (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; }
__device__ __forceinline__ bf16x4 pack_bf16(float a, float b, float c, float d) { return (bf16x4){(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d}; }

#define MF_BM 128
#define MF_BK 16
#define MF_LD 20                 // padded LDS row length (floats): 80 B stride -> conflict-free ds_read_b128 across 16 rows

struct MfmaArgs {
    int N, Hin, Win, Cin;          // tensor read (fwd: x, dgrad: dy)
    int Hout, Wout, Cout;          // tensor written
    int R, S, sh, sw, ph, pw, dh, dw;
    int act; float slope;
    long M;                        // N * Hout * Wout  (Hout x Wout: the grid of pixels this launch computes)
    // strided data gradient: this launch computes phase (oh_off, ow_off) of the full gradient, written at rows oh_mul*i + oh_off, columns ow_mul*j + ow_off
    int phase, oh_mul, ow_mul, oh_off, ow_off, Hfull, Wfull;
    unsigned magic_w, magic_h;     // floor((2^32 - 1) / Wout) + 1, ... / Hout + 1 when M * max(Wout, Hout) < 2^32 (n / d = mulhi(n, magic)), else 0: 64-bit divisions
};

// wb[co][tap][ci] (fwd: w[co][ci][r][s]; dgrad: kernel-cout = original cin, kernel-cin = original cout, taps flipped)
__global__ __launch_bounds__(256) void mfma_prep_weights_kernel(const float* __restrict__ w, float* __restrict__ wb, int kcout, int kcin, int R, int S, int w_cin, int dgrad) {
    const int total = kcout * R * S * kcin;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % kcin; int t = i / kcin;
        const int s = t % S; t /= S;
        const int r = t % R; const int co = t / R;
        float v;
        if (!dgrad) v = w[(((size_t)co * w_cin + ci) * R + r) * S + s];
        else v = w[(((size_t)ci * w_cin + co) * R + (R - 1 - r)) * S + (S - 1 - s)];
        wb[i] = v;
    }
}

// BMR: A rows staged per thread = accumulator tiles along M per wave; block tile = (64 * BMR) pixels x BN channels
// re-lays out one prepared-filter piece (conv2d_internal.h: dcv_prep_element)
__global__ __launch_bounds__(256) void mfma_prep_piece_kernel(dcv_prep_entry e, const float* __restrict__ w, float* __restrict__ dst) {
    e.w = w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < e.total; i += gridDim.x * blockDim.x) dst[i] = dcv_prep_element(e, i);
}

// BF16: the four fp32 K steps of a chunk become ONE v_mfma_f32_16x16x16_bf16 (operands rounded to bf16 in registers, fp32 accumulate)
template <int BN, int BMR, bool BF16>
__global__ __launch_bounds__(256) void conv_mfma_kernel(MfmaArgs a, const float* __restrict__ x, const float* __restrict__ wb, const float* __restrict__ bias,
                                                         float* __restrict__ y, double* __restrict__ stats) {
    constexpr int NT = BN / 16;                                   // accumulator tiles along N per wave
    constexpr int BM = 64 * BMR;
    __shared__ __attribute__((aligned(16))) float As[2][BM * MF_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * MF_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- staging roles: thread stages A rows (t >> 2) and (t >> 2) + 64, channel quad q = t & 3; B row (t >> 2) when < BN
    const int q = t & 3;
    int an[BMR], aho[BMR], awo[BMR]; bool arow_ok[BMR];
#pragma unroll
    for (int j = 0; j < BMR; ++j) {
        const long m = m0 + (t >> 2) + 64 * j;
        arow_ok[j] = m < a.M;
        const long mm = arow_ok[j] ? m : 0;
        if (a.magic_w != 0u) {       // (uniform) 32-bit decode by multiply-high: three 64-bit divisions per row were microseconds of every block's prologue
            const unsigned um = (unsigned)mm, r1 = __umulhi(um, a.magic_w), nn = __umulhi(r1, a.magic_h);
            awo[j] = (int)(um - r1 * (unsigned)a.Wout); aho[j] = (int)(r1 - nn * (unsigned)a.Hout); an[j] = (int)nn;
        } else { awo[j] = (int)(mm % a.Wout); const long r1 = mm / a.Wout; aho[j] = (int)(r1 % a.Hout); an[j] = (int)(r1 / a.Hout); }
    }
    const int brow = t >> 2;
    const bool b_active = brow < BN;
    const int taps = a.R * a.S;
    const size_t wb_row = (size_t)taps * a.Cin;                   // floats per output channel in wb

    // ---- load stream.  MFMAs and ordinary vector instructions share the SIMD's issue port, so every VALU instruction spent on addressing is time the
    // matrix pipe idles (measured: time = MFMA time + VALU time, nothing overlaps).  The per-pixel address arithmetic (tap offset, clamping, validity)
    // is therefore done ONCE PER TAP; inside a tap consecutive chunks only advance the pointers by 16 channels.
    const float* aptr[BMR]; float aokf[BMR];                      // validity as a 0 / 1 factor: a select on the loaded value would be turned into a guarded (branchy) load
    const float* bptr = wb + (size_t)(n0 + (b_active ? brow : 0)) * wb_row + 4 * q;      // wb[co][tap][ci]: chunks are consecutive across taps
    int lr = 0, ls = 0, lcc = 0;                                  // position of the next chunk to LOAD
    auto set_tap = [&](int r, int s) {
#pragma unroll
        for (int j = 0; j < BMR; ++j) {
            const int hi = aho[j] * a.sh - a.ph + r * a.dh, wi = awo[j] * a.sw - a.pw + s * a.dw;
            aokf[j] = (arow_ok[j] && hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win) ? 1.f : 0.f;
            const int hic = hi < 0 ? 0 : (hi >= a.Hin ? a.Hin - 1 : hi), wic = wi < 0 ? 0 : (wi >= a.Win ? a.Win - 1 : wi);
            aptr[j] = x + (((size_t)an[j] * a.Hin + hic) * a.Win + wic) * a.Cin + 4 * q;            // clamped: always a valid address
        }
    };
    auto load_next = [&](float4 (&v)[BMR], float4& vbv) {
#pragma unroll
        for (int j = 0; j < BMR; ++j) {
            const float4 g = *reinterpret_cast<const float4*>(aptr[j]);                             // unconditional (clamped address: finite data)
            v[j] = aokf[j] != 0.f ? g : make_float4(0.f, 0.f, 0.f, 0.f);
            aptr[j] += MF_BK;
        }
        vbv = *reinterpret_cast<const float4*>(bptr);                                              // rows beyond BN are never stored to LDS
        bptr += MF_BK;
        if (++lcc == a.Cin / MF_BK) {                                                               // next tap (wave-uniform)
            lcc = 0;
            if (++ls == a.S) { ls = 0; ++lr; }
            if (lr < a.R) set_tap(lr, ls);
        }
    };
    auto store_stage = [&](int buf, const float4 (&va)[BMR], const float4& vb) {
#pragma unroll
        for (int j = 0; j < BMR; ++j) *reinterpret_cast<float4*>(&As[buf][((t >> 2) + 64 * j) * MF_LD + 4 * q]) = va[j];
        if (b_active) *reinterpret_cast<float4*>(&Bs[buf][brow * MF_LD + 4 * q]) = vb;
    };

    f32x4 acc[BMR][NT];
#pragma unroll
    for (int i = 0; i < BMR; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int cchunks = a.Cin / MF_BK;
    const int nchunks = taps * cchunks;
    float4 va[BMR]; float4 vb;
    set_tap(0, 0);
    load_next(va, vb);
    store_stage(0, va, vb);
    __syncthreads();

    const int li = lane & 15, lg = lane >> 4;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        const bool has_next = (c + 1) < nchunks;
        if (has_next) load_next(va, vb);

        // ---- MFMAs on buffer `buf`
        f32x4 af[BMR], bf[NT];
#pragma unroll
        for (int i = 0; i < BMR; ++i) af[i] = *reinterpret_cast<const f32x4*>(&As[buf][(wave * 16 * BMR + i * 16 + li) * MF_LD + 4 * lg]);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][(j * 16 + li) * MF_LD + 4 * lg]);
        if constexpr (BF16) {
            bf16x4 ah[BMR], bh[NT];
#pragma unroll
            for (int i = 0; i < BMR; ++i) ah[i] = pack_bf16(af[i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) bh[j] = pack_bf16(bf[j]);
#pragma unroll
            for (int i = 0; i < BMR; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[i], bh[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < BMR; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
        }

        if (has_next) store_stage(buf ^ 1, va, vb);
        __syncthreads();
    }

    // ---- epilogue: acc[i][j][reg] -> pixel m0 + wave*16*BMR + i*16 + 4*lg + reg, channel n0 + j*16 + li
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = n0 + j * 16 + li;
        const float bv = bias ? bias[co] : 0.f;
        ssum[j] = 0.f; ssq[j] = 0.f;
#pragma unroll
        for (int i = 0; i < BMR; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long m = m0 + wave * 16 * BMR + i * 16 + 4 * lg + reg;
                if (m < a.M) {
                    const float o = dcv_act_apply(acc[i][j][reg] + bv, a.act, a.slope);
                    size_t off = (size_t)m;
                    if (a.phase) {                                 // (n, i, j) of the phase grid -> position in the full tensor
                        int jj, ii; long nn;
                        if (a.magic_w != 0u) { const unsigned um = (unsigned)m, r1 = __umulhi(um, a.magic_w), n2 = __umulhi(r1, a.magic_h); jj = (int)(um - r1 * (unsigned)a.Wout); ii = (int)(r1 - n2 * (unsigned)a.Hout); nn = n2; }
                        else { jj = (int)(m % a.Wout); const long r1 = m / a.Wout; ii = (int)(r1 % a.Hout); nn = r1 / a.Hout; }
                        off = ((size_t)nn * a.Hfull + (size_t)ii * a.oh_mul + a.oh_off) * a.Wfull + (size_t)jj * a.ow_mul + a.ow_off;
                    }
                    y[off * a.Cout + co] = o;
                    ssum[j] += o; ssq[j] += o * o;
                }
            }
    }
    if (stats != nullptr) {
        // BatchNorm statistics partials of this 128-pixel tile: lanes li, li+16, li+32, li+48 hold the same channel (fixed-order xor butterfly), then waves 0..3
        double* red = reinterpret_cast<double*>(&As[0][0]);      // [4 waves][2][BN]; the staging buffers are dead (last loop iteration ended with a barrier)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s1 = ssum[j], s2 = ssq[j];
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lg == 0) { red[(wave * 2 + 0) * BN + j * 16 + li] = (double)s1; red[(wave * 2 + 1) * BN + j * 16 + li] = (double)s2; }
        }
        __syncthreads();
        if (t < 2 * BN) {
            const int q = t / BN, c = t - q * BN;
            const double v = ((red[(0 * 2 + q) * BN + c] + red[(1 * 2 + q) * BN + c]) + red[(2 * 2 + q) * BN + c]) + red[(3 * 2 + q) * BN + c];
            stats[((size_t)q * a.Cout + n0 + c) * gridDim.x + blockIdx.x] = v;                 // [2][Cout][rows]
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 / stride 1 / pad 1 wide layers: the input window staged once per chunk
// What the per-block time stamps of conv_mfma_kernel showed on 64 -> 64 @32x32, N = 512 (tools/variants/mfma_fwd_stamps.py, lessons 27 - 29 of DESIGN.md): the blocks
// of a CU start together, run their K loops in lock step at 77 % of the matrix pipe (a barrier and a round of global -> VGPR -> LDS staging per 32 MFMAs of a wave)
// and then all sit in a 20 us epilogue of 4-byte stores (19 % of a block's life, pipe idle).  And tools/probes/mfma16_probe.hip: the matrix pipe and the VALU share
// the SIMD's issue port ACROSS waves — time = 32 clocks per MFMA + ~4 per VALU instruction of any resident wave — so every address computation anywhere in the
// kernel is paid in matrix-pipe time.  This kernel serves every wide layer of the BASELINE specs (3x3, stride 1, pad 1: forward and, with the flipped filter, data
// gradient) with four vector-ALU instructions per 64 MFMAs in its K loop:
//   * block = 256 threads, tile = 256 consecutive output pixels x 64 channels, wave = 64 pixels x 64 channels (16 accumulator tiles).
//   * per 16-channel chunk the FLAT input window of the tile — pixels m0 - W - 1 .. m0 + 255 + W + 1, zeros beyond the tensor — goes to LDS ONCE by LDS-DMA
//     (buffer_load ... lds, 16 B per lane, no VGPR round trip) as [row][16 + 4 pad] floats; tap (r, s) of pixel p reads row p + r W + s: nine times fewer bytes
//     into LDS than a gather per tap, ONE barrier per nine taps (576 MFMAs per wave), double buffered.  A DMA instruction of a wave covers 12 rows (lane l: row
//     l / 5, slot l % 5; slot 4 is the pad and fetches out of range = zeros; lanes 60 - 63 repeat the first four pieces of the next instruction's first row), so a
//     lane's offsets advance by a constant: one add per piece.  Offsets are computed modulo 2^32: pixels in front of the tensor wrap beyond num_records, pixels
//     behind it exceed it — both read zeros without a compare.
//   * a tap outside the image reads the buffer's all-zero row instead: every lane keeps the 36 LDS addresses (4 tile rows x 9 taps) it will ever use, computed
//     once per block (two 16-bit addresses per register: one shift / mask per fragment read is all the vector ALU does in the K loop); the second window buffer
//     lies a compile-time distance behind the first (BSTRIDE: the ds_read's immediate offset).
//   * filter fragments never touch LDS: every lane fetches its own 16 B of wb[co][tap][ci] (buffer_load_dwordx4, one VGPR of addressing, scalar offsets), one
//     tap ahead in registers; the four waves fetch the same lines (L1 / L2 hits, 8 B/clk/CU).
//   * operand roles swapped (A = filter, B = x): a lane holds FOUR CONSECUTIVE CHANNELS of one pixel per accumulator tile -> 16-byte stores, 64-B runs per pixel;
//     per-channel statistics by DPP row sums over the 16 pixel lanes.
// Registers: __launch_bounds__(256, 3) = a budget of 168 <= 256 registers, which makes the compiler keep the accumulators in VGPRs (v_mfma ... vgprcd) instead of
// splitting the budget evenly between VGPRs and AGPRs; three blocks per CU (LDS 2 x 26 KB at W <= 32).
struct WinArgs {
    int N, H, W, Cin, Cout;
    int act; float slope;
    int M;                         // N * H * W; x_bytes < 2^31 (32-bit buffer offsets)
    unsigned x_bytes, w_bytes;
    int nrows;                     // 256 + 2 W + 2 window rows
    int ninstr;                    // DMA wave instructions per window: ceil(nrows / 12)
    int tiles;                     // ceil(M / 256)
    unsigned magic_w, magic_h;     // floor((2^32 - 1) / d) + 1: n / d = mulhi(n, magic) for n d < 2^32
};
#define MW_OOB 0x80000000u        // buffer offset beyond any num_records: the load delivers zeros
#define MW_ROWB 80                // bytes per window row: 16 floats + 4 pad (conflict-free ds_read_b128 over 16 consecutive rows)
typedef __attribute__((address_space(3))) void* mw_lds_ptr;

template <int CTRL> __device__ __forceinline__ float mw_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (fixed order), result on every lane of the row
__device__ __forceinline__ float mw_row_sum(float v) {
    v += mw_dpp<0xB1>(v);                       // quad_perm [1, 0, 3, 2]
    v += mw_dpp<0x4E>(v);                       // quad_perm [2, 3, 0, 1]
    v += mw_dpp<0x141>(v);                      // row_half_mirror
    v += mw_dpp<0x140>(v);                      // row_mirror
    return v;
}

// window buffer = [zero row][window rows 0 .. 12 ninstr]; buffer 1 starts BSTRIDE bytes behind buffer 0
// PLAIN_ACT: no activation, ReLU or leaky ReLU as one select + one multiply (the 64 inlined copies of the general activation switch — expf, tanhf — are 8000
// instructions and 120 VGPRs of epilogue)
template <bool PLAIN_ACT, int BSTRIDE>
__global__ __launch_bounds__(256, 3) void conv_mfma_win_kernel(WinArgs a, const float* __restrict__ x, const float* __restrict__ wb, const float* __restrict__ bias,
                                                                float* __restrict__ y, double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char mw_lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 15, lg = lane >> 4;
    // consecutive block indices go to different XCDs: give every XCD a contiguous range of tiles (neighbouring tiles share their halo rows in that XCD's L2)
    int tile = blockIdx.x;
    if ((a.tiles & 7) == 0) tile = (int)(blockIdx.x & 7u) * (a.tiles >> 3) + (int)(blockIdx.x >> 3);
    const int m0 = tile * 256, n0 = blockIdx.y * 64;
    const int W = a.W, H = a.H;
    const i32x4 rx = mf_buffer(x, a.x_bytes);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (int)a.w_bytes, 0x00020000);
    if (t < 10) *reinterpret_cast<float4*>(mw_lds + (t >= 5 ? BSTRIDE : 0) + (t % 5) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);       // the two zero rows

    // ---- window staging: DMA instruction idx = 4 k + wave of a buffer covers window rows 12 idx .. 12 idx + 12 (+ 4 pieces); window row R = flat pixel m0 - W - 1 + R
    const int lrow = (lane * 13) >> 6, lslot = lane - 5 * lrow;                 // lane / 5, lane % 5
    const unsigned vstep = (unsigned)(48 * a.Cin * 4);                           // four instructions (one per wave) further
    const unsigned vb0 = lslot < 4 ? (unsigned)((m0 - W - 1 + 12 * wave + lrow) * a.Cin * 4 + lslot * 16) : MW_OOB;       // (mod 2^32, see above)
    const int kinstr = (a.ninstr + 3) >> 2;
    auto stage = [&](int buf, int chunk) {
        const unsigned base = (unsigned)(size_t)mw_lds + (unsigned)(buf * BSTRIDE + MW_ROWB + wave * 960);
        unsigned vo = vb0;
        for (int k = 0; k < kinstr; ++k) {
            if (4 * k + wave < a.ninstr) mf_dma16(rx, base + (unsigned)(k * 3840), vo, (unsigned)(chunk * 64));
            vo += vstep;
        }
    };

    // ---- operands of tile row i of this wave: pixel pl = 64 wave + 16 i + li of the block; tap (r, s) -> window row pl + r W + s, the lane's four channels at + 16 lg;
    // a tap outside the image -> the zero row (buffer offset 0)
    const unsigned wo0 = (unsigned)m0 - __umulhi((unsigned)m0, a.magic_w) * (unsigned)W;               // (uniform: scalar unit)
    const unsigned r0 = __umulhi((unsigned)m0, a.magic_w), ho0 = r0 - __umulhi(r0, a.magic_h) * (unsigned)H;
    unsigned xad[4][5];                                            // two 16-bit LDS addresses per register: taps 2 q (low half) and 2 q + 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = 64 * wave + 16 * i + li;
        const unsigned u = wo0 + (unsigned)pl, dq = __umulhi(u, a.magic_w), wo = u - dq * (unsigned)W;
        const unsigned v = ho0 + dq, ho = v - __umulhi(v, a.magic_h) * (unsigned)H;
        const bool ok = m0 + pl < a.M;
        const bool rok[3] = {ok && ho > 0, ok, ok && ho + 1 < (unsigned)H}, cok[3] = {wo > 0, true, wo + 1 < (unsigned)W};
        const unsigned base = (unsigned)(MW_ROWB + pl * MW_ROWB + lg * 16);
        unsigned ad[10];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) ad[r * 3 + s2] = (rok[r] && cok[s2]) ? base + (unsigned)((r * W + s2) * MW_ROWB) : (unsigned)(lg * 16);
        ad[9] = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) xad[i][q] = ad[2 * q] | (ad[2 * q + 1] << 16);
    }

    // ---- filter fragments: lane (li, lg) of tile j, tap, chunk: wb[n0 + 16 j + li][tap][16 chunk + 4 lg ..]
    const int wvo = ((n0 + li) * 9 * a.Cin + 4 * lg) * 4;
    auto wload = [&](f32x4 (&w)[4], int chunk, int tap) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wvo, ((j * 144 + tap) * a.Cin + 16 * chunk) * 4, 0));
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int cchunks = a.Cin / 16;
    f32x4 wa[4], wb2[4];                                           // filter fragments of the tap in flight and of the next one (register double buffer)
    stage(0, 0);
    wload(wa, 0, 0);
    MF_WAIT_VMCNT0();
    __syncthreads();
    // one chunk: nine taps out of window buffer PAR (a compile-time distance: the immediate offset of the ds_reads)
    auto chunk_body = [&](auto par, int chunk) {
        constexpr int PAR = decltype(par)::value;
        // (the unpacked addresses are loop invariants: hoisted, they would need 36 registers the kernel does not have)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 5; ++q) asm volatile("" : "+v"(xad[i][q]));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            f32x4 (&wc)[4] = (tap & 1) ? wb2 : wa;
            f32x4 (&wn)[4] = (tap & 1) ? wa : wb2;
            if (tap < 8) wload(wn, chunk, tap + 1);                // requested a whole tap (64 MFMAs) before their first use
            else if (chunk + 1 < cchunks) wload(wn, chunk + 1, 0);
            f32x4 xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const f32x4*>(mw_lds + PAR * BSTRIDE + ((tap & 1) ? xad[i][tap >> 1] >> 16 : xad[i][tap >> 1] & 0xFFFFu));
            __builtin_amdgcn_sched_barrier(0);                     // (the scheduler would sink the loads behind the last use of the registers they overwrite)
            // the next window goes to the other buffer (last read in the previous chunk, which ended with a barrier); requested behind the first tap's operands:
            // vmcnt counts in order, so a wait for filter fragments also waits for every DMA issued before them
            if (tap == 0 && chunk + 1 < cchunks) stage(PAR ^ 1, chunk + 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[j][kk], xv[i][kk], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) wa[j] = wb2[j];                // (nine taps: the fragments of the next chunk's first tap arrived in the second set)
        MF_WAIT_VMCNT0();                                          // this wave's pieces of the next window have landed ...
        __syncthreads();                                           // ... and everybody's; every wave is done reading this one
    };
    int chunk = 0;
    for (; chunk + 1 < cchunks; chunk += 2) { chunk_body(std::integral_constant<int, 0>{}, chunk); chunk_body(std::integral_constant<int, 1>{}, chunk + 1); }
    if (chunk < cchunks) chunk_body(std::integral_constant<int, 0>{}, chunk);

    // ---- epilogue: acc[i][j][r] -> pixel m0 + 64 wave + 16 i + li, channel n0 + 16 j + 4 lg + r
    float ssum[4][4], ssq[4][4];
    float4 bv[4];
    const float neg = a.act == DCV_ACT_NONE ? 1.f : (a.act == DCV_ACT_RELU ? 0.f : a.slope);       // PLAIN_ACT: factor of the negative side
    auto actf = [&](float v) { if constexpr (PLAIN_ACT) return v > 0.f ? v : v * neg; else return dcv_act_apply(v, a.act, a.slope); };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bv[j] = bias ? *reinterpret_cast<const float4*>(bias + n0 + 16 * j + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[j][r] = 0.f; ssq[j][r] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 64 * wave + 16 * i + li;
        if (m < a.M) {
            float* yp = y + (size_t)m * a.Cout + n0 + 4 * lg;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 o;
                o.x = actf(acc[i][j][0] + bv[j].x); o.y = actf(acc[i][j][1] + bv[j].y); o.z = actf(acc[i][j][2] + bv[j].z); o.w = actf(acc[i][j][3] + bv[j].w);
                *reinterpret_cast<float4*>(yp + 16 * j) = o;
                ssum[j][0] += o.x; ssq[j][0] += o.x * o.x; ssum[j][1] += o.y; ssq[j][1] += o.y * o.y;
                ssum[j][2] += o.z; ssq[j][2] += o.z * o.z; ssum[j][3] += o.w; ssq[j][3] += o.w * o.w;
            }
        }
    }
    if (stats != nullptr) {
        // BatchNorm statistics partials of this 256-pixel tile: the 16 pixel lanes of a row (DPP, fixed order), then the four waves through LDS (the windows are dead:
        // the loop ended with a barrier)
        double* red = reinterpret_cast<double*>(mw_lds);           // [4 waves][2][64]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = mw_row_sum(ssum[j][r]), s2 = mw_row_sum(ssq[j][r]);
                if (li == 0) { red[(wave * 2 + 0) * 64 + 16 * j + 4 * lg + r] = (double)s1; red[(wave * 2 + 1) * 64 + 16 * j + 4 * lg + r] = (double)s2; }
            }
        __syncthreads();
        if (t < 128) {
            const int q = t >> 6, c = t & 63;
            const double v = ((red[(0 * 2 + q) * 64 + c] + red[(1 * 2 + q) * 64 + c]) + red[(2 * 2 + q) * 64 + c]) + red[(3 * 2 + q) * 64 + c];
            stats[((size_t)q * a.Cout + n0 + c) * a.tiles + tile] = v;                             // [2][Cout][rows]
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][tap][ci] = sum_m dy[m][co] * x[m (+) tap][ci].  Block = (pixel chunk, tap, 64-channel column tile of ci, 64-row tile of co).
// MFMA roles: A'[i = co][k = pixel], B'[k = pixel][j = ci].  Pixel tiles of 16 are staged in LDS in their natural [pixel][channel] layouts; lane (i, g)
// reads the four pixels 4g..4g+3 of its column with four ds_read_b32 (k permutation as in the forward kernel).
struct MfmaWgArgs {
    int N, H, W, Cin, Ho, Wo, Cout, R, S, sh, sw, ph, pw, dh, dw;
    int step_w, step_h;            // WG_PS % Wo, WG_PS / Wo
    int idx32;                     // x and dy both have fewer than 2^31 elements
    long M, m_per_chunk;           // output pixels, pixels per split-K chunk (multiple of 16)
    size_t wtotal;
    unsigned magic_w, magic_h;     // floor((2^32 - 1) / Wo) + 1, ... / Ho + 1 when M * max(Wo, Ho) < 2^32 (n / d = mulhi(n, magic)), else 0
};

#define WG_T 64                    // tile of output channels (rows) and of input channels (columns) per block
#define WG_LD 68                   // padded row length of the [16 pixels][64 channels] LDS tiles

#define WG_PS 32                   // pixels per pipeline step (one barrier per step: 32 MFMAs per wave between barriers)

template <bool BF16>
__global__ __launch_bounds__(256) void conv_mfma_wgrad_kernel(MfmaWgArgs a, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float Gs[2][WG_PS * WG_LD];     // dy tile  [pixel][co]
    __shared__ __attribute__((aligned(16))) float Xs[2][WG_PS * WG_LD];     // x  tile  [pixel][ci]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int ci_tiles = a.Cin / WG_T;
    const int tap = blockIdx.y / ci_tiles, ci0 = (blockIdx.y % ci_tiles) * WG_T;
    const int r = tap / a.S, s = tap % a.S;
    const int co0 = blockIdx.z * WG_T;
    const long mb = (long)blockIdx.x * a.m_per_chunk;
    long me = mb + a.m_per_chunk; if (me > a.M) me = a.M;

    // staging role: thread loads pixel rows (t >> 4) and (t >> 4) + 16 of the step, float4 column (t & 15), of both tiles
    const int prow = t >> 4, pc4 = t & 15;
    // Pixel coordinates of this thread's two rows are advanced incrementally (WG_PS pixels per step): a 64-bit divide per row and step cost more
    // vector-ALU time than the 32 MFMAs of the step (MFMAs and VALU instructions share the issue port).
    long mrow[2]; int pwo[2], pho[2], pn[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        mrow[h] = mb + prow + 16 * h;
        const long mm = mrow[h] < a.M ? mrow[h] : a.M - 1;
        if (a.magic_w != 0u) { const unsigned um = (unsigned)mm, r1 = __umulhi(um, a.magic_w), nn = __umulhi(r1, a.magic_h); pwo[h] = (int)(um - r1 * (unsigned)a.Wo); pho[h] = (int)(r1 - nn * (unsigned)a.Ho); pn[h] = (int)nn; }
        else { pwo[h] = (int)(mm % a.Wo); const long r1 = mm / a.Wo; pho[h] = (int)(r1 % a.Ho); pn[h] = (int)(r1 / a.Ho); }
    }
    auto load_tiles = [&](float4 (&g)[2], float4 (&xv)[2]) {          // loads the rows at (mrow, pwo, pho, pn), then advances them by WG_PS pixels
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = mrow[h] < me;
            const int n = pn[h] < a.N ? pn[h] : a.N - 1;             // positions past the end stay addressable (their values are zeroed)
            const long mm = ok ? mrow[h] : mb;
            const int hi = pho[h] * a.sh - a.ph + r * a.dh, wi = pwo[h] * a.sw - a.pw + s * a.dw;
            const bool xok = ok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            const int hic = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi), wic = wi < 0 ? 0 : (wi >= a.W ? a.W - 1 : wi);
            float4 gg, xx;
            if (a.idx32) {       // both tensors have fewer than 2^31 elements: 32-bit index arithmetic (half the vector-ALU work of the 64-bit form)
                gg = *reinterpret_cast<const float4*>(dy + ((unsigned)mm * (unsigned)a.Cout + (unsigned)(co0 + 4 * pc4)));
                xx = *reinterpret_cast<const float4*>(x + ((((unsigned)n * (unsigned)a.H + (unsigned)hic) * (unsigned)a.W + (unsigned)wic) * (unsigned)a.Cin + (unsigned)(ci0 + 4 * pc4)));
            } else {
                gg = *reinterpret_cast<const float4*>(dy + (size_t)mm * a.Cout + co0 + 4 * pc4);
                xx = *reinterpret_cast<const float4*>(x + (((size_t)n * a.H + hic) * a.W + wic) * a.Cin + ci0 + 4 * pc4);
            }
            g[h] = ok ? gg : make_float4(0.f, 0.f, 0.f, 0.f);
            xv[h] = xok ? xx : make_float4(0.f, 0.f, 0.f, 0.f);
            mrow[h] += WG_PS;
            pwo[h] += a.step_w; pho[h] += a.step_h;
            if (pwo[h] >= a.Wo) { pwo[h] -= a.Wo; ++pho[h]; }
            while (pho[h] >= a.Ho) { pho[h] -= a.Ho; ++pn[h]; }
        }
    };
    auto store_tiles = [&](int buf, const float4 (&g)[2], const float4 (&xv)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(&Gs[buf][(prow + 16 * h) * WG_LD + 4 * pc4]) = g[h];
            *reinterpret_cast<float4*>(&Xs[buf][(prow + 16 * h) * WG_LD + 4 * pc4]) = xv[h];
        }
    };

    // wave w owns output-channel rows [16w, 16w+16) x all 64 input-channel columns: 4 accumulator tiles
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 g[2], xv[2];
    load_tiles(g, xv);
    store_tiles(0, g, xv);
    __syncthreads();
    int it = 0;
    for (long m = mb; m < me; m += WG_PS, ++it) {
        const int buf = it & 1;
        const bool has_next = (m + WG_PS) < me;
        if (has_next) load_tiles(g, xv);
        if constexpr (BF16) {
#pragma unroll
            for (int hh = 0; hh < WG_PS / 16; ++hh) {                                       // 16 pixels per bf16 MFMA: lane group g supplies pixels 4g..4g+3
                const int p0 = hh * 16 + 4 * lg;
                const float* gp = &Gs[buf][p0 * WG_LD + wave * 16 + li];
                const bf16x4 ah = pack_bf16(gp[0], gp[WG_LD], gp[2 * WG_LD], gp[3 * WG_LD]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* xp = &Xs[buf][p0 * WG_LD + j * 16 + li];
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, pack_bf16(xp[0], xp[WG_LD], xp[2 * WG_LD], xp[3 * WG_LD]), acc[j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < WG_PS / 4; ++kk) {
                const int prw = (kk >> 2) * 16 + 4 * lg + (kk & 3);                         // k permutation: lane group g, step kk -> pixel row of the step
                const float av = Gs[buf][prw * WG_LD + wave * 16 + li];                    // A'[i = co][k = pixel]
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float bvv = Xs[buf][prw * WG_LD + j * 16 + li];                  // B'[k][j = ci]
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bvv, acc[j], 0, 0, 0);
                }
            }
        }
        if (has_next) store_tiles(buf ^ 1, g, xv);
        __syncthreads();
    }
    // partial[chunk][co][ci][r][s]  (OIHW so that the shared ordered reduce kernel can sum slabs directly)
    float* dst = partial + (size_t)blockIdx.x * a.wtotal;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int co = co0 + wave * 16 + 4 * lg + reg, ci = ci0 + j * 16 + li;
            dst[(((size_t)co * a.Cin + ci) * a.R + r) * a.S + s] = acc[j][reg];
        }
}


// ------------------------------------------------------------------------------------------------ weight gradient, all nine taps per block
// 3x3 / stride 1 / pad 1 layers (every wide layer of the BASELINE specs).  The kernel above gives one tap to each block, so x and dy are fetched nine times
// through L2 and a 32-pixel step feeds only 32 MFMAs per wave between two barriers: its global loads and their address arithmetic were 27 % of its time.
// Here a block owns a (pixel chunk, 64 input channels, 64 output channels) problem with ALL nine taps: per step of PS pixels of one image it stages the dy tile
// [PS][64] once and the flat window of x it touches — pixels [p0 - W - 1, p0 + PS + W + 1) of the image, zero outside the image — as [PS + 2W + 2][64]; tap
// (r, s) of pixel p reads window row p + r * W + s.  Rows above / below the image are the window's zero fill; a window row which wrapped around the left / right
// edge is cancelled by zeroing the dy operand of the border pixels for s = 0 / s = 2 (one ds_read of a per-pixel column mask + two selects per A operand).
// Per step and wave: 9 * PS MFMAs (v_mfma_f32_16x16x4_f32) on 36 accumulator tiles, one LDS read per MFMA, one barrier.
struct Wg9Args {
    int N, H, W, Cin, Cout;
    int XR;                         // window rows: PS + 2 * W + 2
    int HWp;                        // pixels per image rounded up to a multiple of PS: a step never straddles images, the padding pixels contribute zeros
    long M, m_per_chunk;            // N * HWp; (padded) pixels per split-K chunk (multiple of PS)
    size_t wtotal;
    unsigned magic_w;               // floor((2^32 - 1) / W) + 1: n / W = mulhi(n, magic_w) for n W < 2^32
};
#define W9_LDX 10                   // float4 loads of the x window per thread and step (XR * 16 <= 256 * W9_LDX)

// four consecutive channels at element index `idx` of an fp32 or (IN16) bf16 tensor, as fp32
template <bool IN16>
__device__ __forceinline__ float4 w9_load4(const float* base, size_t idx) {
    if constexpr (IN16) {
        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float4*>(base + idx);
    }
}

// IN16: x and dy are bf16 tensors (the bf16 storage path, conv2d_bf16.hip); they are widened while staged, everything downstream is unchanged
// fp32 tensors are staged by LDS-DMA (round 3: with the staging through registers — 12 float4 loads with 64-bit address arithmetic, 48 selects and 12 LDS stores per
// thread and step, every one of them paid in matrix-pipe time — the kernel spent 17 % of its time staging, tools/variants/mfma_wgrad9_skip.py): a wave instruction
// writes 64 consecutive 16-B pieces = 3 padded rows of 17 pieces + the first 13 pieces of a fourth (which the next instruction writes again), lane l = row l / 17,
// piece l % 17 (piece 16: the pad, fetched out of range); the buffer resource is the step's IMAGE (base x + n HW Cin, HW Cin 4 bytes), so window rows above / below
// the image and the padding pixels behind it are out of range = zeros without a compare; one add per piece.  Each tile is followed by three spare rows for the last
// instruction's overhang.
template <int PS, bool BF16, bool PIPE, bool IN16 = false>
__global__ __launch_bounds__(256, PIPE ? 1 : 2) void conv_mfma_wgrad9_kernel(Wg9Args a, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem9[];
    const int XR = a.XR;
    constexpr int GRA = PS + 3;                                   // rows per dy tile / x window incl. three spare rows (instructions cover 3 n + 1 rows)
    const int XRA = XR + 3;
    float* Gs = smem9;                                           // [2][GRA * WG_LD]  dy tile [pixel][co]
    float* Xs = Gs + 2 * GRA * WG_LD;                            // [2][XRA * WG_LD]  x window [row][ci]
    int* Cm = reinterpret_cast<int*>(Xs + 2 * XRA * WG_LD);      // [2][PS]           bit 0: w > 0, bit 1: w < W - 1
    float* Cf = reinterpret_cast<float*>(Cm + 2 * PS);           // [2][PS][2]        the same as factors: 1.f / 0.f for w > 0, w < W - 1
    const int t = threadIdx.x, lane = t & 63, wave = IN16 ? (t >> 6) : __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int ci0 = blockIdx.y * WG_T, co0 = blockIdx.z * WG_T;
    const long mb = (long)blockIdx.x * a.m_per_chunk;
    long me = mb + a.m_per_chunk; if (me > a.M) me = a.M;
    const int HW = a.H * a.W;

    const int prow = t >> 4, pc4 = t & 15;
    float4 g[PS / 16], xv[W9_LDX];
    auto load_step = [&](long m0) {
        const long n = m0 / a.HWp; const int p0 = (int)(m0 - n * a.HWp);                 // a step never straddles images (PS divides HWp)
#pragma unroll
        for (int h = 0; h < PS / 16; ++h) {
            const int p = p0 + prow + 16 * h;
            const bool ok = p < HW;
            const float4 v = w9_load4<IN16>(dy, ((size_t)n * HW + (ok ? p : 0)) * a.Cout + co0 + 4 * pc4);
            g[h] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const size_t xi = (size_t)n * HW * a.Cin + ci0 + 4 * pc4;
        const int q0 = p0 - a.W - 1;
#pragma unroll
        for (int u = 0; u < W9_LDX; ++u) {
            const int row = prow + 16 * u, q = q0 + row;
            const bool ok = row < XR && q >= 0 && q < HW;
            const float4 v = w9_load4<IN16>(x, xi + (size_t)(ok ? q : 0) * a.Cin);                       // clamped address, value selected below
            xv[u] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_step = [&](int buf, long m0) {
#pragma unroll
        for (int h = 0; h < PS / 16; ++h) *reinterpret_cast<float4*>(&Gs[(buf * GRA + prow + 16 * h) * WG_LD + 4 * pc4]) = g[h];
#pragma unroll
        for (int u = 0; u < W9_LDX; ++u) {
            const int row = prow + 16 * u;
            if (row < XR) *reinterpret_cast<float4*>(&Xs[((size_t)buf * XRA + row) * WG_LD + 4 * pc4]) = xv[u];
        }
        if (t < PS) { const int w = (int)((m0 % a.HWp + t) % a.W); Cm[buf * PS + t] = (w > 0 ? 1 : 0) | (w < a.W - 1 ? 2 : 0); }
    };
    // ---- LDS-DMA staging (fp32 tensors)
    const int lrow = (lane * 61) >> 10, lslot = lane - 17 * lrow;                // lane / 17, lane % 17
    const unsigned vbx = lslot < 16 ? (unsigned)((lrow + 3 * wave - a.W - 1) * a.Cin * 4 + ci0 * 4 + lslot * 16) : 0x80000000u;     // (mod 2^32: rows above the image wrap out of range)
    const unsigned vbg = lslot < 16 ? (unsigned)((lrow + 3 * wave) * a.Cout * 4 + co0 * 4 + lslot * 16) : 0x80000000u;
    const int nix = (XR + 2) / 3, nig = (PS + 2) / 3;                             // wave instructions per x window / dy tile (three rows each)
    int dn = 0, dp0 = 0;                                                         // image and first pixel of the NEXT step to stage (scalars)
    auto dma_step = [&](int buf) {
        if (t < PS) {
            const unsigned v = (unsigned)(dp0 + t), w = v - __umulhi(v, a.magic_w) * (unsigned)a.W;
            Cm[buf * PS + t] = (w > 0 ? 1 : 0) | (w + 1 < (unsigned)a.W ? 2 : 0);
            *reinterpret_cast<float2*>(&Cf[(buf * PS + t) * 2]) = make_float2(w > 0 ? 1.f : 0.f, w + 1 < (unsigned)a.W ? 1.f : 0.f);
        }
        const i32x4 rxi = mf_buffer(x + (size_t)dn * HW * a.Cin, (unsigned)(HW * a.Cin * 4));
        const i32x4 rgi = mf_buffer(dy + (size_t)dn * HW * a.Cout, (unsigned)(HW * a.Cout * 4));
        const unsigned gb = (unsigned)(size_t)(Gs + buf * GRA * WG_LD) + (unsigned)(wave * (3 * WG_LD * 4));
        const unsigned xb = (unsigned)(size_t)(Xs + (size_t)buf * XRA * WG_LD) + (unsigned)(wave * (3 * WG_LD * 4));
        unsigned vg = vbg + (unsigned)(dp0 * a.Cout * 4), vx = vbx + (unsigned)(dp0 * a.Cin * 4);
        const unsigned sg = (unsigned)(12 * a.Cout * 4), sx = (unsigned)(12 * a.Cin * 4);      // four instructions (one per wave) further
        for (int k = 0; 4 * k < nig; ++k) {
            if (4 * k + wave < nig) mf_dma16(rgi, gb + (unsigned)(k * (12 * WG_LD * 4)), vg, 0u);
            vg += sg;
        }
        for (int k = 0; 4 * k < nix; ++k) {
            if (4 * k + wave < nix) mf_dma16(rxi, xb + (unsigned)(k * (12 * WG_LD * 4)), vx, 0u);
            vx += sx;
        }
        dp0 += PS; if (dp0 >= a.HWp) { dp0 = 0; ++dn; }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the fp32 two-waves-per-SIMD path accumulates in 32 x 32 tiles instead (v_mfma_f32_32x32x2_f32): wave = 32 output channels x 32 input channels x nine taps
    constexpr bool W32 = !BF16 && !PIPE;
    f32x16 acc32[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[k][r] = 0.f;
    const int l31 = lane & 31, half = lane >> 5, cob = (wave >> 1) * 32, cib = (wave & 1) * 32;

    if constexpr (IN16) {
        load_step(mb);
        store_step(0, mb);
    } else {
        dn = (int)(mb / a.HWp); dp0 = (int)(mb - (long)dn * a.HWp);
        dma_step(0);
        MF_WAIT_VMCNT0();
    }
    __syncthreads();
    int it = 0;
    for (long m = mb; m < me; m += PS, ++it) {
        const int buf = it & 1;
        const bool has_next = (m + PS) < me;
        if (has_next) { if constexpr (IN16) load_step(m + PS); else dma_step(buf ^ 1); }
        const float* gs = Gs + buf * GRA * WG_LD;
        const float* xs = Xs + (size_t)buf * XRA * WG_LD;
        const int* cm = Cm + buf * PS;
        if constexpr (BF16) {
#pragma unroll
            for (int hh = 0; hh < PS / 16; ++hh) {
                const int p0q = hh * 16 + 4 * lg;
                float av[4]; int mk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { av[u] = gs[(p0q + u) * WG_LD + wave * 16 + li]; mk[u] = cm[p0q + u]; }
                const bf16x4 ah1 = pack_bf16(av[0], av[1], av[2], av[3]);
                const bf16x4 ah0 = pack_bf16((mk[0] & 1) ? av[0] : 0.f, (mk[1] & 1) ? av[1] : 0.f, (mk[2] & 1) ? av[2] : 0.f, (mk[3] & 1) ? av[3] : 0.f);
                const bf16x4 ah2 = pack_bf16((mk[0] & 2) ? av[0] : 0.f, (mk[1] & 2) ? av[1] : 0.f, (mk[2] & 2) ? av[2] : 0.f, (mk[3] & 2) ? av[3] : 0.f);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) {
                        const bf16x4 ah = sx == 0 ? ah0 : (sx == 1 ? ah1 : ah2);
                        const float* xp = xs + (size_t)(p0q + r * a.W + sx) * WG_LD + li;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[r * 3 + sx][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, pack_bf16(xp[j * 16], xp[WG_LD + j * 16], xp[2 * WG_LD + j * 16], xp[3 * WG_LD + j * 16]),
                                                                                           acc[r * 3 + sx][j], 0, 0, 0);
                    }
            }
        } else if constexpr (!PIPE) {
            // Two waves per SIMD: the other block's wave covers this wave's LDS round trips.  Every instruction a wave issues is paid in matrix-pipe time (~4 clocks per
            // VALU, ~8 per LDS instruction, tools/probes/mfma16_probe.hip), and the 16x16x4 form above needs one operand read per MFMA of 32 clocks: 19 % of the loop.
            // v_mfma_f32_32x32x2_f32 does 2048 MACs on the same two operand values per lane: K = 2 pixels per instruction, lanes 0 - 31 pixel k, lanes 32 - 63 pixel
            // k + 8 (rows 8 apart are 32 banks apart with the 68-float row: conflict-free), two neighbouring pixel pairs per ds_read2_b32, the border factors as floats.
            const float* cf = Cf + buf * PS * 2;
#pragma unroll
            for (int g16 = 0; g16 < PS / 16; ++g16)
#pragma unroll
                for (int pp = 0; pp < 8; pp += 2) {
                    const int row = 16 * g16 + pp + 8 * half;                                    // this lane's pixel of the first pair; the second pair: row + 1
                    const float* gp = gs + row * WG_LD + cob + l31;
                    const float av[2] = {gp[0], gp[WG_LD]};                                       // A'[i = co][k = pixel]
                    const float4 mf = *reinterpret_cast<const float4*>(cf + row * 2);            // (row is even)
                    const float a0[2] = {av[0] * mf.x, av[1] * mf.z}, a2[2] = {av[0] * mf.y, av[1] * mf.w};
                    float bq[9][2];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int sx = 0; sx < 3; ++sx) {
                            const float* xp = xs + (size_t)(row + r * a.W + sx) * WG_LD + cib + l31;  // B'[k = pixel shifted by the tap][j = ci]
                            bq[r * 3 + sx][0] = xp[0]; bq[r * 3 + sx][1] = xp[WG_LD];
                        }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int sx = 0; sx < 3; ++sx) {
                                const float as = sx == 0 ? a0[u] : (sx == 1 ? av[u] : a2[u]);
                                acc32[r * 3 + sx] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bq[r * 3 + sx][u], acc32[r * 3 + sx], 0, 0, 0);
                            }
                }
        } else {
            // One wave per SIMD (the 36 accumulator tiles + a double-buffered operand set need more than 256 registers): nothing but this wave's own
            // instruction stream hides the LDS latency, so the 36 B operands (+ A and the column mask) of pixel quad kk + 1 are read while the 36 MFMAs of
            // quad kk issue (with the reads placed next to their MFMAs the wave stalled for an LDS round trip every second MFMA).
            float bq[2][9][4], aq[2]; int mq[2];
            auto read_quad = [&](int kk, float (&b)[9][4], float& av, int& mk) {
                const int prw = (kk >> 2) * 16 + 4 * lg + (kk & 3);                         // k permutation of the 16x16x4 MFMA: lane group g, step kk -> pixel of the step
                av = gs[prw * WG_LD + wave * 16 + li];                                      // A'[i = co][k = pixel]
                mk = cm[prw];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) {
                        const float* xp = xs + (size_t)(prw + r * a.W + sx) * WG_LD + li;  // B'[k = pixel shifted by the tap][j = ci]
#pragma unroll
                        for (int j = 0; j < 4; ++j) b[r * 3 + sx][j] = xp[j * 16];
                    }
            };
            read_quad(0, bq[0], aq[0], mq[0]);
#pragma unroll
            for (int kk = 0; kk < PS / 4; ++kk) {
                const int cur = kk & 1;
                if (kk + 1 < PS / 4) read_quad(kk + 1, bq[cur ^ 1], aq[cur ^ 1], mq[cur ^ 1]);
                const float av = aq[cur];
                const float av0 = (mq[cur] & 1) ? av : 0.f, av2 = (mq[cur] & 2) ? av : 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) {
                        const float as = sx == 0 ? av0 : (sx == 1 ? av : av2);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[r * 3 + sx][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, bq[cur][r * 3 + sx][j], acc[r * 3 + sx][j], 0, 0, 0);
                    }
                // issue order of this quad: the LDS reads of the NEXT quad spread between this quad's MFMAs (a burst of 20 reads in front of them leaves the
                // matrix pipe without work for ~100 cycles per quad; reads sunk next to their own MFMAs stall the wave for an LDS round trip)
                if (kk + 1 < PS / 4) {
#pragma unroll
                    for (int q = 0; q < 18; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (IN16) { if (has_next) store_step(buf ^ 1, m + PS); }
        else MF_WAIT_VMCNT0();                                    // this wave's pieces of the next step have landed
        __syncthreads();
    }
    // partial[chunk][tap][co][ci]: the accumulator layout as it stands (16 lanes = 64 contiguous bytes of one row); OIHW order (taps fastest) would make every
    // store instruction 64 scattered dwords, 36 B apart.  mfma_wgrad9_reduce_kernel sums the slabs and writes dw in OIHW order.
    float* dst = partial + (size_t)blockIdx.x * a.wtotal;
    if constexpr (W32) {
        // 32x32 accumulator layout: register r of lane l = row 8 (r / 4) + 4 (l / 32) + r % 4 (co), column l % 32 (ci): 128 contiguous bytes per row
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cob + 8 * (r >> 2) + 4 * half + (r & 3), ci = ci0 + cib + l31;
                dst[((size_t)k * a.Cout + co) * a.Cin + ci] = acc32[k][r];
            }
        return;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = co0 + wave * 16 + 4 * lg + reg, ci = ci0 + j * 16 + li;
                dst[((size_t)k * a.Cout + co) * a.Cin + ci] = acc[k][j][reg];
            }
}

// dw[co][ci][tap] = sum over chunks of partial[chunk][tap][co][ci], chunks in order (deterministic).  Block = 8 float4 columns x 32 chunk lanes: a wave
// reads 128 contiguous bytes of 8 slabs per instruction; the 32 chunk lanes are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void mfma_wgrad9_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int chunks, int Cout, int Cin) {
    __shared__ float4 red[32][8];
    const int v4 = threadIdx.x & 7, cl = threadIdx.x >> 3;
    const size_t wtotal = (size_t)9 * Cout * Cin;
    const size_t e4 = (size_t)blockIdx.x * 8 + v4;                   // float4 index inside a slab ([tap][co][ci / 4])
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 * 4 < wtotal) {
        const float* src = partial + e4 * 4;
        int c = cl;
        for (; c + 96 < chunks; c += 128) {                          // four slabs in flight per thread (same summation order as the plain loop)
            const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)c * wtotal), v1 = *reinterpret_cast<const float4*>(src + (size_t)(c + 32) * wtotal);
            const float4 v2 = *reinterpret_cast<const float4*>(src + (size_t)(c + 64) * wtotal), v3 = *reinterpret_cast<const float4*>(src + (size_t)(c + 96) * wtotal);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; c < chunks; c += 32) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)c * wtotal);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[cl][v4] = s;
    __syncthreads();
    if (cl == 0 && e4 * 4 < wtotal) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 32; ++k) { const float4 v = red[k][v4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const size_t e = e4 * 4;
        const int ci = (int)(e % Cin); const size_t r1 = e / Cin; const int co = (int)(r1 % Cout), tap = (int)(r1 / Cout);
        float* d = dw + ((size_t)co * Cin + ci) * 9 + tap;
        d[0] = t.x; d[9] = t.y; d[18] = t.z; d[27] = t.w;
    }
}

// The same sum for FEW slabs (<= 16: wide layers, where one chunk already holds many blocks): with 32 chunk lanes most threads of the kernel above idle and its
// output goes out as 4-B stores 36 B apart.  Here a thread sums all slabs of one (tap, 4 input channels) item itself (every load independent), the block
// gathers the nine taps of (one output channel, CB input channels) in LDS and writes them as one contiguous run of dw.
template <int CB>
__global__ __launch_bounds__(9 * CB / 4 + 32) void mfma_wgrad9_reduce_few_kernel(const float* __restrict__ partial, float* __restrict__ dw, int chunks, int Cout, int Cin) {
    // block = 9 CB / 4 items rounded up to whole waves (CB = 128: 288 of 320 threads, CB = 64: 144 of 192): ONE round of items (with 256 threads the 288 items
    // of CB = 128 took two rounds, the second on 32 threads), up to sixteen independent 16-byte loads per thread in flight; the slabs are added in the same order
    constexpr int ITEMS = 9 * (CB / 4), THREADS = 9 * CB / 4 + 32;
    __shared__ __attribute__((aligned(16))) float out[CB * 9];
    const int co = blockIdx.y, ci0 = blockIdx.x * CB;
    const size_t wtotal = (size_t)9 * Cout * Cin;
    const int i = threadIdx.x;
    if (i < ITEMS) {
        const int tap = i / (CB / 4), v4 = i - tap * (CB / 4);
        const float* src = partial + ((size_t)tap * Cout + co) * Cin + ci0 + 4 * v4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int c = 0;
        for (; c + 8 <= chunks; c += 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(src + (size_t)(c + k) * wtotal);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
        }
        for (; c < chunks; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)c * wtotal);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        out[(4 * v4 + 0) * 9 + tap] = s.x; out[(4 * v4 + 1) * 9 + tap] = s.y; out[(4 * v4 + 2) * 9 + tap] = s.z; out[(4 * v4 + 3) * 9 + tap] = s.w;
    }
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(dw + ((size_t)co * Cin + ci0) * 9);
    for (int j = threadIdx.x; j < CB * 9 / 4; j += THREADS) dst[j] = reinterpret_cast<const float4*>(out)[j];
}

__global__ __launch_bounds__(1024) void mfma_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int chunks, size_t wtotal) {
    __shared__ float red[32][33];
    const int wl = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const size_t wi = (size_t)blockIdx.x * 32 + wl;
    float s = 0.f;
    if (wi < wtotal)
        for (int c = cl; c < chunks; c += 32) s += partial[(size_t)c * wtotal + wi];
    red[cl][wl] = s;
    __syncthreads();
    if (cl == 0 && wi < wtotal) {
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) tsum += red[k][wl];
        dw[wi] = tsum;
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int pick_bn(int cout) { return cout % 64 == 0 ? 64 : (cout % 32 == 0 ? 32 : 16); }

inline void wg_plan(const dcv_conv2d_desc* d, MfmaWgArgs* a, int* chunks) {
    a->N = d->n; a->H = d->h; a->W = d->w; a->Cin = d->cin; a->Ho = d->ho; a->Wo = d->wo; a->Cout = d->cout; a->R = d->r; a->S = d->s;
    a->sh = d->stride_h; a->sw = d->stride_w; a->ph = d->pad_h; a->pw = d->pad_w; a->dh = d->dil_h; a->dw = d->dil_w;
    a->M = (long)d->n * d->ho * d->wo;
    a->step_w = WG_PS % d->wo; a->step_h = WG_PS / d->wo;
    a->idx32 = ((size_t)d->n * d->h * d->w * d->cin < ((size_t)1 << 31) && (size_t)d->n * d->ho * d->wo * d->cout < ((size_t)1 << 31)) ? 1 : 0;
    a->wtotal = (size_t)d->cout * d->cin * d->r * d->s;
    {
        const bool small = d->wo >= 2 && d->ho >= 2 && (size_t)a->M * (size_t)(d->wo > d->ho ? d->wo : d->ho) < ((size_t)1 << 32);
        a->magic_w = small ? 0xFFFFFFFFu / (unsigned)d->wo + 1u : 0u; a->magic_h = small ? 0xFFFFFFFFu / (unsigned)d->ho + 1u : 0u;
    }
    // enough blocks to fill the chip (>= ~2048) but partial slabs bounded to 256 MiB
    const long blocks_per_chunk = (long)d->r * d->s * (d->cin / WG_T) * (d->cout / WG_T);
    long ch = (2048 + blocks_per_chunk - 1) / blocks_per_chunk;
    const long max_by_m = (a->M + 255) / 256;                      // at least 256 pixels per chunk
    if (ch > max_by_m) ch = max_by_m;
    const long max_by_bytes = (long)(((size_t)256 << 20) / (a->wtotal * sizeof(float)));
    if (ch > max_by_bytes) ch = max_by_bytes;
    if (ch < 1) ch = 1;
    long mpc = (a->M + ch - 1) / ch;
    mpc = (mpc + WG_PS - 1) / WG_PS * WG_PS;
    a->m_per_chunk = mpc;
    *chunks = (int)((a->M + mpc - 1) / mpc);
}


inline int wg9_mode() {               // 1: one wave per SIMD with an explicit operand pipeline, 2: two waves per SIMD
    static int m = -1;
    if (m < 0) { const char* e = getenv("DCV_MFMA_WGRAD9_MODE"); m = e ? atoi(e) : 2; }
    return m;
}

// all-taps weight-gradient kernel: 3x3, stride 1, pad 1, no dilation, 64-aligned channel counts, images whose pixel count is a multiple of the step
inline bool wg9_plan(const dcv_conv2d_desc* d, Wg9Args* a, int* ps, int* chunks, size_t* smem) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_MFMA_WGRAD9"); enabled = e ? atoi(e) : 1; }
    if (!enabled) return false;
    if (d->r != 3 || d->s != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1 || d->dil_h != 1 || d->dil_w != 1) return false;
    if (d->cin % WG_T != 0 || d->cout % WG_T != 0 || d->ho != d->h || d->wo != d->w) return false;
    const int hw = d->h * d->w;
    // step size: the one wasting fewer padding pixels per image, 32 on a tie; 16 when the 32-pixel window does not fit the LDS budget of two blocks per CU
    auto smem_of = [&](int ps) { return ((size_t)2 * (ps + 3) * WG_LD + (size_t)2 * (ps + 2 * d->w + 2 + 3) * WG_LD) * sizeof(float) + (size_t)2 * ps * sizeof(int) + (size_t)2 * ps * 2 * sizeof(float); };     // (three spare rows per tile; border masks as bits and as factors)
    auto fits = [&](int ps) { return (ps + 2 * d->w + 2) * 16 <= 256 * W9_LDX && smem_of(ps) <= (size_t)80 * 1024; };
    const int pad32 = (hw + 31) / 32 * 32, pad16 = (hw + 15) / 16 * 16;
    int PS = (pad32 <= pad16 && fits(32)) ? 32 : 16;
    if (!fits(PS)) return false;
    const int XR = PS + 2 * d->w + 2;
    *smem = smem_of(PS);
    a->N = d->n; a->H = d->h; a->W = d->w; a->Cin = d->cin; a->Cout = d->cout; a->XR = XR;
    a->HWp = PS == 32 ? pad32 : pad16;
    a->M = (long)d->n * a->HWp;
    a->wtotal = (size_t)d->cout * d->cin * 9;
    a->magic_w = 0xFFFFFFFFu / (unsigned)d->w + 1u;
    if (d->w < 2 || (size_t)hw * d->cin * 4 >= ((size_t)1 << 31) || (size_t)hw * d->cout * 4 >= ((size_t)1 << 31)) return false;      // (32-bit buffer offsets inside an image)
    const long blocks_per_chunk = (long)(d->cin / WG_T) * (d->cout / WG_T);
    const long per_cu = wg9_mode() == 1 ? 1 : 2;                                     // resident blocks per CU (the kernel's register budget): one round
    long ch = (per_cu * DCV_NUM_CU + blocks_per_chunk - 1) / blocks_per_chunk;
    const long max_by_m = a->M / (4 * PS) > 0 ? a->M / (4 * PS) : 1;                 // at least four steps per chunk
    if (ch > max_by_m) ch = max_by_m;
    const long max_by_bytes = (long)(((size_t)256 << 20) / (a->wtotal * sizeof(float)));
    if (ch > max_by_bytes) ch = max_by_bytes;
    if (ch < 1) ch = 1;
    long mpc = (a->M + ch - 1) / ch;
    mpc = (mpc + PS - 1) / PS * PS;
    a->m_per_chunk = mpc;
    *chunks = (int)((a->M + mpc - 1) / mpc);
    *ps = PS;
    return true;
}

}  // namespace

bool dcv_mfma_conv2d_supported(const dcv_conv2d_desc* d, int op) {
    if (d->groups != 1 || d->x_layout != DCV_LAYOUT_NHWC) return false;
    if (d->cin % 16 != 0 || d->cout % 16 != 0) return false;
    if (op == 1) {   // data gradient through the forward kernel
        if (d->stride_h == 1 && d->stride_w == 1) {
            if (d->dil_h * (d->r - 1) < d->pad_h || d->dil_w * (d->s - 1) < d->pad_w) return false;   // non-negative transposed padding only
        } else {
            // strided: one stride-1 launch per output phase (hi % stride_h, wi % stride_w) with the sub-filter of matching parity
            if (d->dil_h != 1 || d->dil_w != 1 || d->stride_h > 4 || d->stride_w > 4) return false;
            static int strided = -1;
            if (strided < 0) { const char* e = getenv("DCV_MFMA_STRIDED_DGRAD"); strided = e ? atoi(e) : 1; }
            if (!strided) return false;
        }
    }
    if (op == 2 && (d->cin % WG_T != 0 || d->cout % WG_T != 0)) return false;
    // measured on MI355X: from 16 channels on either side the MFMA kernel already beats the direct one (16->16 3x3 @16x16, N=512: 18.7 vs 29.5 us)
    static int min_c = -1;
    if (min_c < 0) { const char* e = getenv("DCV_MFMA_MIN_C"); min_c = e ? atoi(e) : 16; }
    return (d->cin >= min_c || d->cout >= min_c);
}

size_t dcv_mfma_conv2d_ws(const dcv_conv2d_desc* d, int op) {
    if (op == 2) {
        Wg9Args a9{}; int ps9, chunks9; size_t sm9;
        if (wg9_plan(d, &a9, &ps9, &chunks9, &sm9)) return dcv_align_up((size_t)chunks9 * a9.wtotal * sizeof(float), 256);
        MfmaWgArgs a{}; int chunks; wg_plan(d, &a, &chunks); return dcv_align_up((size_t)chunks * a.wtotal * sizeof(float), 256);
    }
    return dcv_align_up((size_t)d->cout * d->cin * d->r * d->s * sizeof(float), 256);
}

// 256-pixel tiles (BMR = 4) halve the non-MFMA instructions per MFMA (fragment reads, stores, loop) at the price of fewer resident waves
inline int pick_bmr(long M, int cin, int cout, int bn) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("DCV_MFMA_BMR"); forced = e ? atoi(e) : 0; }
    if (forced == 2 || forced == 4) return forced;
    const long blocks256 = ((M + 255) / 256) * (cout / bn);
    return (bn == 64 && cin >= 256 && blocks256 >= 2 * DCV_NUM_CU) ? 4 : 2;      // measured: 256-pixel tiles pay off from 256 input channels (110 vs 104 TF), lose below (86 vs 97)
}

// ------------------------------------------------------------------------------------------------ narrow layers: halo tile in LDS
// For 16 / 32 input channels the implicit-GEMM kernel above is a chain of latencies: every tap gathers a 128 x 16 operand tile from global memory for
// only 8 MFMAs per wave.  Here the zero-padded input tile of TH output rows of one image is staged ONCE (pixel-major, pixel stride Cin + 4 words: the 16
// pixels x 4 channels of an A operand fall into 64 different banks), the filter slice of the block's 16 output channels lives in registers as B
// fragments (taps * Cin / 4 VGPRs, read once from the K-major prepared filter), and every wave walks 16-pixel row segments: taps * Cin / 4 MFMAs per
// segment, each fed by one ds_read_b32.  Same epilogue (bias, activation, BatchNorm statistics partials [2][Cout][tiles]) and the same filter layout as
// the wide kernel, so it serves the forward pass and the stride-1 data gradient alike.
struct HaloArgs {
    int N, Hin, Win, Cin, Hout, Wout, Cout, R, S, ph, pw, act; float slope;
    int TH, XH, XW, PS;            // output rows per tile, halo rows / columns, pixel stride in LDS (words)
    int tiles_per_img, total_tiles;
};

template <int CIN, int TAPS>
__global__ __launch_bounds__(256) void conv_mfma_halo_kernel(HaloArgs a, const float* __restrict__ x, const float* __restrict__ wb, const float* __restrict__ bias,
                                                              float* __restrict__ y, double* __restrict__ stats) {
    constexpr int KC = CIN / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);                   // [XH][XW][PS]
    const int t = threadIdx.x, lane = t & 63, lg = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int co0 = blockIdx.y * 16;
    // B'[k = ci][j = co]: lane (lg, li) holds w[co0 + li][tap][4 * kc + lg].  The 16 filter rows of the block are contiguous in the K-major prepared filter:
    // coalesced 16-B loads into LDS (row stride + 4 words: conflict-free fragment reads), then TAPS * KC ds_read_b32 per lane — gathering the fragments
    // straight from global memory (16 cache lines per load instruction) cost 5.5 us per launch.
    constexpr int WROW = TAPS * CIN, WRS = WROW + 4;
    {
        const float* wsrc = wb + (size_t)co0 * WROW;
        for (int i = t; i < 16 * WROW / 4; i += 256) {
            const int row = i / (WROW / 4), c4 = i - row * (WROW / 4);
            *reinterpret_cast<float4*>(xs + row * WRS + c4 * 4) = *reinterpret_cast<const float4*>(wsrc + (size_t)row * WROW + c4 * 4);
        }
    }
    __syncthreads();
    float bf[TAPS][KC];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) bf[tap][kc] = xs[li * WRS + tap * CIN + 4 * kc + lg];
    const float bv = bias != nullptr ? bias[co0 + li] : 0.f;
    __syncthreads();                                              // the filter slice is in registers: the buffer becomes the input tile
    const int tile = blockIdx.x;
    const int n = tile / a.tiles_per_img, ho0 = (tile - n * a.tiles_per_img) * a.TH;
    const int th = (a.Hout - ho0 < a.TH) ? (a.Hout - ho0) : a.TH;
    // staging: four 16-B loads in flight per thread before the first LDS store
    const int nx = a.XH * a.XW * KC;
    for (int i0 = t; i0 < nx; i0 += 4 * 256) {
        float4 v[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256 < nx ? i0 + u * 256 : t;
            const int c4 = i % KC, p = i / KC;
            const int xr = p / a.XW, xc = p - xr * a.XW;
            const int hi = ho0 - a.ph + xr, wi = xc - a.pw;
            ok[u] = hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win;
            const int hic = hi < 0 ? 0 : (hi >= a.Hin ? a.Hin - 1 : hi), wic = wi < 0 ? 0 : (wi >= a.Win ? a.Win - 1 : wi);
            v[u] = *reinterpret_cast<const float4*>(x + (((size_t)n * a.Hin + hic) * a.Win + wic) * CIN + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            if (i < nx) *reinterpret_cast<float4*>(xs + (size_t)(i / KC) * a.PS + (i % KC) * 4) = ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    const int spr = a.Wout >> 4;                                  // 16-pixel segments per output row
    const int nseg = th * spr;
    float ssum = 0.f, ssq = 0.f;                                  // this lane's channel (li) over its pixels: BatchNorm statistics partials
    for (int sg = wave; sg < nseg; sg += 4) {
        const int hl = sg / spr, w0 = (sg - hl * spr) << 4;
        const float* ap = xs + ((size_t)hl * a.XW + w0 + li) * a.PS + lg;     // A'[i = pixel][k = ci]
        // all A operands of the segment are read from LDS before the first MFMA (an MFMA waiting on its own ds_read is a chain of LDS latencies), and
        // two accumulators alternate so that consecutive MFMAs do not depend on each other
        float av[TAPS][KC];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = ((tap / a.S) * a.XW + tap % a.S) * a.PS;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) av[tap][kc] = ap[toff + 4 * kc];
        }
        __builtin_amdgcn_sched_barrier(0);                         // keep the reads up here: the scheduler otherwise sinks each one next to its MFMA
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                if ((tap * KC + kc) & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap][kc], bf[tap][kc], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap][kc], bf[tap][kc], acc, 0, 0, 0);
            }
        acc += acc2;
        float* yp = y + (((size_t)n * a.Hout + ho0 + hl) * a.Wout + w0 + 4 * lg) * a.Cout + co0 + li;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float v = dcv_act_apply(acc[reg] + bv, a.act, a.slope);
            yp[(size_t)reg * a.Cout] = v;
            ssum += v; ssq += v * v;
        }
    }
    if (stats != nullptr) {
        __shared__ float red[4][2][16];
        ssum += __shfl_xor(ssum, 16, 64); ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 16, 64); ssq += __shfl_xor(ssq, 32, 64);
        if (lg == 0) { red[wave][0][li] = ssum; red[wave][1][li] = ssq; }
        __syncthreads();
        if (t < 32) {
            const int q = t >> 4, c = t & 15;
            const float v = (red[0][q][c] + red[1][q][c]) + (red[2][q][c] + red[3][q][c]);
            stats[((size_t)q * a.Cout + co0 + c) * gridDim.x + blockIdx.x] = (double)v;
        }
    }
}

static bool halo_plan(const MfmaArgs& m, int math, HaloArgs* h, size_t* smem) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_MFMA_HALO"); enabled = e ? atoi(e) : 1; }
    if (!enabled || math != DCV_MATH_FP32 || m.phase) return false;
    if (m.sh != 1 || m.sw != 1 || m.dh != 1 || m.dw != 1) return false;
    if ((m.Cin != 16 && m.Cin != 32) || m.Cout % 16 != 0 || m.Wout % 16 != 0) return false;
    const int taps = m.R * m.S;
    if (!(taps == 9 || taps == 1 || (taps == 25 && m.Cin == 16))) return false;
    h->N = m.N; h->Hin = m.Hin; h->Win = m.Win; h->Cin = m.Cin; h->Hout = m.Hout; h->Wout = m.Wout; h->Cout = m.Cout; h->R = m.R; h->S = m.S;
    h->ph = m.ph; h->pw = m.pw; h->act = m.act; h->slope = m.slope;
    h->PS = m.Cin + 4;
    h->XW = m.Wout + m.S - 1;
    int th = m.Hout;
    size_t bytes;
    for (;;) {
        h->TH = th; h->XH = th + m.R - 1;
        bytes = (size_t)h->XH * h->XW * h->PS * sizeof(float);
        if (th == 1 || bytes <= 48 * 1024) break;
        th = (th + 1) / 2;
    }
    if (bytes > 64 * 1024) return false;
    const size_t wbytes = (size_t)16 * (taps * m.Cin + 4) * sizeof(float);     // the buffer first holds the block's filter slice
    if (bytes < wbytes) bytes = wbytes;
    h->tiles_per_img = (m.Hout + th - 1) / th;
    h->total_tiles = m.N * h->tiles_per_img;
    *smem = bytes;
    return true;
}

// the window kernel's plan: fp32 3x3 / stride 1 / pad 1 / no dilation on the whole tensor (not a phase of a strided data gradient), 64-aligned output channels,
// 32-bit buffer offsets, at least one block per CU.  DCV_MFMA_WIN=0 switches it off, =2 forces it for every shape it can run.
static bool win_plan(const MfmaArgs& m, int math, WinArgs* w, size_t* smem, int* bstride) {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("DCV_MFMA_WIN"); mode = e ? atoi(e) : 1; }
    if (mode == 0 || math != DCV_MATH_FP32 || m.phase || m.R != 3 || m.S != 3 || m.sh != 1 || m.sw != 1 || m.ph != 1 || m.pw != 1 || m.dh != 1 || m.dw != 1) return false;
    if (m.Hin != m.Hout || m.Win != m.Wout || m.Cout % 64 != 0 || m.Cin % 16 != 0 || m.Win > 64 || m.Win < 2 || m.Hin < 2) return false;
    if ((size_t)m.M * m.Cin * 4 >= ((size_t)1 << 31) || (size_t)m.Cout * 9 * m.Cin * 4 >= ((size_t)1 << 31)) return false;
    const int tiles = (int)((m.M + 255) / 256);
    if (mode != 2 && (long)tiles * (m.Cout / 64) < DCV_NUM_CU) return false;
    w->N = m.N; w->H = m.Hin; w->W = m.Win; w->Cin = m.Cin; w->Cout = m.Cout; w->act = m.act; w->slope = m.slope; w->M = (int)m.M;
    w->x_bytes = (unsigned)((size_t)m.M * m.Cin * 4); w->w_bytes = (unsigned)((size_t)m.Cout * 9 * m.Cin * 4);
    w->nrows = 256 + 2 * m.Win + 2; w->ninstr = (w->nrows + 11) / 12; w->tiles = tiles;
    w->magic_w = 0xFFFFFFFFu / (unsigned)m.Win + 1u; w->magic_h = 0xFFFFFFFFu / (unsigned)m.Hin + 1u;
    const int buf = MW_ROWB + (12 * w->ninstr + 1) * MW_ROWB;      // zero row + the rows the DMA instructions cover
    *bstride = buf <= 26624 ? 26624 : 32768;                       // (W <= 32: three blocks per CU; W <= 64: 31840 B)
    if (buf > *bstride) return false;
    *smem = (size_t)*bstride + buf;
    return true;
}

template <bool BF16>
static void launch_mfma(int bn, int bmr, dim3 grid, hipStream_t st, const MfmaArgs& a, const float* x, const float* wb, const float* bias, float* y, double* stats) {
    dcv_count_launch(DCV_FAMILY_F32_IGEMM);
    if (bmr == 4) hipLaunchKernelGGL((conv_mfma_kernel<64, 4, BF16>), grid, dim3(256), 0, st, a, x, wb, bias, y, stats);
    else switch (bn) {
        case 64: hipLaunchKernelGGL((conv_mfma_kernel<64, 2, BF16>), grid, dim3(256), 0, st, a, x, wb, bias, y, stats); break;
        case 32: hipLaunchKernelGGL((conv_mfma_kernel<32, 2, BF16>), grid, dim3(256), 0, st, a, x, wb, bias, y, stats); break;
        default: hipLaunchKernelGGL((conv_mfma_kernel<16, 2, BF16>), grid, dim3(256), 0, st, a, x, wb, bias, y, stats); break;
    }
}

static int run_mfma(const MfmaArgs& a0, const float* x, const float* wb, const float* bias, float* y, hipStream_t st, double* stats = nullptr, int math = DCV_MATH_FP32) {
    MfmaArgs a = a0;
    const bool small = a.Wout >= 2 && a.Hout >= 2 && (size_t)a.M * (size_t)(a.Wout > a.Hout ? a.Wout : a.Hout) < ((size_t)1 << 32);
    a.magic_w = small ? 0xFFFFFFFFu / (unsigned)a.Wout + 1u : 0u; a.magic_h = small ? 0xFFFFFFFFu / (unsigned)a.Hout + 1u : 0u;
    HaloArgs h{}; size_t hsm;
    if (halo_plan(a, math, &h, &hsm)) {
        dim3 grid((unsigned)h.total_tiles, (unsigned)(a.Cout / 16));
        const int taps = a.R * a.S;
        dcv_count_launch(DCV_FAMILY_F32_HALO);
#define DCV_HALO(C_, T_) hipLaunchKernelGGL((conv_mfma_halo_kernel<C_, T_>), grid, dim3(256), hsm, st, h, x, wb, bias, y, stats)
        if (a.Cin == 16) { if (taps == 9) DCV_HALO(16, 9); else if (taps == 25) DCV_HALO(16, 25); else DCV_HALO(16, 1); }
        else { if (taps == 9) DCV_HALO(32, 9); else DCV_HALO(32, 1); }
#undef DCV_HALO
        return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
    }
    WinArgs wa{}; size_t wsm; int wbs;
    if (win_plan(a, math, &wa, &wsm, &wbs)) {
        const dim3 wgrid((unsigned)wa.tiles, (unsigned)(a.Cout / 64));
        const bool plain = a.act == DCV_ACT_NONE || a.act == DCV_ACT_RELU || a.act == DCV_ACT_LEAKY_RELU;
        dcv_count_launch(DCV_FAMILY_F32_WINDOW);
#define DCV_WIN(P_, B_) hipLaunchKernelGGL((conv_mfma_win_kernel<P_, B_>), wgrid, dim3(256), wsm, st, wa, x, wb, bias, y, stats)
        if (wbs == 26624) { if (plain) DCV_WIN(true, 26624); else DCV_WIN(false, 26624); }
        else { if (plain) DCV_WIN(true, 32768); else DCV_WIN(false, 32768); }
#undef DCV_WIN
        return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
    }
    const int bn = pick_bn(a.Cout);
    const int bmr = pick_bmr(a.M, a.Cin, a.Cout, bn);
    dim3 grid((unsigned)dcv_ceil_div((size_t)a.M, (size_t)64 * bmr), (unsigned)(a.Cout / bn));
    if (math == DCV_MATH_BF16) launch_mfma<true>(bn, bmr, grid, st, a, x, wb, bias, y, stats);
    else launch_mfma<false>(bn, bmr, grid, st, a, x, wb, bias, y, stats);
    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
}

int dcv_mfma_conv2d_fwd_stats_rows(const dcv_conv2d_desc* d, int* rows_per_sample) {
    if (rows_per_sample) *rows_per_sample = 0;
    {   // the halo kernel emits one row of partials per tile (a tile = rows of ONE image)
        MfmaArgs a{};
        a.N = d->n; a.Hin = d->h; a.Win = d->w; a.Cin = d->cin; a.Hout = d->ho; a.Wout = d->wo; a.Cout = d->cout;
        a.R = d->r; a.S = d->s; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
        HaloArgs h{}; size_t hsm;
        if (halo_plan(a, d->math, &h, &hsm)) { if (rows_per_sample) *rows_per_sample = h.tiles_per_img; return h.total_tiles; }
    }
    const long M = (long)d->n * d->ho * d->wo;
    bool win = false;
    {   // the window kernel: 256-pixel tiles
        MfmaArgs a{};
        a.N = d->n; a.Hin = d->h; a.Win = d->w; a.Cin = d->cin; a.Hout = d->ho; a.Wout = d->wo; a.Cout = d->cout;
        a.R = d->r; a.S = d->s; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w; a.M = M;
        WinArgs wa{}; size_t wsm; int wbs;
        win = win_plan(a, d->math, &wa, &wsm, &wbs);
    }
    const int tile = win ? 256 : 64 * pick_bmr(M, d->cin, d->cout, pick_bn(d->cout));
    if (rows_per_sample && (d->ho * d->wo) % tile == 0) *rows_per_sample = d->ho * d->wo / tile;    // pixel tiles which never straddle images
    return (int)dcv_ceil_div((size_t)M, (size_t)tile);
}

int dcv_mfma_conv2d_fwd(const dcv_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, hipStream_t st, bool prepared, double* stats) {
    if (!ws || ws_bytes < dcv_mfma_conv2d_ws(d, 0)) return DCV_ERR_WORKSPACE;
    if (!aligned16(x) || !aligned16(ws)) return DCV_ERR_INVALID_ARG;
    float* wb = reinterpret_cast<float*>(ws);
    const int total = d->cout * d->cin * d->r * d->s;
    if (!prepared) hipLaunchKernelGGL(mfma_prep_weights_kernel, dim3((unsigned)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256)), dim3(256), 0, st, w, wb, d->cout, d->cin, d->r, d->s, d->cin, 0);
    MfmaArgs a{};
    a.N = d->n; a.Hin = d->h; a.Win = d->w; a.Cin = d->cin; a.Hout = d->ho; a.Wout = d->wo; a.Cout = d->cout;
    a.R = d->r; a.S = d->s; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    a.act = d->act; a.slope = d->act_slope; a.M = (long)d->n * d->ho * d->wo;
    return run_mfma(a, x, wb, bias, y, st, stats, d->math);
}

namespace {
// Phase a (= output row index mod stride) of a strided data gradient along one axis: dx[stride*i + a] = sum_{r'} dy[i + oa - r'] * w[r0 + stride*r'].
struct PhaseAxis { int r0, taps, pad, count; };      // first matching tap, number of taps, transposed padding of the stride-1 problem, outputs in this phase
inline PhaseAxis phase_axis(int a, int size, int R, int stride, int pad) {
    PhaseAxis p{};
    p.r0 = (a + pad) % stride;
    p.taps = p.r0 < R ? (R - p.r0 + stride - 1) / stride : 0;
    const int oa = (a + pad - p.r0) / stride;
    p.pad = p.taps - 1 - oa;
    p.count = a < size ? (size - a + stride - 1) / stride : 0;
    return p;
}
}  // namespace

int dcv_mfma_conv2d_prep_entries(const dcv_conv2d_desc* d, int op, dcv_prep_entry* out) {
    dcv_prep_entry e{};
    e.kind = 1; e.w_cin = d->cin; e.dgrad = op ? 1 : 0;
    e.kcout = op ? d->cin : d->cout; e.kcin = op ? d->cout : d->cin;
    e.Rw = d->r; e.Sw = d->s;
    if (op == 0 || (d->stride_h == 1 && d->stride_w == 1)) {
        e.R = d->r; e.S = d->s; e.r0 = e.s0 = 0; e.rstep = e.sstep = 1; e.dst_off = 0;
        e.total = d->cout * d->cin * d->r * d->s;
        out[0] = e;
        return 1;
    }
    int n = 0, off = 0;
    for (int a = 0; a < d->stride_h; ++a)
        for (int b = 0; b < d->stride_w; ++b) {
            const PhaseAxis ph = phase_axis(a, d->h, d->r, d->stride_h, d->pad_h), pw = phase_axis(b, d->w, d->s, d->stride_w, d->pad_w);
            dcv_prep_entry q = e;
            q.R = ph.taps; q.S = pw.taps; q.r0 = ph.r0; q.s0 = pw.r0; q.rstep = d->stride_h; q.sstep = d->stride_w;
            q.dst_off = off; q.total = d->cout * d->cin * ph.taps * pw.taps;
            off += q.total;
            out[n++] = q;
        }
    return n;
}

int dcv_mfma_conv2d_dgrad(const dcv_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes, hipStream_t st, bool prepared) {
    if (!ws || ws_bytes < dcv_mfma_conv2d_ws(d, 1)) return DCV_ERR_WORKSPACE;
    if (!aligned16(dy) || !aligned16(ws)) return DCV_ERR_INVALID_ARG;
    float* wb = reinterpret_cast<float*>(ws);
    dcv_prep_entry pieces[DCV_PREP_PIECES];
    const int np = dcv_mfma_conv2d_prep_entries(d, 1, pieces);
    if (!prepared) {
        for (int q = 0; q < np; ++q) {
            if (pieces[q].total == 0) continue;
            const dcv_prep_entry& e = pieces[q];
            // same element mapping as the one-launch multi-filter kernel of conv2d.hip (dcv_prep_element), one launch per piece here
            hipLaunchKernelGGL(mfma_prep_piece_kernel, dim3((unsigned)((e.total + 255) / 256 > 1024 ? 1024 : (e.total + 255) / 256)), dim3(256), 0, st, e, w, wb + e.dst_off);
        }
    }
    // kernel view: reads dy (Cin' = Cout), writes dx (Cout' = Cin)
    MfmaArgs a{};
    a.N = d->n; a.Hin = d->ho; a.Win = d->wo; a.Cin = d->cout; a.Cout = d->cin;
    a.sh = 1; a.sw = 1; a.dh = d->dil_h; a.dw = d->dil_w; a.act = DCV_ACT_NONE; a.slope = 0.f;
    if (d->stride_h == 1 && d->stride_w == 1) {
        // filter flipped + transposed, pad' = dil*(R-1) - pad
        a.Hout = d->h; a.Wout = d->w; a.R = d->r; a.S = d->s;
        a.ph = d->dil_h * (d->r - 1) - d->pad_h; a.pw = d->dil_w * (d->s - 1) - d->pad_w; a.M = (long)d->n * d->h * d->w;
        return run_mfma(a, dy, wb, nullptr, dx, st, nullptr, d->math);
    }
    bool any_empty = false;
    for (int q = 0; q < np; ++q) any_empty = any_empty || pieces[q].total == 0;
    if (any_empty && hipMemsetAsync(dx, 0, (size_t)d->n * d->h * d->w * d->cin * sizeof(float), st) != hipSuccess) return DCV_ERR_LAUNCH;
    int q = 0;
    for (int pa = 0; pa < d->stride_h; ++pa)
        for (int pb = 0; pb < d->stride_w; ++pb, ++q) {
            const PhaseAxis ph = phase_axis(pa, d->h, d->r, d->stride_h, d->pad_h), pw = phase_axis(pb, d->w, d->s, d->stride_w, d->pad_w);
            if (ph.taps == 0 || pw.taps == 0 || ph.count == 0 || pw.count == 0) continue;
            MfmaArgs p = a;
            p.Hout = ph.count; p.Wout = pw.count; p.R = ph.taps; p.S = pw.taps; p.ph = ph.pad; p.pw = pw.pad; p.dh = p.dw = 1;
            p.M = (long)d->n * ph.count * pw.count;
            p.phase = 1; p.oh_mul = d->stride_h; p.ow_mul = d->stride_w; p.oh_off = pa; p.ow_off = pb; p.Hfull = d->h; p.Wfull = d->w;
            const int rc = run_mfma(p, dy, wb + pieces[q].dst_off, nullptr, dx, st, nullptr, d->math);
            if (rc != DCV_OK) return rc;
        }
    return DCV_OK;
}

template <int PS, bool BF16, bool PIPE, bool IN16 = false>
static int launch_wgrad9(const Wg9Args& a, dim3 grid, size_t smem, hipStream_t st, const float* x, const float* dy, float* partial) {
    static bool configured = false;            // dynamic LDS above 64 KiB needs the attribute once per instantiation
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_wgrad9_kernel<PS, BF16, PIPE, IN16>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return DCV_ERR_LAUNCH;
        configured = true;
    }
    if (!IN16) dcv_count_launch(DCV_FAMILY_F32_WGRAD9);             // (IN16: counted by dcv_conv2d_bf16_wgrad as the fp32-staging weight gradient of bf16 tensors)
    hipLaunchKernelGGL((conv_mfma_wgrad9_kernel<PS, BF16, PIPE, IN16>), grid, dim3(256), smem, st, a, x, dy, partial);
    return DCV_OK;
}

// dw[i] = sum over chunks of partial[chunk][i], chunks in order (slabs whose layout already is dw's: the one-tap weight gradient)
int dcv_mfma_wgrad_reduce_plain(const float* partial, float* dw, int chunks, size_t wtotal, hipStream_t st) {
    hipLaunchKernelGGL(mfma_wgrad_reduce_kernel, dim3((unsigned)dcv_ceil_div(wtotal, 32)), dim3(1024), 0, st, partial, dw, chunks, wtotal);
    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
}

int dcv_mfma_wgrad9_reduce(const float* partial, float* dw, int chunks, int cout, int cin, hipStream_t st) {
    if (chunks <= 16 && cin % 64 == 0 && cout <= 65535 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0) {
        if (cin % 128 == 0) hipLaunchKernelGGL(mfma_wgrad9_reduce_few_kernel<128>, dim3((unsigned)(cin / 128), (unsigned)cout), dim3(320), 0, st, partial, dw, chunks, cout, cin);
        else hipLaunchKernelGGL(mfma_wgrad9_reduce_few_kernel<64>, dim3((unsigned)(cin / 64), (unsigned)cout), dim3(176), 0, st, partial, dw, chunks, cout, cin);
        return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(mfma_wgrad9_reduce_kernel, dim3((unsigned)dcv_ceil_div((size_t)9 * cout * cin / 4, 8)), dim3(256), 0, st, partial, dw, chunks, cout, cin);
    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
}

// bf16 storage path: x and dy are bf16 tensors; only the layers the all-taps kernel plans (3x3, stride 1, pad 1, 64-aligned channels)
bool dcv_mfma_conv2d_wgrad_in16_supported(const dcv_conv2d_desc* d) {
    Wg9Args a9{}; int ps9, chunks9; size_t sm9;
    return d->groups == 1 && d->x_layout == DCV_LAYOUT_NHWC && wg9_plan(d, &a9, &ps9, &chunks9, &sm9);
}

int dcv_mfma_conv2d_wgrad_in16(const dcv_conv2d_desc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
    Wg9Args a9{}; int ps9, chunks9; size_t sm9;
    if (!wg9_plan(d, &a9, &ps9, &chunks9, &sm9)) return DCV_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < (size_t)chunks9 * a9.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(x) & 7) || (reinterpret_cast<uintptr_t>(dy) & 7)) return DCV_ERR_INVALID_ARG;
    float* partial = reinterpret_cast<float*>(ws);
    dim3 grid((unsigned)chunks9, (unsigned)(d->cin / WG_T), (unsigned)(d->cout / WG_T));
    if (grid.y > 65535 || grid.z > 65535) return DCV_ERR_UNSUPPORTED;
    const float* xf = reinterpret_cast<const float*>(x); const float* dyf = reinterpret_cast<const float*>(dy);
    const int rc = ps9 == 32 ? launch_wgrad9<32, true, false, true>(a9, grid, sm9, st, xf, dyf, partial) : launch_wgrad9<16, true, false, true>(a9, grid, sm9, st, xf, dyf, partial);
    if (rc != DCV_OK) return rc;
    return dcv_mfma_wgrad9_reduce(partial, dw, chunks9, d->cout, d->cin, st);
}

int dcv_mfma_conv2d_wgrad(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
    {
        Wg9Args a9{}; int ps9, chunks9; size_t sm9;
        if (wg9_plan(d, &a9, &ps9, &chunks9, &sm9)) {
            if (!ws || ws_bytes < (size_t)chunks9 * a9.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
            if (!aligned16(x) || !aligned16(dy)) return DCV_ERR_INVALID_ARG;
            float* partial = reinterpret_cast<float*>(ws);
            dim3 grid((unsigned)chunks9, (unsigned)(d->cin / WG_T), (unsigned)(d->cout / WG_T));
            if (grid.y > 65535 || grid.z > 65535) return DCV_ERR_UNSUPPORTED;
            const bool bf = d->math == DCV_MATH_BF16;
            int rc;
            const bool pipe = wg9_mode() == 1 && !bf;
            if (ps9 == 32) rc = bf ? launch_wgrad9<32, true, false>(a9, grid, sm9, st, x, dy, partial) : (pipe ? launch_wgrad9<32, false, true>(a9, grid, sm9, st, x, dy, partial) : launch_wgrad9<32, false, false>(a9, grid, sm9, st, x, dy, partial));
            else rc = bf ? launch_wgrad9<16, true, false>(a9, grid, sm9, st, x, dy, partial) : (pipe ? launch_wgrad9<16, false, true>(a9, grid, sm9, st, x, dy, partial) : launch_wgrad9<16, false, false>(a9, grid, sm9, st, x, dy, partial));
            if (rc != DCV_OK) return rc;
            return dcv_mfma_wgrad9_reduce(partial, dw, chunks9, d->cout, d->cin, st);
        }
    }
    MfmaWgArgs a{}; int chunks;
    wg_plan(d, &a, &chunks);
    if (!ws || ws_bytes < (size_t)chunks * a.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
    if (!aligned16(x) || !aligned16(dy)) return DCV_ERR_INVALID_ARG;
    float* partial = reinterpret_cast<float*>(ws);
    dim3 grid((unsigned)chunks, (unsigned)(d->r * d->s * (d->cin / WG_T)), (unsigned)(d->cout / WG_T));
    if (grid.y > 65535 || grid.z > 65535) return DCV_ERR_UNSUPPORTED;
    dcv_count_launch(DCV_FAMILY_F32_WGRAD_TAP);
    if (d->math == DCV_MATH_BF16) hipLaunchKernelGGL((conv_mfma_wgrad_kernel<true>), grid, dim3(256), 0, st, a, x, dy, partial);
    else hipLaunchKernelGGL((conv_mfma_wgrad_kernel<false>), grid, dim3(256), 0, st, a, x, dy, partial);
    hipLaunchKernelGGL(mfma_wgrad_reduce_kernel, dim3((unsigned)dcv_ceil_div(a.wtotal, 32)), dim3(1024), 0, st, partial, dw, chunks, a.wtotal);
    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
}
