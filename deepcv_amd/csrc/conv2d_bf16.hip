// conv2d_bf16.hip — convolution with bf16 STORAGE: activations and prepared filters are bf16 in HBM and in LDS, the matrix cores run
// `v_mfma_f32_16x16x32_bf16` (gfx950: K = 32 per instruction, 16 cycles), accumulation / bias / statistics are fp32 / fp64, master weights stay fp32.
// This is the "bf16 MFMA conv" of BASELINE configs 3 / 5; the fp32 kernels (conv2d_mfma.hip) remain the parity mode.
//
// Forward as implicit GEMM:  D[m][n] = sum_k A[m][k] B[k][n],  m = output pixel (n, ho, wo), n = output channel, k = (tap, ci), ci fastest.
//   * block = 256 threads = 4 waves = WM x WN, tile = 128 pixels x (64 WN) channels, K step = 64 input channels of one filter tap;
//   * both operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 16 B per lane, no VGPR round trip, no `ds_write`): a row of the A image is
//     the 128 B (64 channels) of one input pixel of the tap, a row of the B image the 128 B of one output channel of the prepared filter [tap][co][ci].
//     An LDS-DMA image is lane-linear, so the bank swizzle is applied on the SOURCE side: LDS 16-B slot s' of row r holds logical slot s' ^ ((r >> 1) & 7);
//     with that image the four 16-lane groups of a `ds_read_b128` fragment read (rows 16i + li, logical slot 4 ks + lg) touch 16 distinct 16-B bank groups.
//   * padding and tile tails cost nothing: an out-of-image pixel gets a buffer offset beyond `num_records`, for which the hardware delivers zeros;
//     per row the validity of all taps is one 64-bit mask computed once, per K step the address work is a shift, a select and an add per 16-B piece;
//   * two LDS stages: the DMA of K step c + 1 is in flight while the MFMAs of step c run; one barrier per step; two blocks per CU hide each other's waits;
//   * output channels are permuted between the B image and the accumulator tiles (tile j, column c <-> channel 4 c + j), so a lane ends up with FOUR
//     CONSECUTIVE channels of a pixel: 8-B (bf16) or 16-B (fp32) stores, 128 B / 256 B contiguous per pixel across 16 lanes;
//   * epilogue: bias + activation + (optionally) per-block sums of y and y^2 per channel in fp64 ([2][Cout][rows], the layout of the fp32 kernels).
// Data gradient (stride 1) = the same kernel on dy with the transposed + flipped filter; strided data gradients run one launch per output phase.
// Weight gradient: conv2d_mfma.hip's all-taps kernel reading bf16 operands (see dcv_conv2d_bf16_wgrad below).
//
// Roofline: bf16 MFMA (~2.5 PFLOP/s dense).  FLOPs per launch = 2 * N*Ho*Wo * Cout * Cin*R*S.
// Replaces torch.nn.Conv2d forward / backward under autocast-style bf16 (reference src/deepcv/meta/submodule_creators.py:251, ignite_training.py:253).
#include "conv2d_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define BF_BM 128                // pixels per block tile
#define BF_BK 64                 // input channels per K step (128-B LDS rows)
#define BF_OOB 0x80000000u       // buffer offset beyond any num_records: the load delivers zeros

struct Bf16Args {
    int N, Hin, Win, Cin;          // tensor read (fwd: x, dgrad: dy)
    int Hout, Wout, Cout;          // tensor written
    int R, S, sh, sw, ph, pw, dh, dw;
    int act; float slope;
    long M;                        // N * Hout * Wout
    unsigned x_bytes, w_bytes;     // buffer sizes (num_records)
    int phase, oh_mul, ow_mul, oh_off, ow_off, Hfull, Wfull;      // strided data gradient: this launch writes phase (oh_off, ow_off) of the full tensor
    unsigned magic_w, magic_h;     // the window kernel: floor((2^32 - 1) / Win) + 1, ... / Hin + 1: n / d = mulhi(n, magic) for n d < 2^32
    int pix;                       // elements between neighbouring pixels of the tensor read (Cin; the folded first layer: the real channel count, see fold_rows below)
    int xcd_tiles;                 // the one-tile-per-block window kernel: > 0 = pixel tiles of the launch, blocks mapped so that a tile's channel tiles share an XCD (below)
    int stat_rows, stat_row0;      // a launch over a SLICE of the batch (tensors beyond 2 GiB run as several launches): rows of the whole statistics array (0: this launch's own) and this slice's first row
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int bw_i32x4 __attribute__((ext_vector_type(4)));
// LDS-DMA from inline assembly (see conv2d_mfma.hip: the compiler must not track it as an LDS store); the kernel waits itself (s_waitcnt builtin) before its barrier
__device__ __forceinline__ void bw_dma16(bw_i32x4 rsrc0, unsigned lds_addr, unsigned voffset, unsigned soffset) {
    // (under scalar-register pressure the compiler keeps the uniform resource words in VGPRs, which the instruction cannot take)
    const bw_i32x4 rsrc = {__builtin_amdgcn_readfirstlane(rsrc0.x), __builtin_amdgcn_readfirstlane(rsrc0.y), __builtin_amdgcn_readfirstlane(rsrc0.z), __builtin_amdgcn_readfirstlane(rsrc0.w)};
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voffset), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soffset)) : "memory");
}


__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, (int)voffset, 0, 0, 0);
}

// WM x WN waves; wave tile = (16 TI) pixels x 64 channels; BM = 16 TI WM = 128
template <int WM, int WN, bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(Bf16Args a, const u16* __restrict__ x, const u16* __restrict__ wb, const float* __restrict__ bias,
                                                            void* __restrict__ yv, double* __restrict__ stats) {
    constexpr int TI = BF_BM / (16 * WM);
    constexpr int BN = 64 * WN;
    constexpr int PA = BF_BM / 32, PB = BN / 32;                  // 16-B pieces per thread and stage
    constexpr int A_BYTES = BF_BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;
    // (XCD-aware block order when the launch has several channel tiles: see conv_bf16_halo_kernel — a pixel tile's channel tiles read the same pixels)
    int bx = blockIdx.x, by = blockIdx.y, gx = gridDim.x;
    if (a.xcd_tiles > 0) {
        const int cts = a.Cout / BN, L = (int)blockIdx.x, grp = L / (8 * cts), r = L - grp * 8 * cts;
        bx = grp * 8 + (r & 7); by = r >> 3; gx = a.xcd_tiles;
        if (bx >= gx) return;
    }
    const long m0 = (long)bx * BF_BM;
    const int n0 = by * BN;

    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)a.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (int)a.w_bytes, 0x00020000);

    // ---- staging roles: piece p of this thread = LDS row 32 p + (t >> 3), LDS slot t & 7, which holds logical slot (t & 7) ^ ((t >> 4) & 7)
    const int lslot = (t & 7) ^ ((t >> 4) & 7);
    int offa[PA]; unsigned long long maska[PA];
    const int taps = a.R * a.S;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const long m = m0 + 32 * p + (t >> 3);
        const bool ok = m < a.M;
        const long mm = ok ? m : 0;
        int wo, ho, n;
        if (a.magic_w != 0u) {       // (uniform) 32-bit decode by multiply-high: three 64-bit divisions per row were several microseconds of every block's prologue
            const unsigned um = (unsigned)mm, r1 = __umulhi(um, a.magic_w), nn = __umulhi(r1, a.magic_h);
            wo = (int)(um - r1 * (unsigned)a.Wout); ho = (int)(r1 - nn * (unsigned)a.Hout); n = (int)nn;
        } else { wo = (int)(mm % a.Wout); const long r1 = mm / a.Wout; ho = (int)(r1 % a.Hout); n = (int)(r1 / a.Hout); }
        const int hi0 = ho * a.sh - a.ph, wi0 = wo * a.sw - a.pw;
        offa[p] = (((n * a.Hin + hi0) * a.Win + wi0) * a.pix) * 2 + lslot * 16;      // may be negative for border pixels: only used under its validity bit
        unsigned long long mk = 0;
        if (ok)
            for (int r = 0; r < a.R; ++r) {
                const int hi = hi0 + r * a.dh;
                for (int s = 0; s < a.S; ++s) {
                    const int wi = wi0 + s * a.dw;
                    if (hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win) mk |= 1ull << (r * a.S + s);
                }
            }
        maska[p] = mk;
    }
    int offb[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int rho = 32 * p + (t >> 3);                        // LDS row of the B image: 64 wn' + 16 j + c  <-  channel n0 + 64 wn' + 4 c + j
        const int co = n0 + (rho & ~63) + 4 * (rho & 15) + ((rho >> 4) & 3);
        offb[p] = (co * a.Cin) * 2 + lslot * 16;
    }

    const int cchunks = a.Cin / BF_BK;
    const int nk = taps * cchunks;
    // position of the next K step to LOAD (scalar state)
    int l_tap = 0, l_chunk = 0, l_r = 0, l_s = 0;
    auto stage = [&](int buf) {
        char* base = lds + buf * STAGE;
        const int ua = ((l_r * a.dh) * a.Win + l_s * a.dw) * a.pix * 2 + l_chunk * 128;
        const int ub = (l_tap * a.Cout) * a.Cin * 2 + l_chunk * 128;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const bool v = (maska[p] >> l_tap) & 1ull;
            dma16(rx, base + (p * 256 + (t & ~63)) * 16, v ? (unsigned)(offa[p] + ua) : BF_OOB);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) dma16(rw, base + A_BYTES + (p * 256 + (t & ~63)) * 16, (unsigned)(offb[p] + ub));
        if (++l_chunk == cchunks) { l_chunk = 0; ++l_tap; if (++l_s == a.S) { l_s = 0; ++l_r; } }
    };

    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment addresses: row stride 128 B, slot (4 ks + lg) ^ ((row >> 1) & 7); rows 16 i + li of the wave's range -> swizzle term (li >> 1) & 7
    const int swz = (li >> 1) & 7;
    const int a_row0 = (16 * TI * wm + li) * 128, b_row0 = A_BYTES + (64 * wn + li) * 128;
    const int s0 = ((0 + lg) ^ swz) * 16, s1 = ((4 + lg) ^ swz) * 16;

    stage(0);
    for (int c = 0; c < nk; ++c) {
        const int buf = c & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of step c have landed ...
        __syncthreads();                                           // ... and everybody's; every wave is also done reading the other stage
        if (c + 1 < nk) stage(buf ^ 1);
        const char* sb = lds + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = ks ? s1 : s0;
            bf16x8 af[TI], bf[4];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_row0 + i * 2048 + so);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + b_row0 + j * 2048 + so);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: acc[i][j][reg] -> pixel m0 + 16 TI wm + 16 i + 4 lg + reg, channel n0 + 64 wn + 4 li + j; the activation and the bounds test as compile-time
    // cases (round 4, as in the window kernel: the per-element five-way switch was most of the epilogue's instructions)
    const int cbase = n0 + 64 * wn + 4 * li;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + cbase); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    auto epilogue = [&](auto actc, auto fullc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
        constexpr bool FULL = decltype(fullc)::value;
        const int act = ACT >= 0 ? ACT : a.act;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long m = m0 + 16 * TI * wm + 16 * i + 4 * lg + reg;
                if (FULL || m < a.M) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = dcv_act_apply(acc[i][j][reg] + bv[j], act, a.slope);
                        ssum[j] += o[j]; ssq[j] += o[j] * o[j];
                    }
                    size_t off = (size_t)m;
                    if (a.phase) {
                        int jj, ii; long nn;
                        if (a.magic_w != 0u) { const unsigned um = (unsigned)m, r1 = __umulhi(um, a.magic_w), n2 = __umulhi(r1, a.magic_h); jj = (int)(um - r1 * (unsigned)a.Wout); ii = (int)(r1 - n2 * (unsigned)a.Hout); nn = n2; }
                        else { jj = (int)(m % a.Wout); const long r1 = m / a.Wout; ii = (int)(r1 % a.Hout); nn = r1 / a.Hout; }
                        off = ((size_t)nn * a.Hfull + (size_t)ii * a.oh_mul + a.oh_off) * a.Wfull + (size_t)jj * a.ow_mul + a.ow_off;
                    }
                    if constexpr (OUT_BF16) {
                        const bf16x4 h = (bf16x4){(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<u16*>(yv) + off * a.Cout + cbase) = h;
                    } else {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + off * a.Cout + cbase) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
    };
    {
        const bool full = m0 + BF_BM <= a.M;
        if (full && a.act == DCV_ACT_NONE) epilogue(std::integral_constant<int, DCV_ACT_NONE>{}, std::true_type{});
        else if (full && a.act == DCV_ACT_RELU) epilogue(std::integral_constant<int, DCV_ACT_RELU>{}, std::true_type{});
        else epilogue(std::integral_constant<int, -1>{}, std::false_type{});
    }
    if (stats != nullptr) {
        // per-block partial sums: lanes li, li + 16, li + 32, li + 48 hold the same four channels (fixed-order butterfly), then the WM waves of a column
        __syncthreads();                                           // the last K step's fragment reads are done: the stages are free
        double* red = reinterpret_cast<double*>(lds);              // [4 waves][2][64]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s1v = ssum[j], s2v = ssq[j];
            s1v += __shfl_xor(s1v, 16, 64); s2v += __shfl_xor(s2v, 16, 64);
            s1v += __shfl_xor(s1v, 32, 64); s2v += __shfl_xor(s2v, 32, 64);
            if (lg == 0) { red[(wave * 2 + 0) * 64 + 4 * li + j] = (double)s1v; red[(wave * 2 + 1) * 64 + 4 * li + j] = (double)s2v; }
        }
        __syncthreads();
        if (t < 2 * BN) {
            const int q = t / BN, c = t - q * BN;                  // channel c of the block: column group wn' = c / 64, waves wm' * WN + wn'
            const int wnn = c >> 6, cc = c & 63;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) v += red[((w * WN + wnn) * 2 + q) * 64 + cc];
            stats[((size_t)q * a.Cout + n0 + c) * (a.stat_rows ? a.stat_rows : gx) + a.stat_row0 + bx] = v;      // [2][Cout][rows]
        }
    }
}

// ------------------------------------------------------------------------------------------------ round 4: SHORT K (at most four K steps) — the filter in registers
// conv_bf16_kernel above on a layer whose whole reduction is one to four K steps (the folded first layer of a ResNet: four tap rows of 64; a CIFAR first layer as
// im2col + 1x1: one) is all prologue and epilogue: a block lives for four dependent memory round trips and an epilogue, 10 us for 64 MFMAs per wave — 683 us for the
// stem of resnet18_style at 512 images (217 MB in, 822 MB out).  Here the block's 64 output channels x <= 256 reduction values are MFMA B fragments in REGISTERS
// (<= 128 per lane), the blocks are persistent (two per CU), and the 128-pixel x 64-value A tiles of consecutive K steps — of consecutive tiles — flow through a
// ring of four LDS stages requested three steps ahead: no filter traffic, no pipeline restart between tiles, an epilogue every nk steps.
// Same A image (rows of 128 bytes, slot ^ ((row >> 1) & 7)), tap masks, epilogue and statistics rows (one per 128-pixel tile) as conv_bf16_kernel<4, 1>.
#define RFK_STAGES 4
// s_waitcnt vmcnt(n) (expcnt / lgkmcnt untouched) as its immediate: vmcnt = n[3:0] | n[5:4] << 14
#define DCV_VMCNT(n) (0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
__device__ __forceinline__ void rfk_wait(int n) {                   // s_waitcnt vmcnt(n) for n = 8 + 8 e, e = tile epilogues (8 stores each) among the last three steps
    // (exactly the number of younger operations: one too few and the wave waits for the FIRST STORE of the previous tile's epilogue — a trip to memory per tile)
    switch (n) {
        case 8: __builtin_amdgcn_s_waitcnt(DCV_VMCNT(8)); break; case 16: __builtin_amdgcn_s_waitcnt(DCV_VMCNT(16)); break;
        case 24: __builtin_amdgcn_s_waitcnt(DCV_VMCNT(24)); break; case 32: __builtin_amdgcn_s_waitcnt(DCV_VMCNT(32)); break;
        case 17: __builtin_amdgcn_s_waitcnt(DCV_VMCNT(17)); break; case 26: __builtin_amdgcn_s_waitcnt(DCV_VMCNT(26)); break;
        case 35: __builtin_amdgcn_s_waitcnt(DCV_VMCNT(35)); break;
        default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    }
}
template <int ACT>                                                // the activation at compile time (-1: the descriptor's, a five-way switch around each of a lane's 32 results per tile)
__global__ __launch_bounds__(256, 2) void conv_bf16_rfk_kernel(Bf16Args a, const u16* __restrict__ x, const u16* __restrict__ wb, const float* __restrict__ bias,
                                                                u16* __restrict__ yv, double* __restrict__ stats) {
    constexpr int TI = 2, PA = 4, A_BYTES = BF_BM * 128;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.y * 64;
    const int cchunks = a.Cin / BF_BK, taps = a.R * a.S, nk = taps * cchunks;      // 1 .. 4
    const int ntiles = (int)((a.M + BF_BM - 1) / BF_BM), gstep = (int)gridDim.x;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - (int)blockIdx.x + gstep - 1) / gstep : 0, total = my_tiles * nk;
    const unsigned long long xa = (unsigned long long)x;
    const bw_i32x4 rx = {(int)(unsigned)xa, (int)(unsigned)((xa >> 32) & 0xFFFFu), (int)a.x_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)lds + (unsigned)((t & ~63) * 16);

    // ---- the filter: B fragment (step, K half ks, channel tile j) of lane (li, lg) = channel n0 + 4 li + j, reduction values 32 ks + 8 lg .. + 7 of step (tap, chunk)
    bf16x8 fb[4][2][4];
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        const int tap = sidx / cchunks, chunk = sidx - tap * cchunks;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[sidx][ks][j] = *reinterpret_cast<const bf16x8*>(wb + (sidx < nk ? ((size_t)(tap * a.Cout + n0 + 4 * li + j) * a.Cin + chunk * 64 + 32 * ks + 8 * lg) : 0));
    }
    // ---- staging of a tile's A pieces (as conv_bf16_kernel): piece p = LDS row 32 p + (t >> 3), physical slot t & 7 holds logical slot (t & 7) ^ ((t >> 4) & 7)
    const int lslot = (t & 7) ^ ((t >> 4) & 7);
    int offa[PA]; unsigned mka[PA];                                 // of the tile the NEXT request belongs to (taps <= 16: a 32-bit mask)
    auto decode = [&](int tile) __attribute__((always_inline)) {
        const long m0 = (long)tile * BF_BM;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const long m = m0 + 32 * p + (t >> 3);
            const bool ok = m < a.M;
            const unsigned um = (unsigned)(ok ? m : 0), r1 = __umulhi(um, a.magic_w), nn = __umulhi(r1, a.magic_h);
            const int wo = (int)(um - r1 * (unsigned)a.Wout), ho = (int)(r1 - nn * (unsigned)a.Hout), n = (int)nn;
            const int hi0 = ho * a.sh - a.ph, wi0 = wo * a.sw - a.pw;
            offa[p] = (((n * a.Hin + hi0) * a.Win + wi0) * a.pix) * 2 + lslot * 16;
            unsigned mk = 0u;
            if (ok)
                for (int r = 0; r < a.R; ++r) {
                    const int hi = hi0 + r * a.dh;
                    for (int sx = 0; sx < a.S; ++sx) {
                        const int wi = wi0 + sx * a.dw;
                        if (hi >= 0 && hi < a.Hin && wi >= 0 && wi < a.Win) mk |= 1u << (r * a.S + sx);
                    }
                }
            mka[p] = mk;
        }
    };
    int l_g = 0, l_ts = 0, l_tile = blockIdx.x;                    // the next step to REQUEST: global index, step of its tile, its tile
    auto request = [&]() __attribute__((always_inline)) {
        if (l_ts == 0) decode(l_tile);
        const int tap = l_ts / cchunks, chunk = l_ts - tap * cchunks, r = tap / a.S, sx = tap - r * a.S;
        const int ua = ((r * a.dh) * a.Win + sx * a.dw) * a.pix * 2 + chunk * 128;
        const unsigned base = lds0 + (unsigned)((l_g % RFK_STAGES) * A_BYTES);
#pragma unroll
        for (int p = 0; p < PA; ++p)
            bw_dma16(rx, base + (unsigned)(p * 4096), ((mka[p] >> tap) & 1u) ? (unsigned)(offa[p] + ua) : BF_OOB, 0u);      // (inline assembly: the compiler does not wait for it in front of LDS reads)
        ++l_g;
        if (++l_ts == nk) { l_ts = 0; l_tile += gstep; }
    };
    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int swz = (li >> 1) & 7;
    const int a_row0 = (16 * TI * wave + li) * 128;
    const int s0 = ((0 + lg) ^ swz) * 16, s1 = ((4 + lg) ^ swz) * 16;
    const int cbase = n0 + 4 * li;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + cbase); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
    double* red = reinterpret_cast<double*>(lds + RFK_STAGES * A_BYTES);      // [4 waves][2][64]

    for (int k = 0; k < 3; ++k) if (l_g < total) request();
    const int xst = (stats != nullptr && t < 128) ? 1 : 0;           // this wave's statistics store behind a tile's eight
    int tile = blockIdx.x, ts = 0, ends = 0;                        // ends: bit e set = the step e + 1 iterations ago closed a tile (its 8 stores are younger than this step's request)
    for (int g = 0; g < total; ++g) {
        if (g + 2 < total) rfk_wait(8 + (8 + xst) * __builtin_popcount(ends & 7));
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                           // everybody's pieces of step g have landed; every wave is done with the stage of step g - 1
        if (l_g < total) request();                                // step g + 3 into the stage of step g - 1
        const char* sb = lds + (g % RFK_STAGES) * A_BYTES;
        bf16x8 af[2][TI];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TI; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(sb + a_row0 + i * 2048 + (ks ? s1 : s0));
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
            if (sidx == ts) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], fb[sidx][ks][j], acc[i][j], 0, 0, 0);
            }
        ends <<= 1;
        if (++ts < nk) continue;
        // ---- the tile's epilogue (conv_bf16_kernel's): acc[i][j][reg] -> pixel m0 + 32 wave + 16 i + 4 lg + reg, channel cbase + j
        ts = 0; ends |= 1;
        const long m0 = (long)tile * BF_BM;
        float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const long m = m0 + 16 * TI * wave + 16 * i + 4 * lg + reg;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = dcv_act_apply(acc[i][j][reg] + bv[j], ACT >= 0 ? ACT : a.act, a.slope);
                    if (m < a.M) { ssum[j] += o[j]; ssq[j] += o[j] * o[j]; }
                }
                // (the counted waits above assume eight stores per tile and lane: stores are only skipped in the tensor's last tile, which is its block's last)
                if (m < a.M) *reinterpret_cast<bf16x4*>(yv + (size_t)m * a.Cout + cbase) = (bf16x4){(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
            }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (stats != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s1v = ssum[j], s2v = ssq[j];
                s1v += __shfl_xor(s1v, 16, 64); s2v += __shfl_xor(s2v, 16, 64);
                s1v += __shfl_xor(s1v, 32, 64); s2v += __shfl_xor(s2v, 32, 64);
                if (lg == 0) { red[(wave * 2 + 0) * 64 + 4 * li + j] = (double)s1v; red[(wave * 2 + 1) * 64 + 4 * li + j] = (double)s2v; }
            }
            __syncthreads();
            if (t < 128) {
                const int q = t >> 6, c = t & 63;
                double v = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[(w * 2 + q) * 64 + c];
                stats[((size_t)q * a.Cout + n0 + c) * (a.stat_rows ? a.stat_rows : ntiles) + a.stat_row0 + tile] = v;      // [2][Cout][rows]
            }
            // (the next write of `red` is a whole tile — at least one barrier — away)
        }
        tile += gstep;
    }
}

// ------------------------------------------------------------------------------------------------ round 4: the FOLDED first layer on a window staged once
// conv_bf16_rfk_kernel above requests, for every 128-pixel tile of the folded stem, four A tiles of 128 rows x 128 bytes — sixteen LDS-DMA instructions per wave
// for 64 MFMAs, and the ISSUE of those requests (60 - 185 clocks each) sets its time: 403 us of the 690 with every request answered with zeros.  But the rows
// overlap (pixel stride 32 bytes under 128-byte rows): the tile's whole input is its 128 pixels + R - 1 rows of the tensor = ~490 pixels of 32 bytes = 16 KB,
// four requests per wave, and a K step's A fragment is 16 pixels x 64 consecutive bytes of it — lane (li, lg) reads bytes 16 lg .. + 15 behind pixel li, pixels
// 32 bytes apart, which the 16-lane groups of ds_read_b128 serve without bank conflicts (34 distinct 16-byte slots, no two of a group 256 bytes apart).
// Filter in registers (R x 64 x 64 values: 32 R per lane), persistent blocks (two per CU), two window buffers, tiles of 128 output pixels of ONE image (the
// statistics rows' granularity; the pixel count of an image must be a multiple of 128), epilogue of conv_bf16_rfk_kernel.
#define BFW_MAX_PX 1024                                              // pixels of a window buffer (32 KB at 16 channels)
template <int ACT>
__global__ __launch_bounds__(256, 2) void conv_bf16_foldwin_kernel(Bf16Args a, const u16* __restrict__ x, const u16* __restrict__ wb, const float* __restrict__ bias,
                                                                    u16* __restrict__ yv, double* __restrict__ stats, int win_px) {
    constexpr int TI = 2;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.y * 64;
    const int PB = a.pix * 2;                                       // bytes per pixel of the tensor (32)
    const int ntiles = (int)(a.M / BF_BM), gstep = (int)gridDim.x, tpi = a.Hout * a.Wout / BF_BM;      // tiles per image
    const unsigned long long xa = (unsigned long long)x;
    const bw_i32x4 rx = {(int)(unsigned)xa, (int)(unsigned)((xa >> 32) & 0xFFFFu), (int)a.x_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)lds + (unsigned)(t * 16);
    const int wbuf_bytes = win_px * PB;
    // ---- the filter: B fragment (tap row r, K half ks, channel tile j) of lane (li, lg) = channel n0 + 4 li + j, values 32 ks + 8 lg .. + 7 of [r][Cout][64]
    bf16x8 fb[4][2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[r][ks][j] = *reinterpret_cast<const bf16x8*>(wb + (r < a.R ? ((size_t)(r * a.Cout + n0 + 4 * li + j) * 64 + 32 * ks + 8 * lg) : 0));
    // the window of tile `tile`: pixels [q0, q0 + win_px) of the flat tensor, q0 = the tile's first output pixel as an input pixel (stride 1: same row, same column)
    auto tile_q0 = [&](int tile) { const int n = tile / tpi, p0 = (tile - n * tpi) * BF_BM, ho = p0 / a.Wout, wo = p0 - ho * a.Wout; return (n * a.Hin + ho * a.sh) * a.Win + wo * a.sw; };
    auto request = [&](int buf, int tile) {
        const unsigned base = (unsigned)(tile_q0(tile) * PB) + (unsigned)(t * 16);
        for (int k = 0; k * 4096 < wbuf_bytes; ++k) bw_dma16(rx, lds0 + (unsigned)(buf * wbuf_bytes + k * 4096), base + (unsigned)(k * 4096), 0u);
    };
    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int cbase = n0 + 4 * li;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + cbase); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
    double* red = reinterpret_cast<double*>(lds + 2 * wbuf_bytes);   // [2 slots][4 waves][2][64]
    const int nreq = (wbuf_bytes + 4095) / 4096;                     // requests per thread and window

    int pend_tile = -1, pend_par = 0;                              // a tile whose statistics partials wait in red[pend_par] for the next barrier
    bool row_stored = false;                                       // ... and whether the previous iteration stored such a row (one more store younger than this tile's window)
    bool have_out = false;                                         // a tile before this one has stored its output
    auto stats_row = [&](int tl, int pr) {
        if (t < 128) {
            const int q = t >> 6, c = t & 63;
            const double* rk = red + pr * 512;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += rk[(w * 2 + q) * 64 + c];
            stats[((size_t)q * a.Cout + n0 + c) * (a.stat_rows ? a.stat_rows : ntiles) + a.stat_row0 + tl] = v;      // [2][Cout][rows]
        }
    };
    int tile = blockIdx.x, par = 0;
    if (tile < ntiles) request(0, tile);
    for (; tile < ntiles; tile += gstep, par ^= 1) {
        // exactly the operations younger than this tile's window request (rfk_wait): the previous tile's four 16-byte stores and, on two waves, the statistics
        // row which the previous iteration stored (of the tile before it)
        {
            const int young = (have_out ? 4 : 0) + ((row_stored && t < 128) ? 1 : 0);
            if (young == 5) __builtin_amdgcn_s_waitcnt(DCV_VMCNT(5)); else if (young == 4) __builtin_amdgcn_s_waitcnt(DCV_VMCNT(4)); else __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        __syncthreads();                                           // everybody's pieces have landed; every wave is done with the other buffer
        if (tile + gstep < ntiles) request(par ^ 1, tile + gstep);
        row_stored = pend_tile >= 0;
        if (row_stored) { stats_row(pend_tile, pend_par); pend_tile = -1; }
        // operand addresses: tile row i of this wave = output pixel pl = 32 wave + 16 i + li of the tile -> input pixel q - q0 (its row and column in the image)
        const int nimg = tile / tpi, p0 = (tile - nimg * tpi) * BF_BM, ho0 = p0 / a.Wout, wo0 = p0 - ho0 * a.Wout;
        unsigned arow[TI];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int pl = 32 * wave + 16 * i + li, u = wo0 + pl, dh = u / a.Wout, wo = u - dh * a.Wout;
            arow[i] = (unsigned)(par * wbuf_bytes + ((dh * a.sh * a.Win + wo * a.sw) - wo0 * a.sw) * PB + lg * 16);
        }
        const char* lb = lds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < a.R) {
                const unsigned roff = (unsigned)(r * a.dh * a.Win * PB);
                bf16x8 af[2][TI];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < TI; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(lb + arow[i] + roff + ks * 64);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], fb[r][ks][j], acc[i][j], 0, 0, 0);
            }
        }
        // ---- epilogue: acc[i][j][reg] -> pixel m0 + 32 wave + 16 i + 4 lg + reg, channel cbase + j.  The phase-skip builds (tools/variants/foldwin.py: 476 us; 218
        // without this epilogue, 347 without its stores, 396 without the statistics row) say the layer's 822 MB of output are what it waits for: neighbouring lanes
        // swap halves (DPP) so that a lane stores two pixels x 8 channels as 16 bytes, and the waves' statistics partials wait in a second LDS slot for the NEXT
        // tile's opening barrier instead of a barrier of their own (together 476 -> 459 us; stores spread over the next tile's taps: 491, a start skew between
        // the two blocks of a CU: no effect).
        const long m0 = (long)tile * BF_BM;
        float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
        const bool odd = (li & 1) != 0;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            unsigned pk[4][2];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = dcv_act_apply(acc[i][j][reg] + bv[j], ACT >= 0 ? ACT : a.act, a.slope);
                    ssum[j] += o[j]; ssq[j] += o[j] * o[j];
                }
                union { bf16x4 h; unsigned u[2]; } cv; cv.h = (bf16x4){(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                pk[reg][0] = cv.u[0]; pk[reg][1] = cv.u[1];
            }
            // even lanes keep pixels reg 0, 1 and take the odd neighbour's four channels of them; odd lanes keep reg 2, 3
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned s0 = odd ? pk[q][0] : pk[q + 2][0], s1 = odd ? pk[q][1] : pk[q + 2][1];
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, false), r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, false);
                const int reg = odd ? q + 2 : q;
                const long m = m0 + 16 * TI * wave + 16 * i + 4 * lg + reg;
                const uint4 v = odd ? make_uint4(r0, r1, pk[q + 2][0], pk[q + 2][1]) : make_uint4(pk[q][0], pk[q][1], r0, r1);
                *reinterpret_cast<uint4*>(yv + (size_t)m * a.Cout + n0 + 8 * (li >> 1)) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        have_out = true;
        if (stats != nullptr) {
            double* rk = red + (par * 512);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s1v = ssum[j], s2v = ssq[j];
                s1v += __shfl_xor(s1v, 16, 64); s2v += __shfl_xor(s2v, 16, 64);
                s1v += __shfl_xor(s1v, 32, 64); s2v += __shfl_xor(s2v, 32, 64);
                if (lg == 0) { rk[(wave * 2 + 0) * 64 + 4 * li + j] = (double)s1v; rk[(wave * 2 + 1) * 64 + 4 * li + j] = (double)s2v; }
            }
            pend_tile = tile; pend_par = par;
        }
    }
    if (pend_tile >= 0) { __syncthreads(); stats_row(pend_tile, pend_par); }
    (void)nreq;
}

// ------------------------------------------------------------------------------------------------ 3x3 / stride 1 / pad 1: the input window staged once for all nine taps
// conv_bf16_kernel above re-stages its 128-pixel x tile for every K step — nine times per pixel for a 3x3 layer — and keeps the matrix pipe busy 20 % of the
// time on cifar100_deep (profiles/r03_pmc_cifar100_deep_bf16.json): it is bound by the bytes it moves into LDS (24 - 32 KB per K step of 16 - 32 us of MFMAs per
// SIMD), not by arithmetic.  Here a block of 512 threads (8 waves: one block per CU, the same occupancy as two of the blocks above) owns 256 consecutive output
// pixels x BN channels.  Per 64-channel chunk the FLAT input window of the tile — rows q = m0 - W - 1 .. m0 + 255 + W + 1 of the [n h w][Cin] tensor, zeros outside
// it (buffer loads beyond num_records) — is staged ONCE; tap (r, s) of pixel p reads window row p + r W + s.  Window rows which belong to a neighbouring image row
// / image (the tile ignores image borders) are never read: every lane keeps, per accumulator-tile row and tap, the LDS address of its operand — the window row, or
// an all-zero row when the tap falls outside the image — computed once per block (36 registers), so the inner loop has no address or border arithmetic at all.
// Only the filter tile is staged per K step (double buffered), the window of the next chunk lands in a second window buffer during the current chunk.
// LDS: 2 windows x (256 + 2 W + 2 rows rounded up to 64) x 128 B + a zero row + 2 x BN x 128 B = 122 KB at W = 32, BN = 128.
// Bytes into LDS per 128 pixels x 128 channels x 256 input channels: 0.38 MB (window 0.09 + filters 0.29) against 1.18 MB above.
// Same fragment layouts, swizzle (applied on the source side of the LDS-DMA), epilogue and statistics rows (one per 128 pixels: two per block) as above.
// Round 3 (what the fp32 window kernel taught, conv2d_mfma.hip): every instruction a wave issues is paid in matrix-pipe time, and a tap of this kernel is only
// 32 MFMAs x 8 clocks.  So: (1) the nine taps are unrolled — a tap's operand addresses are registers picked at compile time (the loop over taps selected them with a
// chain of eight v_cndmask per tile row) and both window buffers / all filter stages lie at compile-time distances (immediate offsets of the ds_reads; windows of up
// to 384 rows: W <= 32); (2) THREE filter stages: the filter tile of tap t + 2 is requested while tap t computes (one tap = 0.2 us is less than an L2 round trip);
// the next chunk's window is requested one 64-row piece per tap, so that every request has the same two taps to land; waits are counted (vmcnt(N) with N = the
// requests issued behind the one needed; the counter is in order); (3) LDS-DMA from inline assembly, so that the compiler does not wait for it in front of LDS reads.
// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate): the window kernel's counted waits, n <= 2 + 1 + 16
__device__ __forceinline__ void bh_wait(int n) {
    // (lgkmcnt(0) rides along: the wave's own LDS reads of the stage the next requests overwrite are complete before it reaches the barrier)
    switch (n) {
        case 1: __builtin_amdgcn_s_waitcnt(0x0071); break; case 2: __builtin_amdgcn_s_waitcnt(0x0072); break; case 3: __builtin_amdgcn_s_waitcnt(0x0073); break;
        case 9: __builtin_amdgcn_s_waitcnt(0x0079); break; case 10: __builtin_amdgcn_s_waitcnt(0x007A); break; case 11: __builtin_amdgcn_s_waitcnt(0x007B); break;
        case 17: __builtin_amdgcn_s_waitcnt(0x4071); break; case 18: __builtin_amdgcn_s_waitcnt(0x4072); break; case 19: __builtin_amdgcn_s_waitcnt(0x4073); break;
        default: __builtin_amdgcn_s_waitcnt(0x0070); break;
    }
}
#define BH_WROWS 384                                                // rows of a window buffer (256 + 2 W + 2 <= 384)
#define BH_WSTRIDE (BH_WROWS * 128 + 128)                           // window + its zero row
#define BH_BST (2 * BH_WSTRIDE)                                     // the three filter stages
// PERSIST (the 64-channel-wide instance: short K loops, where prologue and epilogue were half of a block's life — 454 -> 544 TF on 64 -> 64 @32x32; on the 128-wide
// instance the extra state costs registers and the longer loops gain nothing: 1057 -> 973 TF on 256 -> 256 @8x8, so it keeps one tile per block)
// MOM (round 4; data gradients only): the statistics rows carry, instead of the sums of the result d and d^2, the two moments which the normalisation backward of
// the layer BELOW needs from this gradient — sum d and sum d u, u = (yprev - mean) rstd, yprev that layer's activated convolution output (same pixels and
// channels as d) — so that layer's pass over dout and y for these sums (nc_sample_sums_kernel: 8 % of cifar100_deep's bf16 step) is not launched at all.
template <int WN, bool OUT_BF16, bool PERSIST = (WN == 1), bool MOM = false>
__global__ __launch_bounds__(512, 1) void conv_bf16_halo_kernel(Bf16Args a, const u16* __restrict__ x, const u16* __restrict__ wb, const float* __restrict__ bias,
                                                                 void* __restrict__ yv, double* __restrict__ stats, int win_rows,
                                                                 const u16* __restrict__ yprev = nullptr, const float* __restrict__ mom_mean = nullptr, const float* __restrict__ mom_rstd = nullptr,
                                                                 const u16* __restrict__ addend = nullptr) {
    constexpr int WM = 8 / WN, BMH = 256, TI = BMH / (16 * WM), BN = 64 * WN, PB = BN / 64;       // PB: 16-B pieces of the filter tile per thread and stage
    constexpr int B_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;
    // XCD-aware block order of the one-tile-per-block instance (round 4): the channel tiles of one pixel tile read the SAME input window, but as grid (tiles, channel
    // tiles) they were hundreds of blocks apart — the second one fetched the window again from beyond the L2 (PMC: 258 MB per launch on 256 -> 256 @16x16, N = 512,
    // against 134 MB of tensors).  Blocks are dispatched round-robin over the eight XCDs in linear order, so in a 1-D grid the ids L and L + 8 share an XCD and run
    // side by side: L = (tile / 8) 8 cts + ct 8 + tile % 8.
    int tile0 = blockIdx.x, ct0 = blockIdx.y;
    if (!PERSIST && a.xcd_tiles > 0) {
        const int cts = a.Cout / BN, L = (int)blockIdx.x, grp = L / (8 * cts), r = L - grp * 8 * cts;
        tile0 = grp * 8 + (r & 7); ct0 = r >> 3;
        if (tile0 >= a.xcd_tiles) return;                            // (the last group of eight may be partial)
    }
    const int n0 = ct0 * BN;
    const int W = a.Win, H = a.Hin;
    const int npieces = win_rows >> 6;                              // 64-row pieces of a window (5 or 6)
    // PERSISTENT over tiles (tile = blockIdx.x + k gridDim.x): at one block per CU nothing overlaps a block's prologue (first window + filter tiles: a memory round
    // trip) and epilogue (64 KB of stores draining) — 2.5 + 6.7 us of a 42 us block on 256 -> 256 @16x16, 2.1 + 3.0 of 9 us on 64 -> 64 @32x32 (per-block stamps,
    // tools/halo_stamps.py).  The chunk sequence runs on across tiles: the next tile's first window and filter tiles are requested during this tile's last chunk, its
    // stores drain under the next tile's taps.
    const int ntiles = PERSIST ? (int)((a.M + BMH - 1) / BMH) : 0, gstep = PERSIST ? (int)gridDim.x : 0;      // (one tile per block otherwise)
    int tile = tile0, m0 = tile * BMH;

    const unsigned long long xa = (unsigned long long)x, wa = (unsigned long long)wb;
    const bw_i32x4 rx = {(int)(unsigned)xa, (int)(unsigned)((xa >> 32) & 0xFFFFu), (int)a.x_bytes, 0x00020000};
    const bw_i32x4 rw = {(int)(unsigned)wa, (int)(unsigned)((wa >> 32) & 0xFFFFu), (int)a.w_bytes, 0x00020000};
    if (t < 16) *reinterpret_cast<uint4*>(lds + (t >> 3) * BH_WSTRIDE + BH_WROWS * 128 + (t & 7) * 16) = make_uint4(0u, 0u, 0u, 0u);      // the two zero rows

    // ---- staging roles (as above): piece p of this thread = LDS row 64 p + (t >> 3), physical slot t & 7, which holds logical slot (t & 7) ^ ((t >> 4) & 7)
    // WINDOW rows (round 4): physical slot = logical slot ^ (row & 6).  A tap reads 16 consecutive window rows starting at ANY row (pixel + r W + s), and a
    // ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): lanes of K groups lg and lg + 1 share a
    // group.  With the filter tile's swizzle ((row >> 1) & 7, conflict-free only for runs which start at a multiple of four rows) six of the nine taps paid two
    // LDS cycles per lane group (8 instead of 4 per read: the window reads were 2/3 of a tap's LDS time).  (row & 6) takes four values: same-parity rows 8 apart
    // share one, and those never sit in one lane group with the same K group; no two values differ by 1, which is what separates lg from lg + 1.
    const int lslot = (t & 7) ^ ((t >> 4) & 7);                     // filter tile rows (runs start at multiples of 16)
    const int lslot_w = (t & 7) ^ ((t >> 3) & 6);                   // window rows: row = 64 p + (t >> 3)
    const int cchunks = a.Cin / BF_BK;
    const unsigned lds0 = (unsigned)(size_t)lds + (unsigned)((t & ~63) * 16);
    // (offsets modulo 2^32: a pixel in front of the tensor wraps beyond num_records, one behind it exceeds it — zeros either way, no compare; x_bytes < 2^31)
    const unsigned wvstep = (unsigned)(64 * a.Cin * 2);
    auto wv_of = [&](int m0v) { return (unsigned)(m0v - W - 1 + (t >> 3)) * (unsigned)(a.Cin * 2) + (unsigned)(lslot_w * 16); };      // window row 0 = flat pixel m0 - W - 1
    auto stage_window_piece = [&](int wbuf, unsigned wv, int chunk, int p) {
        bw_dma16(rx, lds0 + (unsigned)(wbuf * BH_WSTRIDE + p * 8192), wv + (unsigned)p * wvstep, (unsigned)(chunk * 128));
    };
    int offb[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int rho = 64 * p + (t >> 3);                          // LDS row of the B image: 64 wn' + 16 j + c  <-  channel n0 + 64 wn' + 4 c + j
        const int co = n0 + (rho & ~63) + 4 * (rho & 15) + ((rho >> 4) & 3);
        offb[p] = (co * a.Cin) * 2 + lslot * 16;
    }
    auto stage_filter = [&](int stg, int chunk, int tap) {
        const unsigned ub = (unsigned)((tap * a.Cout) * a.Cin * 2 + chunk * 128);
#pragma unroll
        for (int p = 0; p < PB; ++p) bw_dma16(rw, lds0 + (unsigned)(BH_BST + stg * B_BYTES + p * 8192), (unsigned)offb[p], ub);
    };

    // ---- operand addresses (relative to a window buffer): tile row i of this wave = pixel pl = 16 TI wm + 16 i + li of the block; tap (r, s) -> window row
    // pl + r W + s, logical slot lg (K half 0; the other half is the address ^ 64), or the buffer's zero row when the tap leaves the image.  32-bit index decode with
    // multiply-high by host-made reciprocals (the 64-bit % and / this started with are ~150 instructions each: twelve of them were 5 us of every prologue).
    unsigned aaddr[TI][9];
    auto set_tile = [&](int m0v) __attribute__((always_inline)) {
        const unsigned um0 = (unsigned)m0v, q0w = __umulhi(um0, a.magic_w), wo0 = um0 - q0w * (unsigned)W, ho0 = q0w - __umulhi(q0w, a.magic_h) * (unsigned)H;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int pl = 16 * TI * wm + 16 * i + li;
            const bool ok = m0v + pl < (int)a.M;
            const unsigned u = wo0 + (unsigned)pl, dq = __umulhi(u, a.magic_w), wo = u - dq * (unsigned)W;
            const unsigned vv = ho0 + dq, ho = vv - __umulhi(vv, a.magic_h) * (unsigned)H;
            const bool rok[3] = {ok && ho > 0, ok, ok && ho + 1 < (unsigned)H}, cok[3] = {wo > 0, true, wo + 1 < (unsigned)W};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const int R = pl + r * W + sx;
                    aaddr[i][r * 3 + sx] = (rok[r] && cok[sx]) ? (unsigned)(R * 128 + ((lg ^ (R & 6)) << 4)) : (unsigned)(BH_WROWS * 128 + lg * 16);
                }
        }
    };
    set_tile(m0);
    const int swz = (li >> 1) & 7;
    const unsigned bad0 = (unsigned)(BH_BST + (64 * wn + li) * 128 + ((0 + lg) ^ swz) * 16), bad1 = (unsigned)(BH_BST + (64 * wn + li) * 128 + ((4 + lg) ^ swz) * 16);

    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- epilogue of a tile: acc[i][j][reg] -> pixel m0 + 16 TI wm + 16 i + 4 lg + reg, channel n0 + 64 wn + 4 li + j; `dead`: the window buffer nobody reads any more
    const int cbase = n0 + 64 * wn + 4 * li;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + cbase); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
    float mmu[4] = {0.f, 0.f, 0.f, 0.f}, mrs[4] = {0.f, 0.f, 0.f, 0.f};
    // (the MOM instance serves three uses, block-uniform: moments (yprev), a second gradient added to the result before it is stored — the residual link's,
    //  which autograd otherwise adds with a kernel of its own (`addend`) —, or both)
    if constexpr (MOM) {
        if (yprev != nullptr) {
            const float4 m4 = *reinterpret_cast<const float4*>(mom_mean + cbase), r4 = *reinterpret_cast<const float4*>(mom_rstd + cbase);
            mmu[0] = m4.x; mmu[1] = m4.y; mmu[2] = m4.z; mmu[3] = m4.w; mrs[0] = r4.x; mrs[1] = r4.y; mrs[2] = r4.z; mrs[3] = r4.w;
        }
    }
    // ACT: the activation as a compile-time constant (-1: the descriptor's, decided per element), FULL: all 256 pixels inside the tensor (round 4: the per-element
    // five-way switch and bounds test around each of a lane's 16 TI results were most of a tile's epilogue — 6.7 us of a 42 us block on 256 -> 256 @16x16)
    auto finish_tile_as = [&](int dead, auto actc, auto fullc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
        constexpr bool FULL = decltype(fullc)::value;
        const int act = ACT >= 0 ? ACT : a.act;
        float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
        uint2 yq[MOM ? TI : 1][4], aq[MOM ? TI : 1][4];             // MOM: the tile's yprev / addend values, all requested before the first use
        if constexpr (MOM) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int m = m0 + 16 * TI * wm + 16 * i + 4 * lg + reg;
                    const size_t at = (size_t)((FULL || m < (int)a.M) ? m : 0) * a.Cout + cbase;
                    yq[i][reg] = yprev != nullptr ? *reinterpret_cast<const uint2*>(yprev + at) : make_uint2(0u, 0u);
                    aq[i][reg] = addend != nullptr ? *reinterpret_cast<const uint2*>(addend + at) : make_uint2(0u, 0u);
                }
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = m0 + 16 * TI * wm + 16 * i + 4 * lg + reg;
                if (FULL || m < (int)a.M) {
                    float o[4];
                    float uu[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (MOM) {
                        const uint2 q = yq[i][reg];
                        uu[0] = (__uint_as_float(q.x << 16) - mmu[0]) * mrs[0]; uu[1] = (__uint_as_float(q.x & 0xffff0000u) - mmu[1]) * mrs[1];
                        uu[2] = (__uint_as_float(q.y << 16) - mmu[2]) * mrs[2]; uu[3] = (__uint_as_float(q.y & 0xffff0000u) - mmu[3]) * mrs[3];
                    }
                    float ad[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (MOM) {
                        const uint2 q = aq[i][reg];
                        ad[0] = __uint_as_float(q.x << 16); ad[1] = __uint_as_float(q.x & 0xffff0000u); ad[2] = __uint_as_float(q.y << 16); ad[3] = __uint_as_float(q.y & 0xffff0000u);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = dcv_act_apply(acc[i][j][reg] + bv[j], act, a.slope) + ad[j];
                        ssum[j] += o[j]; ssq[j] += MOM ? o[j] * uu[j] : o[j] * o[j];
                    }
                    if constexpr (OUT_BF16) {
                        const bf16x4 h = (bf16x4){(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<u16*>(yv) + (size_t)m * a.Cout + cbase) = h;
                    } else {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + (size_t)m * a.Cout + cbase) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (stats != nullptr) {
            // statistics rows are per 128 pixels (the layout dcv_conv2d_bf16_fwd_stats_layout promises): the block's two halves = waves wm < WM / 2 and the others
            __syncthreads();                                       // every wave is done reading the dead window
            double* red = reinterpret_cast<double*>(lds + dead * BH_WSTRIDE);      // [8 waves][2][64]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s1v = ssum[j], s2v = ssq[j];
                s1v += __shfl_xor(s1v, 16, 64); s2v += __shfl_xor(s2v, 16, 64);
                s1v += __shfl_xor(s1v, 32, 64); s2v += __shfl_xor(s2v, 32, 64);
                if (lg == 0) { red[(wave * 2 + 0) * 64 + 4 * li + j] = (double)s1v; red[(wave * 2 + 1) * 64 + 4 * li + j] = (double)s2v; }
            }
            __syncthreads();
            const int rows_here = (int)((a.M + 127) / 128), rows_total = a.stat_rows ? a.stat_rows : rows_here;
            for (int e = t; e < 2 * 2 * BN; e += 512) {
                const int half = e / (2 * BN), r2 = e - half * 2 * BN, q = r2 / BN, c = r2 - q * BN;
                const int wnn = c >> 6, cc = c & 63;
                double v = 0.0;
#pragma unroll
                for (int w = 0; w < WM / 2; ++w) v += red[(((half * (WM / 2) + w) * WN + wnn) * 2 + q) * 64 + cc];
                const int row = 2 * tile + half;
                if (row < rows_here) stats[((size_t)q * a.Cout + n0 + c) * rows_total + a.stat_row0 + row] = v;            // [2][Cout][rows]
            }
        }
    };
    auto finish_tile = [&](int dead) __attribute__((always_inline)) {
        const bool full = m0 + BMH <= (int)a.M;
        if (full && a.act == DCV_ACT_NONE) finish_tile_as(dead, std::integral_constant<int, DCV_ACT_NONE>{}, std::true_type{});
        else if (full && a.act == DCV_ACT_RELU) finish_tile_as(dead, std::integral_constant<int, DCV_ACT_RELU>{}, std::true_type{});
        else finish_tile_as(dead, std::integral_constant<int, -1>{}, std::false_type{});
    };

    unsigned wv_cur = wv_of(m0), wv_nxt = (PERSIST && tile + gstep < ntiles) ? wv_of(m0 + gstep * BMH) : 0u;
    for (int p = 0; p < npieces; ++p) stage_window_piece(0, wv_cur, 0, p);
    stage_filter(0, 0, 0);
    stage_filter(1, 0, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0)
    const int total = PERSIST ? ((ntiles - tile + gstep - 1) / gstep) * cchunks : cchunks;      // chunks of all this block's tiles
    int chunk = 0;
    bool fresh = false;                                            // the 4 TI stores of the previous tile's epilogue are younger than the requests the first two taps wait for
    // one chunk out of window buffer PAR; tap t out of filter stage t % 3 (nine taps per chunk: the stages stay in step).
    // Round 4: ROLLING OPERAND PIPELINE.  A block's eight waves pass the tap's barrier together, so in the round-3 form (read a K half's fragments, wait, sixteen
    // MFMAs, twice per tap) both waves of a SIMD sat in their LDS latencies at the same time and the matrix pipe was busy 0.29 of the cycles (a tap took ~2150 clocks
    // for 1024 of MFMAs).  Now the operands live in two fragment sets: set 0 serves K half 0 of a tap, set 1 K half 1, and the reads of a set are issued one
    // sixteen-MFMA block AHEAD of the block which consumes it — set 0 of tap t right behind the tap's barrier (its filter stage has landed), in front of the MFMAs of
    // tap t - 1's second half; set 1 in front of tap t's first half.  Every fragment read has a whole MFMA block (256 clocks of this wave's, 512 of the SIMD's) to land.
    // The tap's barrier therefore sits between the two halves of the previous tap's arithmetic; the vector-memory request sequence is unchanged.
    bf16x8 fa0[TI], fb0[4], fa1[TI], fb1[4];
    bool pend = false;                                             // set 1 holds the operands of sixteen MFMAs not issued yet
    unsigned k64 = 64u;
    auto mma_set = [&](const bf16x8 (&fa)[TI], const bf16x8 (&fb)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    auto chunk_body = [&](auto par, int gc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        const bool more = gc + 1 < total, last = chunk + 1 == cchunks;
        const unsigned nwv = last ? wv_nxt : wv_cur;               // the next chunk's window: this tile's next 64 channels, or the next tile's first
        const int nchunk = last ? 0 : chunk + 1;
        asm volatile("" : "+s"(k64));                              // (loop-variant to the optimiser: the 36 addresses ^ 64 are not worth 36 more registers)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // the filter tile of this tap was requested two taps ago; behind it (vmcnt counts in order, stores included): the request(s) of the previous tap — PB
            // filter pieces + one window piece while the next window is being fetched (taps 0 .. npieces - 1 >= 4) — and, in a tile's first two taps, the stores
            // of the previous tile.  lgkmcnt(0): this wave's reads of set 1 (the previous tap's stage, which the requests below overwrite) are complete.
            if (tap == 8 && !more) __builtin_amdgcn_s_waitcnt(0x0070);
            else bh_wait(PB + ((tap >= 1 && tap <= 5 && more) ? 1 : 0) + ((PERSIST && tap < 2 && fresh) ? 4 * TI : 0));
            __syncthreads();                                       // everybody's pieces have landed; every wave is done reading filter stage (tap + 2) % 3
            if (more && tap < npieces) stage_window_piece(PAR ^ 1, nwv, nchunk, tap);               // (the other window buffer: last read in the previous chunk)
            if (tap + 2 < 9) stage_filter((tap + 2) % 3, chunk, tap + 2);
            else if (more) stage_filter((tap + 2) % 3, nchunk, tap - 7);
            // K half 0 of this tap -> set 0, in front of the previous tap's second half
#pragma unroll
            for (int j = 0; j < 4; ++j) fb0[j] = *reinterpret_cast<const bf16x8*>(lds + (tap % 3) * B_BYTES + j * 2048 + bad0);
#pragma unroll
            for (int i = 0; i < TI; ++i) fa0[i] = *reinterpret_cast<const bf16x8*>(lds + PAR * BH_WSTRIDE + aaddr[i][tap]);
            __builtin_amdgcn_sched_barrier(0);
            if (tap > 0 || pend) mma_set(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            // K half 1 -> set 1, in front of this tap's first half
#pragma unroll
            for (int j = 0; j < 4; ++j) fb1[j] = *reinterpret_cast<const bf16x8*>(lds + (tap % 3) * B_BYTES + j * 2048 + bad1);
#pragma unroll
            for (int i = 0; i < TI; ++i) fa1[i] = *reinterpret_cast<const bf16x8*>(lds + PAR * BH_WSTRIDE + (aaddr[i][tap] ^ k64));
            __builtin_amdgcn_sched_barrier(0);
            mma_set(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
        }
        pend = true;
        if constexpr (PERSIST) {
            fresh = false;
            if (last) {
                mma_set(fa1, fb1); pend = false;                   // (a tile's last sixteen MFMAs: nothing to hide them behind)
                finish_tile(PAR);
                chunk = 0; tile += gstep; m0 += gstep * BMH;
                if (more) { set_tile(m0); wv_cur = wv_nxt; wv_nxt = tile + gstep < ntiles ? wv_of(m0 + gstep * BMH) : 0u; fresh = true; }
            } else ++chunk;
        } else ++chunk;
    };
    int gc = 0;
    for (; gc + 1 < total; gc += 2) { chunk_body(std::integral_constant<int, 0>{}, gc); chunk_body(std::integral_constant<int, 1>{}, gc + 1); }
    if (gc < total) chunk_body(std::integral_constant<int, 0>{}, gc);
    if (pend) mma_set(fa1, fb1);
    if constexpr (!PERSIST) finish_tile((cchunks - 1) & 1);         // (one tile: its epilogue once, behind the loop)
}

// ------------------------------------------------------------------------------------------------ round 4: 64 input channels — the filters in REGISTERS
// With Cin = 64 a tile of the window kernel above is ONE chunk: nine taps of sixteen MFMAs per wave (its 64-channel-wide instance), each behind a barrier and a
// filter-tile request — 0.17 - 0.23 of the bf16 peak on 64 -> 64 at 32 x 32 and 56 x 56 (cifar100_deep's largest layer, four layers of resnet18_style: the slowest
// convolutions of both steps).  But the whole filter of 64 output channels is 9 x 64 x 64 values = 288 registers per lane as MFMA B fragments: a block of FOUR
// waves (one per SIMD: 512 registers each), wave tile 64 pixels x 64 channels, loads them ONCE and then walks its tiles with nothing but the window in LDS
// (two buffers: the next tile's lands under this tile's taps), four fragment reads per sixteen MFMAs, no barrier inside a tile and no filter traffic at all.
// Same window layout, border handling (zero row), epilogue (activation, bf16 store, statistics rows per 128 pixels / the MOM moments and addend) as above.
// The filter fragments live in the ACCUMULATION registers and the MFMA reads them there (gfx950 takes AGPRs as A / B sources): left to the compiler, B operands are
// vector registers and the 288 - 256 that do not fit were copied back (v_accvgpr_read) in front of every group of MFMAs — 16 copies + the operand addresses in front
// of 16 MFMAs on a block with ONE wave per SIMD: a tap took 1200 clocks for 512 of MFMAs.  Inline assembly is opaque to the hazard recogniser: the accumulators are
// read by vector instructions only behind rf_mfma_drain(acc).
__device__ __forceinline__ void rf_mfma_a(f32x4& c, const bf16x8& av, const bf16x8& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(b)); }
__device__ __forceinline__ void rf_mfma_v(f32x4& c, const bf16x8& av, const bf16x8& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(b)); }
// (the accumulators are operands of the drain: every later read of them depends on it, so the scheduler cannot place one between the last MFMA and the nops)
__device__ __forceinline__ void rf_mfma_drain(f32x4 (&c)[4][4]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7"
                 : "+v"(c[0][0]), "+v"(c[0][1]), "+v"(c[0][2]), "+v"(c[0][3]), "+v"(c[1][0]), "+v"(c[1][1]), "+v"(c[1][2]), "+v"(c[1][3]),
                   "+v"(c[2][0]), "+v"(c[2][1]), "+v"(c[2][2]), "+v"(c[2][3]), "+v"(c[3][0]), "+v"(c[3][1]), "+v"(c[3][2]), "+v"(c[3][3]) :: "memory");
}
#define RF_ATAPS 8                  // taps whose fragments are held in AGPRs (8 x 32 = 256); the ninth in vector registers

template <bool MOM>
__global__ __launch_bounds__(256, 1) void conv_bf16_halo_rf_kernel(Bf16Args a, const u16* __restrict__ x, const u16* __restrict__ wb, const float* __restrict__ bias,
                                                                    u16* __restrict__ yv, double* __restrict__ stats, int win_rows,
                                                                    const u16* __restrict__ yprev, const float* __restrict__ mom_mean, const float* __restrict__ mom_rstd,
                                                                    const u16* __restrict__ addend) {
    constexpr int BMH = 256, TI = 4;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wm = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.y * 64;
    const int W = a.Win, H = a.Hin;
    const int nhalf = win_rows >> 5;                                // 32-row pieces of a window (a piece = one LDS-DMA instruction of the block's four waves)
    const int ntiles = (int)((a.M + BMH - 1) / BMH), gstep = (int)gridDim.x;
    int tile = blockIdx.x, m0 = tile * BMH;
    const unsigned long long xa = (unsigned long long)x;
    const bw_i32x4 rx = {(int)(unsigned)xa, (int)(unsigned)((xa >> 32) & 0xFFFFu), (int)a.x_bytes, 0x00020000};
    if (t < 16) *reinterpret_cast<uint4*>(lds + (t >> 3) * BH_WSTRIDE + BH_WROWS * 128 + (t & 7) * 16) = make_uint4(0u, 0u, 0u, 0u);      // the two zero rows

    // ---- the filter: B fragment (tap, K half ks, channel tile j) of lane (li, lg) = channel n0 + 4 li + j, input channels 32 ks + 8 lg .. + 7 of [tap][Cout][64]
    // Taps 0 .. 7: 256 AGPRs.  The ninth tap's eight fragments stay in LDS ([ks][j][lane], 8 KB behind the window buffers; the same for every wave) and are read
    // while the eighth tap computes: as 32 more vector registers they pushed the MOM instance (moments / addend epilogue) into scratch — 260 us against 149 for
    // the plain one on 64 -> 64 @56x56.
    bf16x8 fb[RF_ATAPS][2][4];
#pragma unroll
    for (int tap = 0; tap < RF_ATAPS; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[tap][ks][j] = *reinterpret_cast<const bf16x8*>(wb + ((size_t)(tap * a.Cout + n0 + 4 * li + j) * 64 + 32 * ks + 8 * lg));
    bf16x8* const fb8 = reinterpret_cast<bf16x8*>(lds + 2 * BH_WSTRIDE);
    if (wm == 0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb8[(ks * 4 + j) * 64 + lane] = *reinterpret_cast<const bf16x8*>(wb + ((size_t)(8 * a.Cout + n0 + 4 * li + j) * 64 + 32 * ks + 8 * lg));
    }

    // ---- window staging: piece hp = rows 32 hp + (t >> 3), physical slot t & 7 holds logical slot (t & 7) ^ (row & 6)
    const int lslot_w = (t & 7) ^ ((t >> 3) & 6);
    const unsigned lds0 = (unsigned)(size_t)lds + (unsigned)((t & ~63) * 16);
    const unsigned wvstep = (unsigned)(32 * 64 * 2);
    auto wv_of = [&](int m0v) { return (unsigned)(m0v - W - 1 + (t >> 3)) * 128u + (unsigned)(lslot_w * 16); };
    auto stage_piece = [&](int wbuf, unsigned wv, int hp) __attribute__((always_inline)) {
        bw_dma16(rx, lds0 + (unsigned)(wbuf * BH_WSTRIDE + hp * 4096), wv + (unsigned)hp * wvstep, 0u);
    };
    auto stage_window = [&](int wbuf, unsigned wv) {
        for (int hp = 0; hp < nhalf; ++hp) stage_piece(wbuf, wv, hp);
    };
    // ---- operand addresses (relative to a window buffer; the other K half is the address ^ 64).  Tile row i = pixel pl = 64 wm + 16 i + li of the block reads, for
    // tap (r, s), window row R = pl + r W + s: R changes by 16 per i, the swizzle term (R & 6) does not — ONE address per tap and lane (rc[tap], row i at + 2048 i,
    // an immediate of the read) is the same for every tile.  What a tile changes is which taps leave the image: 36 bits per lane (vmask: bit 9 i + tap), decoded
    // once per tile; a tap outside reads the buffer's zero row (zc[i] = its address - 2048 i).  (The first version kept the 36 selected addresses: with the 288
    // filter registers that spilled, and the reloads waited for the window request in front of them — 7 us of every 18 us tile.)
    unsigned rc[9], zc[TI];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) { const int R = 64 * wm + li + r * W + sx; rc[r * 3 + sx] = (unsigned)(R * 128 + ((lg ^ (R & 6)) << 4)); }
#pragma unroll
    for (int i = 0; i < TI; ++i) zc[i] = (unsigned)(BH_WROWS * 128 + lg * 16 - 2048 * i);
    // (opaque values: left visible, the select between two array elements below becomes an indexed load from a stack copy of the arrays)
#pragma unroll
    for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(rc[k]));
#pragma unroll
    for (int i = 0; i < TI; ++i) asm volatile("" : "+v"(zc[i]));
    unsigned aaddr[TI][9];                                          // this tile's: rc[tap] or zc[i]
    auto set_tile = [&](int m0v) __attribute__((always_inline)) {
        const unsigned um0 = (unsigned)m0v, q0w = __umulhi(um0, a.magic_w), wo0 = um0 - q0w * (unsigned)W, ho0 = q0w - __umulhi(q0w, a.magic_h) * (unsigned)H;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int pl = 64 * wm + 16 * i + li;
            const unsigned u = wo0 + (unsigned)pl, dq = __umulhi(u, a.magic_w), wo = u - dq * (unsigned)W;
            const unsigned vv = ho0 + dq, ho = vv - __umulhi(vv, a.magic_h) * (unsigned)H;
            // (pixels beyond the tensor's end compute garbage nobody stores)
            const bool rok[3] = {ho > 0, true, ho + 1 < (unsigned)H}, cok[3] = {wo > 0, true, wo + 1 < (unsigned)W};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                unsigned rv_ = rc[tap], zv_ = zc[i];
                asm("" : "+v"(rv_), "+v"(zv_));                      // (a select of two loads is folded into a load from a selected address, and the arrays stay in memory)
                aaddr[i][tap] = (rok[tap / 3] && cok[tap % 3]) ? rv_ : zv_;
            }
        }
    };
    set_tile(m0);
    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int cbase = n0 + 4 * li;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) { const float4 b4 = *reinterpret_cast<const float4*>(bias + cbase); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
    float mmu[4] = {0.f, 0.f, 0.f, 0.f}, mrs[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MOM) {
        if (yprev != nullptr) {
            const float4 m4 = *reinterpret_cast<const float4*>(mom_mean + cbase), r4 = *reinterpret_cast<const float4*>(mom_rstd + cbase);
            mmu[0] = m4.x; mmu[1] = m4.y; mmu[2] = m4.z; mmu[3] = m4.w; mrs[0] = r4.x; mrs[1] = r4.y; mrs[2] = r4.z; mrs[3] = r4.w;
        }
    }
    // the epilogue of a tile: acc[i][j][reg] -> pixel m0 + 64 wm + 16 i + 4 lg + reg, channel cbase + j; `dead`: the window buffer the tile has finished with.
    // ACT: the activation as a compile-time constant (-1: the descriptor's, decided per element), FULL: all 256 pixels inside the tensor — the per-element form
    // (a five-way switch and a bounds test around each of a lane's 64 results, on a block which has no second wave per SIMD to hide them behind) was 13 of the
    // first version's 18 us per tile.
    auto finish_tile = [&](int dead, auto actc, auto fullc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
        constexpr bool FULL = decltype(fullc)::value;
        const int act = ACT >= 0 ? ACT : a.act;
        float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            uint2 yq[MOM ? 4 : 1], aq[MOM ? 4 : 1];
            if constexpr (MOM) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int m = m0 + 64 * wm + 16 * i + 4 * lg + reg;
                    const size_t at = (size_t)((FULL || m < (int)a.M) ? m : 0) * a.Cout + cbase;
                    yq[reg] = yprev != nullptr ? *reinterpret_cast<const uint2*>(yprev + at) : make_uint2(0u, 0u);
                    aq[reg] = addend != nullptr ? *reinterpret_cast<const uint2*>(addend + at) : make_uint2(0u, 0u);
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = m0 + 64 * wm + 16 * i + 4 * lg + reg;
                if (FULL || m < (int)a.M) {
                    float o[4], uu[4] = {0.f, 0.f, 0.f, 0.f}, ad[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (MOM) {
                        const uint2 q = yq[reg], q2 = aq[reg];
                        uu[0] = (__uint_as_float(q.x << 16) - mmu[0]) * mrs[0]; uu[1] = (__uint_as_float(q.x & 0xffff0000u) - mmu[1]) * mrs[1];
                        uu[2] = (__uint_as_float(q.y << 16) - mmu[2]) * mrs[2]; uu[3] = (__uint_as_float(q.y & 0xffff0000u) - mmu[3]) * mrs[3];
                        ad[0] = __uint_as_float(q2.x << 16); ad[1] = __uint_as_float(q2.x & 0xffff0000u); ad[2] = __uint_as_float(q2.y << 16); ad[3] = __uint_as_float(q2.y & 0xffff0000u);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = dcv_act_apply(acc[i][j][reg] + bv[j], act, a.slope) + ad[j];
                        ssum[j] += o[j]; ssq[j] += MOM ? o[j] * uu[j] : o[j] * o[j];
                    }
                    *reinterpret_cast<bf16x4*>(yv + (size_t)m * a.Cout + cbase) = (bf16x4){(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (stats != nullptr) {
            // statistics rows per 128 pixels: the tile's halves = waves 0, 1 and waves 2, 3 (same butterfly and wave order as the window kernel's 64-wide instance)
            __syncthreads();                                       // every wave is done reading the dead window
            double* red = reinterpret_cast<double*>(lds + dead * BH_WSTRIDE);      // [4 waves][2][64]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s1v = ssum[j], s2v = ssq[j];
                s1v += __shfl_xor(s1v, 16, 64); s2v += __shfl_xor(s2v, 16, 64);
                s1v += __shfl_xor(s1v, 32, 64); s2v += __shfl_xor(s2v, 32, 64);
                if (lg == 0) { red[(wm * 2 + 0) * 64 + 4 * li + j] = (double)s1v; red[(wm * 2 + 1) * 64 + 4 * li + j] = (double)s2v; }
            }
            __syncthreads();
            const int rows_here = (int)((a.M + 127) / 128), rows_total = a.stat_rows ? a.stat_rows : rows_here;
            {
                const int half = t >> 7, q = (t >> 6) & 1, c = t & 63;      // 256 threads = 2 halves x 2 sums x 64 channels
                const double v = red[((half * 2 + 0) * 2 + q) * 64 + c] + red[((half * 2 + 1) * 2 + q) * 64 + c];
                const int row = 2 * tile + half;
                if (row < rows_here) stats[((size_t)q * a.Cout + n0 + c) * rows_total + a.stat_row0 + row] = v;
            }
        }
    };

    stage_window(0, wv_of(m0));
    unsigned k64 = 64u;
    bf16x8 fa0[TI], fa1[TI];
    int par = 0;
    bool first = true;
    for (; tile < ntiles; tile += gstep, m0 += gstep * BMH, par ^= 1) {
        // this tile's window was requested a tile ago; younger than it: the previous tile's 16 stores
        // (exactly the younger operations — 16 stores, and the statistics store when there is one: one too few waits for the first store's trip to memory)
        if (first) __builtin_amdgcn_s_waitcnt(0x0F70); else if (stats != nullptr) __builtin_amdgcn_s_waitcnt(DCV_VMCNT(17)); else __builtin_amdgcn_s_waitcnt(DCV_VMCNT(16));
        __syncthreads();                                           // everybody's pieces have landed; every wave is done with the other buffer (its taps, its statistics scratch)
        // the next tile's window: its 32-row pieces are requested one or two per tap (an LDS-DMA instruction costs its wave 60 - 185 issue clocks: ten to twelve in a
        // row were half a microsecond in front of every tile's first MFMA)
        const bool more = tile + gstep < ntiles;
        const unsigned nwv = more ? wv_of(m0 + gstep * BMH) : 0u;
        // MOM: the epilogue reads this tile's rows of yprev / addend (128 bytes per pixel and tensor) four times a tile with nothing to hide a trip to memory
        // behind (one wave per SIMD; 32 + 32 more registers to prefetch them do not exist): one dword of each row is requested NOW, by the thread of that pixel,
        // into a register nobody reads — the line waits in L2 when the epilogue asks for it.  (The register stays reserved until the epilogue's loads, younger and
        // returned in order, have been waited for.)
        unsigned pf0 = 0u, pf1 = 0u;
        if constexpr (MOM) {
            const int mp = m0 + t < (int)a.M ? m0 + t : m0;
            if (yprev != nullptr) { const u16* q = yprev + (size_t)mp * a.Cout + n0; asm volatile("global_load_dword %0, %1, off" : "=v"(pf0) : "v"(q)); }
            if (addend != nullptr) { const u16* q = addend + (size_t)mp * a.Cout + n0; asm volatile("global_load_dword %0, %1, off" : "=v"(pf1) : "v"(q)); }
        }
        if (!first) set_tile(m0);
        first = false;
        asm volatile("" : "+s"(k64));
        const char* wbase = lds + par * BH_WSTRIDE;
#pragma unroll
        for (int i = 0; i < TI; ++i) fa0[i] = *reinterpret_cast<const bf16x8*>(wbase + aaddr[i][0] + 2048 * i);
        bf16x8 f8[2][4];                                            // the ninth tap's fragments: requested in front of the eighth tap's second half
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int i = 0; i < TI; ++i) fa1[i] = *reinterpret_cast<const bf16x8*>(wbase + (aaddr[i][tap] ^ k64) + 2048 * i);
            if (more) { if (tap < nhalf) stage_piece(par ^ 1, nwv, tap); if (tap + 9 < nhalf) stage_piece(par ^ 1, nwv, tap + 9); }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { if (tap < RF_ATAPS) rf_mfma_a(acc[i][j], fa0[i], fb[tap < RF_ATAPS ? tap : 0][0][j]); else rf_mfma_v(acc[i][j], fa0[i], f8[0][j]); }
            if (tap + 1 < 9) {
#pragma unroll
                for (int i = 0; i < TI; ++i) fa0[i] = *reinterpret_cast<const bf16x8*>(wbase + aaddr[i][tap + 1] + 2048 * i);
            }
            if (tap == RF_ATAPS - 1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 4; ++j) f8[ks][j] = fb8[(ks * 4 + j) * 64 + lane];
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { if (tap < RF_ATAPS) rf_mfma_a(acc[i][j], fa1[i], fb[tap < RF_ATAPS ? tap : 0][1][j]); else rf_mfma_v(acc[i][j], fa1[i], f8[1][j]); }
        }
        rf_mfma_drain(acc);
        if (m0 + BMH <= (int)a.M && a.act == DCV_ACT_NONE) finish_tile(par, std::integral_constant<int, DCV_ACT_NONE>{}, std::true_type{});
        else if (m0 + BMH <= (int)a.M && a.act == DCV_ACT_RELU) finish_tile(par, std::integral_constant<int, DCV_ACT_RELU>{}, std::true_type{});
        else finish_tile(par, std::integral_constant<int, -1>{}, std::false_type{});
        if constexpr (MOM) asm volatile("" :: "v"(pf0), "v"(pf1));
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient, all nine taps per block, bf16 operands
// dW[tap][co][ci] = sum over pixels of dz[p][co] * x[p (+) tap][ci] for 3x3 / stride 1 / pad 1 layers: GEMM with M = co, N = ci, K = pixels.  Both operands
// are K-strided in memory ([pixel][channel] rows), so the LDS images keep the natural layout and the fragments are read TRANSPOSED (`ds_read_b64_tr_b16`:
// a 16-lane group fetches a 4-pixel x 16-channel block and every lane receives the four pixels of its channel).  Structure as conv2d_mfma.hip's all-taps
// kernel: a block owns (pixel chunk, 64 ci, 64 co); per step of 32 pixels of one image it stages the dz tile [32][64] and the flat x window
// [32 + 2W + 2][64] (zero outside the image) once for all nine taps; tap (r, s) of pixel p reads window row p + r W + s; window rows which wrapped around the
// left / right edge are cancelled by zeroing the dz operand of border pixels for s = 0 / s = 2 (two mask words per pixel pair, applied with v_and).
//   * wave w owns input-channel tile w (16 ci) and ALL four output-channel tiles: per step 4 x 2 transposed reads of dz + 9 x 2 of x feed 36 MFMAs
//     (v_mfma_f32_16x16x32_bf16), a third of the LDS traffic of one-co-tile-per-wave; 36 accumulator tiles = 144 VGPRs, two waves per SIMD.
//   * K order inside a step: lane group g takes pixels 4g..4g+3 and 16+4g..16+4g+3, the same for both operands (any common permutation of K is valid), so a
//     32-lane half always reads 8 CONSECUTIVE rows; with 160-B rows (128 B + 32 B pad: 40 r mod 64 visits every multiple of 8) those reads are conflict-free
//     at every tap offset.
//   * partial[chunk][tap][co][ci] fp32 slabs + conv2d_mfma.hip's ordered reduce (deterministic).
struct W16Args {
    int N, H, W, Cin, Cout;
    int XR;                         // window rows: 32 + 2 W + 2
    int HWp;                        // pixels per image rounded up to a multiple of 32
    long M, m_per_chunk;            // N * HWp; padded pixels per split-K chunk (multiple of 32)
    size_t wtotal;
    int stages;                     // LDS stages of a wave group: 3 (tiles requested two steps ahead) where two blocks per CU can afford them, else 2
    unsigned magic_w;               // n / W = mulhi(n, magic_w)
    int ips;                        // images per step: 2 for maps of at most 16 pixels (round 4, below), else 1
};
#define W16_PS 32
#define W16_LD 160                  // LDS row stride in bytes
#define W16_XP 8                    // 16-B pieces of the x window per thread and step (XR * 8 <= 256 * W16_XP)

// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate): n = instructions per wave and step, 2 .. 12
__device__ __forceinline__ void w16_wait(int n) {
    switch (n) {
        case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break; case 3: __builtin_amdgcn_s_waitcnt(0x0F73); break; case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
        case 5: __builtin_amdgcn_s_waitcnt(0x0F75); break; case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break; case 7: __builtin_amdgcn_s_waitcnt(0x0F77); break;
        case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break; case 9: __builtin_amdgcn_s_waitcnt(0x0F79); break; case 10: __builtin_amdgcn_s_waitcnt(0x0F7A); break;
        case 11: __builtin_amdgcn_s_waitcnt(0x0F7B); break; case 12: __builtin_amdgcn_s_waitcnt(0x0F7C); break;
        default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ s16x4 tr_read(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p); }
__device__ __forceinline__ s16x4 tr_read_lds(unsigned lds_byte_address) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(unsigned long long)lds_byte_address); }

// GROUPS: wave groups of 256 threads splitting the block's pixel range; their accumulators are summed through LDS at the end (group 1's 36 tiles = 144 KB,
// the staging buffers are dead by then), which halves the number of fp32 partial slabs the ordered reduce has to read for the same number of resident waves.
// TAPS: 9 (3x3 / pad 1) or 1 (1x1 / pad 0: a plain GEMM dW[co][ci] = sum over pixels — the weight gradient of a first layer rewritten as im2col + 1x1,
// dcv_im2col_bf16): one tap, the x "window" is the step's 32 pixels themselves, no border masks.
// Round 3: the tiles are staged by LDS-DMA (no VGPR round trip, no address selects: a step is only 36 MFMAs x 8 clocks per wave, and every other instruction is
// paid in matrix-pipe time), TWO steps ahead where three stages fit the LDS budget of two blocks per CU (a step is shorter than a memory round trip).  A stage is one
// array of 160-B rows: [36 rows of dz (32 + 4 that belong to the next step)][a gap row for the overhang][6 nix + 1 rows of the x window]; a wave instruction writes 64 consecutive 16-B pieces =
// 6 rows of 10 pieces (8 of data, 2 of pad fetched out of range) + the first 4 pieces of a seventh row (which the next instruction writes again); the buffer
// resources are the step's IMAGE, so window rows outside it and the padding pixels behind it read zeros.  Every wave issues the same number K of instructions per step
// (the ones beyond the stage's 6 + nix go, all lanes out of range, to a scratch KB), so that `s_waitcnt vmcnt(K)` means "everything but the youngest step".
// Round 4, maps of at most 16 pixels (the 4 x 4 layers of a CIFAR-sized spec at its deepest stage): a step of 32 pixels of ONE image was half padding — half of the
// MFMAs of three of cifar100_deep's ten layers multiplied zeros.  There a step takes TWO images (`ips` = 2): pixels 0 .. 15 of the K axis are image 2 l_n, pixels
// 16 .. 31 image 2 l_n + 1 (a fragment's second K half), each with its own buffer resources, dz rows and x window; the stage is [19 rows of dz of the first image
// (16 + the instruction granularity and its overhang)][19 of the second][window of the first: 6 nix + 1 rows][window of the second], so the second K half of
// a fragment sits `dzh` / `xh` rows behind the first instead of 16.
template <int GROUPS, int TAPS = 9>
__global__ __launch_bounds__(256 * GROUPS, 2) void conv_bf16_wgrad9_kernel(W16Args a, const u16* __restrict__ x, const u16* __restrict__ dy, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    const int XR = a.XR, nix = (XR + 5) / 6, nst = a.stages, ips = a.ips;
    const int dz_rows = ips == 2 ? 38 : 37, dzh = ips == 2 ? 19 : 16, xh = ips == 2 ? 6 * nix + 1 : 16;
    const int STAGE_BYTES = (dz_rows + ips * (6 * nix + 1)) * W16_LD;
    const int GROUP_BYTES = nst * STAGE_BYTES + nst * 16 * 2 * (int)sizeof(unsigned) + 1024;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), t = threadIdx.x & 255, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    char* St = smem16 + grp * GROUP_BYTES;                       // [nst] stages: dz tile [36 rows][co], x window [6 nix + 1 rows][ci]
    unsigned* Cm = reinterpret_cast<unsigned*>(St + nst * STAGE_BYTES);   // [nst][16 pixel pairs][2]: and-masks of the pair's two bf16 for s = 0 (w > 0) and s = 2 (w < W - 1)
    const unsigned scratch = (unsigned)(size_t)St + (unsigned)(nst * STAGE_BYTES + nst * 128);
    const int li = lane & 15, lg = lane >> 4;
    const int ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
    // pixel counts fit 32 bits (tensors are below 2 GiB); all of this is wave-uniform (scalar registers)
    const int per_group = (int)(a.m_per_chunk / GROUPS);         // multiple of 32
    const int mb = (int)blockIdx.x * (int)a.m_per_chunk + grp * per_group;
    int me = mb + per_group; if (me > (int)a.M) me = (int)a.M;
    const int nsteps = per_group / W16_PS;                       // the same for every group: they share the barriers
    const int steps_g = me > mb ? (me - mb + W16_PS - 1) / W16_PS : 0;        // steps this group really has
    int l_n = mb / a.HWp, l_p0 = mb - l_n * a.HWp;               // image and first pixel of the next step to STAGE (a step never straddles images: 32 | HWp)
    const int HW = a.H * a.W;

    // ---- staging: lane l of an instruction = row l / 10, piece l % 10 of its six (and a bit) rows
    const int lrow = (lane * 13) >> 7, lslot = lane - 10 * lrow;                 // lane / 10, lane % 10
    const unsigned vbg = lslot < 8 ? (unsigned)((lrow * a.Cout + co0 + 8 * lslot) * 2) : BF_OOB;
    const unsigned vbx = lslot < 8 ? (unsigned)(((lrow - (TAPS == 9 ? a.W + 1 : 0)) * a.Cin + ci0 + 8 * lslot) * 2) : BF_OOB;      // (mod 2^32: rows above the image wrap out of range)
    const int ktot = (6 + ips * nix + 3) >> 2;                    // instructions per wave and step
    auto dma_step = [&](int stg) {
        const int img = l_n * ips;
        const unsigned long long ga = (unsigned long long)(dy + (size_t)img * HW * a.Cout), xa = (unsigned long long)(x + (size_t)img * HW * a.Cin);
        const bw_i32x4 rg = {(int)(unsigned)ga, (int)(unsigned)((ga >> 32) & 0xFFFFu), HW * a.Cout * 2, 0x00020000};
        const bw_i32x4 rxi = {(int)(unsigned)xa, (int)(unsigned)((xa >> 32) & 0xFFFFu), HW * a.Cin * 2, 0x00020000};
        const unsigned sb = (unsigned)(size_t)St + (unsigned)(stg * STAGE_BYTES);
        if (TAPS == 9 && t < 16) {
            const unsigned pw = ips == 2 ? (unsigned)(2 * (t & 7)) : (unsigned)(l_p0 + 2 * t), w0 = pw - __umulhi(pw, a.magic_w) * (unsigned)a.W, w1 = (w0 + 1 == (unsigned)a.W) ? 0u : w0 + 1;
            Cm[(stg * 16 + t) * 2 + 0] = (w0 > 0 ? 0x0000ffffu : 0u) | (w1 > 0 ? 0xffff0000u : 0u);
            Cm[(stg * 16 + t) * 2 + 1] = (w0 + 1 < (unsigned)a.W ? 0x0000ffffu : 0u) | (w1 + 1 < (unsigned)a.W ? 0xffff0000u : 0u);
        }
        if (ips == 2) {
            // the pair's second image: its own resources (an odd batch's last pair: zero records, every load delivers zeros)
            const bool has2 = img + 1 < a.N;
            const unsigned long long ga2 = ga + (unsigned long long)HW * a.Cout * 2, xa2 = xa + (unsigned long long)HW * a.Cin * 2;
            const bw_i32x4 rg2 = {(int)(unsigned)ga2, (int)(unsigned)((ga2 >> 32) & 0xFFFFu), has2 ? HW * a.Cout * 2 : 0, 0x00020000};
            const bw_i32x4 rx2 = {(int)(unsigned)xa2, (int)(unsigned)((xa2 >> 32) & 0xFFFFu), has2 ? HW * a.Cin * 2 : 0, 0x00020000};
            for (int k = 0; k < ktot; ++k) {
                const int idx = 4 * k + wave;                     // (uniform)
                if (idx < 3) bw_dma16(rg, sb + (unsigned)(idx * 960), vbg + (unsigned)(6 * idx * a.Cout * 2), 0u);
                else if (idx < 6) bw_dma16(rg2, sb + (unsigned)(19 * W16_LD + (idx - 3) * 960), vbg + (unsigned)(6 * (idx - 3) * a.Cout * 2), 0u);
                else if (idx < 6 + nix) bw_dma16(rxi, sb + (unsigned)(38 * W16_LD + (idx - 6) * 960), vbx + (unsigned)(6 * (idx - 6) * a.Cin * 2), 0u);
                else if (idx < 6 + 2 * nix) bw_dma16(rx2, sb + (unsigned)((38 + 6 * nix + 1) * W16_LD + (idx - 6 - nix) * 960), vbx + (unsigned)(6 * (idx - 6 - nix) * a.Cin * 2), 0u);
                else bw_dma16(rxi, scratch, BF_OOB, 0u);
            }
        } else
        for (int k = 0; k < ktot; ++k) {
            const int idx = 4 * k + wave;                         // (uniform)
            if (idx < 6) bw_dma16(rg, sb + (unsigned)(idx * 960), vbg + (unsigned)((l_p0 + 6 * idx) * a.Cout * 2), 0u);
            else if (idx < 6 + nix) bw_dma16(rxi, sb + (unsigned)(W16_LD + idx * 960), vbx + (unsigned)((l_p0 + 6 * (idx - 6)) * a.Cin * 2), 0u);
            else bw_dma16(rxi, scratch, BF_OOB, 0u);
        }
        l_p0 += W16_PS; if (l_p0 >= a.HWp) { l_p0 = 0; ++l_n; }
    };

    f32x4 acc[TAPS][4];
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposed-read addresses of this lane: row q = li >> 2 of a 4-pixel block, channels 4 (li & 3) .. + 3 of a 16-channel tile
    const int tr_row = 4 * lg + (li >> 2), tr_col = (li & 3) * 8;
    const int a_off = tr_row * W16_LD + tr_col;                  // + 32 t for output-channel tile t, + 16 rows for the second half of K
    const int b_off = tr_row * W16_LD + wave * 32 + tr_col;      // + (r W + s) rows for tap (r, s)

    // prologue: the first nst - 1 steps
    for (int j = 0; j < nst - 1; ++j) if (j < steps_g) dma_step(j);
    for (int it = 0; it < nsteps; ++it) {
        const int stg = it % nst;
        const bool active = it < steps_g;                        // group-uniform: a group whose range ended early only keeps the barriers
        // this step's tiles were requested nst - 1 steps ago; behind them (three stages): the K instructions of the next step, if it exists
        if (active) { if (nst == 3 && it + 1 < steps_g) w16_wait(ktot); else __builtin_amdgcn_s_waitcnt(0x0F70); }
        __syncthreads();                                         // everybody's pieces have landed; every wave is done with the stage of step it - 1
        if (it + nst - 1 < steps_g) dma_step((it + nst - 1) % nst);
        if (active) {
            const char* gs = St + stg * STAGE_BYTES;
            const char* xs = gs + dz_rows * W16_LD;
            // masks of this lane's eight pixels (pairs 2g, 2g+1 and 8+2g, 8+2g+1): {left01, right01, left23, right23}
            const uint4 mk0 = *reinterpret_cast<const uint4*>(Cm + (stg * 16 + 2 * lg) * 2);
            const uint4 mk1 = *reinterpret_cast<const uint4*>(Cm + (stg * 16 + 8 + 2 * lg) * 2);
            union Frag { bf16x8 v; s16x4 h[2]; unsigned u[4]; };
            Frag af[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) { af[tt].h[0] = tr_read(gs + a_off + tt * 32); af[tt].h[1] = tr_read(gs + a_off + tt * 32 + dzh * W16_LD); }
            if constexpr (TAPS == 1) {
                Frag bfr;
                bfr.h[0] = tr_read(xs + b_off); bfr.h[1] = tr_read(xs + b_off + 16 * W16_LD);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[0][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt].v, bfr.v, acc[0][tt], 0, 0, 0);
            } else
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                Frag av[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    if (sx == 1) av[tt] = af[tt];
                    else {
                        av[tt].u[0] = af[tt].u[0] & (sx == 0 ? mk0.x : mk0.y); av[tt].u[1] = af[tt].u[1] & (sx == 0 ? mk0.z : mk0.w);
                        av[tt].u[2] = af[tt].u[2] & (sx == 0 ? mk1.x : mk1.y); av[tt].u[3] = af[tt].u[3] & (sx == 0 ? mk1.z : mk1.w);
                    }
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    Frag bfr;
                    const char* bp = xs + b_off + (r * a.W + sx) * W16_LD;
                    bfr.h[0] = tr_read(bp); bfr.h[1] = tr_read(bp + xh * W16_LD);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[(r * 3 + sx) % TAPS][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[tt].v, bfr.v, acc[(r * 3 + sx) % TAPS][tt], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                             // (the stages are dead: group 1's accumulators go there)
    if constexpr (GROUPS == 2) {
        // acc of group 1 -> LDS -> added by group 0 (fixed order: group 0 + group 1)
        f32x4* red = reinterpret_cast<f32x4*>(smem16);
        if (grp == 1) {
#pragma unroll
            for (int k = 0; k < TAPS; ++k)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) red[(k * 4 + tt) * 256 + t] = acc[k][tt];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int k = 0; k < TAPS; ++k)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[k][tt] += red[(k * 4 + tt) * 256 + t];
    }
    // acc[tap][tt][reg]: co = co0 + 16 tt + 4 lg + reg, ci = ci0 + 16 wave + li
    float* dst = partial + (size_t)blockIdx.x * a.wtotal;
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = co0 + tt * 16 + 4 * lg + reg, ci = ci0 + wave * 16 + li;
                dst[((size_t)k * a.Cout + co) * a.Cin + ci] = acc[k][tt][reg];
            }
}

// ------------------------------------------------------------------------------------------------ round 4: the same operator on a zero-PADDED window, 128-byte rows
// What the kernel above pays per 36 MFMAs of a wave and step (a 16-clock MFMA leaves its SIMD eight issue clocks: two other instructions): 26 transposed reads,
// 32 v_and (the border masks of the s = 0 / s = 2 taps), four LDS-DMA instructions of 6.4 rows (160-byte rows: a quarter of every instruction fetches padding)
// and their address arithmetic — about 70 instructions, i.e. the issue port, not the matrix pipe, sets the pace (0.23 - 0.35 of the peak).  Here, for the maps
// whose width divides 32 (4, 8, 16, 32: the CIFAR-sized specs):
//   * the x window is staged in PADDED coordinates — image row h at window rows (h - h0 + 1) (W + 2) + 1 + w, the two pad columns and the rows outside the image
//     fetched out of range (zeros): a tap which leaves the image reads a zero row, NO masks; a step's 32 pixels are whole image rows, so a lane's window row is
//     the same in every step: the 18 + 8 operand addresses of a lane are registers computed once;
//   * rows are 128 bytes, bank-swizzled on the SOURCE side of the DMA: 32-byte column c of row q lies at c ^ ((q >> 1) & 3) — a transposed read's lane group
//     touches eight consecutive rows (any tap offset), whose four same-parity rows then sit in four different columns: conflict-free like the 160-byte rows,
//     but an instruction moves eight whole rows (dz: 4 instead of 6 instructions per step, the window of a 16-wide map 9 instead of 11);
//   * three stages at compile-time distances (the step loop is unrolled by three): the stage offset is the immediate of the read.
// Maps of at most 16 pixels take two images per step as above (two windows, the K halves of a fragment in different windows).
struct W16pArgs {
    int N, H, W, Cin, Cout;
    int HWp;                        // K-axis pixels per image (per PAIR of images when ips == 2): a multiple of 32
    long M, m_per_chunk;
    size_t wtotal;
    int ips, nx;                    // images per step; 8-row DMA instructions per window
    int sw, segs;                   // the window's geometry: segment width (a power of two <= 32) and segments per VIRTUAL image row (below)
};
// Maps of any width up to 64 (round 4, second step: resnet18_style's 56 / 28 / 14 / 7): the image is treated as VH x VW VIRTUAL pixels, VW the next power of two,
// cut into segments of `sw` = min(VW, 32) columns; a step is 32 / sw rows of one segment.  Virtual pixels beyond the image's columns are fetched out of range (dz
// zeros: they add nothing), the window's columns come from the neighbouring segment or from beyond the image's edge (zeros); global addresses use the TRUE width.
// A lane's validity pattern depends on the segment only: two sets of staging offsets (segment 0 / 1), picked per step by a scalar.
#define W16P_XOFF 4096              // dz: 32 rows of 128 B
#define W16P_STAGE (4096 + 13 * 1024)      // + the largest window: 104 rows (W = 32); two windows of 40 rows with image pairs
#define W16P_GROUP (3 * W16P_STAGE)

template <int GROUPS, int IPS>
__global__ __launch_bounds__(256 * GROUPS, 2) void conv_bf16_wgrad9p_kernel(W16pArgs a, const u16* __restrict__ x, const u16* __restrict__ dy, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(1024))) char smem16p[];
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), t = threadIdx.x & 255, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned St = (unsigned)(size_t)smem16p + (unsigned)(grp * W16P_GROUP);
    const int li = lane & 15, lg = lane >> 4;
    const int ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
    constexpr int ips = IPS;
    const int Wt = a.W, SW = a.sw, WP = SW + 2, wsh = 31 - __builtin_clz((unsigned)SW), NX = a.nx, ssh = a.segs > 1 ? 1 : 0;
    const int per_group = (int)(a.m_per_chunk / GROUPS);         // multiple of 32
    const int mb = (int)blockIdx.x * (int)a.m_per_chunk + grp * per_group;
    int me = mb + per_group; if (me > (int)a.M) me = (int)a.M;
    const int nsteps = per_group / W16_PS;
    const int steps_g = me > mb ? (me - mb + W16_PS - 1) / W16_PS : 0;
    int l_n = __builtin_amdgcn_readfirstlane(mb / a.HWp), l_p0 = __builtin_amdgcn_readfirstlane(mb - (mb / a.HWp) * a.HWp);      // (scalars: the division runs on the vector unit)
    const int HW = a.H * a.W;

    // ---- operand addresses of this lane (bytes inside a stage, + the group's base): pixel j0 of each K half
    const int j0 = 4 * lg + (li >> 2), sub = (li & 3) * 8;
    auto wpos = [&](int j) { return (j >> wsh) * WP + (j & (SW - 1)); };      // K-axis pixel -> window row (before the tap offset: the centre tap is + WP + 1)
    // (computed BEHIND the first two steps' requests — `operand_addresses()` below —: a few hundred instructions which used to run in front of them)
    unsigned aaddr[2][4], baddr[2][9];
    auto operand_addresses = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int ra = j0 + 16 * hf;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) aaddr[hf][tt] = St + (unsigned)(ra * 128 + ((tt ^ ((ra >> 1) & 3)) << 5) + sub);
        const int rb = ips == 2 ? hf * NX * 8 + wpos(j0) : wpos(j0 + 16 * hf);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const int row = rb + r * WP + sx;
                baddr[hf][r * 3 + sx] = St + (unsigned)(W16P_XOFF + row * 128 + ((wave ^ ((row >> 1) & 3)) << 5) + sub);
            }
    }
    // (opaque to the optimiser: left visible, the 26 addresses are re-derived from their common terms in front of every read — an addition per read)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) asm volatile("" : "+v"(aaddr[hf][tt]));
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) asm volatile("" : "+v"(baddr[hf][tp]));
    }
    };
    // ---- staging, by wave ROLES (every per-step decision is a scalar of the wave; a step issues no dummy instructions: each wave counts its own requests):
    // wave 0 stages the dz tile (4 instructions of 8 rows; image pairs: rows 16 .. 31 from the second image), waves 1 - 3 the window instructions
    // j = wave - 1, wave + 2, ... (image pairs: instructions NX .. 2 NX - 1 are the second image's window).  Lane = row 8 j + lane / 8, 16-byte piece lane % 8,
    // which holds logical piece (lane % 8) ^ 2 ((row >> 1) & 3).
    const bool dzw = wave == 0;
    const int Cw = dzw ? a.Cout : a.Cin;
    const int cnt = dzw ? 4 : (ips * NX - (wave - 1) + 2) / 3;    // requests of this wave per step (<= 5)
    const int n0 = dzw ? (ips == 2 ? 2 : 4) : (ips == 2 ? (NX - (wave - 1) + 2) / 3 : 5);      // the first n0 of them read the step's first image
    unsigned voff[2][5];                                         // [segment][request]: offsets relative to the step's first pixel (h0, c0)
    const unsigned wp_magic = 65536u / (unsigned)WP + 1u;        // WP = 6 / 10 / 18 / 34: one division instead of ten
    const unsigned sdst0 = dzw ? 0u : (unsigned)(W16P_XOFF + (wave - 1) * 1024), sdstep = dzw ? 1024u : 3072u;      // LDS destination of request i: sdst0 + i sdstep
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int r8 = lane >> 3, piece = lane & 7;
            unsigned v = BF_OOB;
            if (dzw) {
                const int row = 8 * i + r8, j = ips == 2 ? (row & 15) : row, jr = j >> wsh, jc = j & (SW - 1);
                if (sg * SW + jc < Wt) v = (unsigned)(((jr * Wt + jc) * a.Cout + co0) * 2 + ((piece ^ (((row >> 1) & 3) << 1)) << 4));
            } else {
                const int jx = (wave - 1) + 3 * i, wi = jx >= NX ? jx - NX : jx;
                const int q = 8 * wi + r8, hh = (int)(((unsigned)q * wp_magic) >> 16), ww = q - hh * WP, col = sg * SW - 1 + ww;      // (q / WP: exact for q < 2000)
                if (col >= 0 && col < Wt) v = (unsigned)((((hh - 1) * Wt + ww - 1) * a.Cin + ci0) * 2 + ((piece ^ (((q >> 1) & 3) << 1)) << 4));      // (mod 2^32: the row above the image wraps out of range)
            }
            voff[sg][i] = v;
            asm volatile("" : "+v"(voff[sg][i]));
        }
    const u16* tens = dzw ? dy : x;
    const unsigned img_bytes = (unsigned)(HW * Cw * 2);
    // the step's image (first image of the pair) as a byte address, advanced with l_n (two scalar registers instead of a 64-bit product per step)
    unsigned long long ibase = (unsigned long long)tens + (unsigned long long)(l_n * ips) * img_bytes;
    const unsigned sbase = St + sdst0;
    auto dma_step = [&](int stg) {
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)ibase), bhi = __builtin_amdgcn_readfirstlane((unsigned)(ibase >> 32));
        const bw_i32x4 r0 = {(int)blo, (int)(bhi & 0xFFFFu), (int)img_bytes, 0x00020000};
        // the step's first pixel: step t of the image = (row block t >> ssh, segment t & ssh)
        const int tstep = l_p0 >> 5, seg = tstep & ssh, h0 = (tstep >> ssh) << (5 - wsh);
        const unsigned adv = (unsigned)(((h0 * Wt + (seg << wsh)) * Cw) * 2);
        int c = __builtin_amdgcn_readfirstlane(cnt), c0 = __builtin_amdgcn_readfirstlane(n0);
        unsigned sb2 = __builtin_amdgcn_readfirstlane(sbase);
        asm volatile("" : "+s"(c), "+s"(c0), "+s"(sb2));         // (loop-variant to the optimiser: hoisted, the comparisons and the fifteen destinations below live in
                                                                 //  scalar registers the kernel does not have, and the step counters move to vector registers)
        unsigned dst = sb2 + (unsigned)(stg * W16P_STAGE);
        if constexpr (IPS == 2) {
            const unsigned long long b1 = ibase + img_bytes;
            const bool has2 = l_n * 2 + 1 < a.N;                 // (an odd batch's last pair: zero records, every load delivers zeros)
            const bw_i32x4 r1 = {(int)__builtin_amdgcn_readfirstlane((unsigned)b1), (int)(__builtin_amdgcn_readfirstlane((unsigned)(b1 >> 32)) & 0xFFFFu), has2 ? (int)img_bytes : 0, 0x00020000};
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (i < c) { if (i < c0) bw_dma16(r0, dst, voff[0][i], 0u); else bw_dma16(r1, dst, voff[0][i], 0u); }
                dst += sdstep;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (i < c) bw_dma16(r0, dst, (seg ? voff[1][i] : voff[0][i]) + adv, 0u);
                dst += sdstep;
            }
        }
        l_p0 += W16_PS;
        if (l_p0 >= a.HWp) { l_p0 = 0; ++l_n; ibase += (unsigned long long)ips * img_bytes; }
        l_p0 = __builtin_amdgcn_readfirstlane(l_p0); l_n = __builtin_amdgcn_readfirstlane(l_n);
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    union Frag { bf16x8 v; s16x4 h[2]; };
    auto body = [&](auto stgc, int it) __attribute__((always_inline)) {
        constexpr int STG = decltype(stgc)::value;
        const bool active = it < steps_g;                        // group-uniform: a group whose range ended early only keeps the barriers
        if (active) { if (it + 1 < steps_g) w16_wait(cnt); else __builtin_amdgcn_s_waitcnt(0x0F70); }
        __syncthreads();                                         // everybody's pieces of this step have landed; every wave is done with the stage of step it - 1
        if (it + 2 < steps_g) dma_step((STG + 2) % 3);
        if (active) {
            constexpr unsigned SO = (unsigned)(STG * W16P_STAGE);
            Frag af[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) { af[tt].h[0] = tr_read_lds(aaddr[0][tt] + SO); af[tt].h[1] = tr_read_lds(aaddr[1][tt] + SO); }
            // the x fragments of tap t + 1 are requested in front of tap t's MFMAs (two fragment registers sets): a read has four MFMAs to land
            Frag bq[2];
            bq[0].h[0] = tr_read_lds(baddr[0][0] + SO); bq[0].h[1] = tr_read_lds(baddr[1][0] + SO);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) { bq[(tap + 1) & 1].h[0] = tr_read_lds(baddr[0][tap + 1] + SO); bq[(tap + 1) & 1].h[1] = tr_read_lds(baddr[1][tap + 1] + SO); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[tap][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt].v, bq[tap & 1].v, acc[tap][tt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int j = 0; j < 2; ++j) if (j < steps_g) dma_step(j);
    operand_addresses();
    for (int it = 0; it < nsteps; it += 3) {
        body(std::integral_constant<int, 0>{}, it);
        if (it + 1 < nsteps) body(std::integral_constant<int, 1>{}, it + 1);
        if (it + 2 < nsteps) body(std::integral_constant<int, 2>{}, it + 2);
    }
    __syncthreads();                                             // (the stages are dead: group 1's accumulators go there)
    if constexpr (GROUPS == 2) {
        f32x4* red = reinterpret_cast<f32x4*>(smem16p);
        if (grp == 1) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) red[(k * 4 + tt) * 256 + t] = acc[k][tt];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[k][tt] += red[(k * 4 + tt) * 256 + t];
    }
    // acc[tap][tt][reg]: co = co0 + 16 tt + 4 lg + reg, ci = ci0 + 16 wave + li
    // a store = a wave-uniform row pointer (tap, 16-channel tile, register: scalar arithmetic) + the lane's own 32-bit offset (the 64-bit index product per store
    // was ~1200 vector instructions on the one wave per SIMD which is left here)
    float* dst = partial + (size_t)blockIdx.x * a.wtotal + (size_t)co0 * a.Cin + ci0 + wave * 16;
    const unsigned lane_off = (unsigned)(4 * lg * a.Cin + li);
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float* row = dst + ((size_t)k * a.Cout + tt * 16 + reg) * a.Cin;
                row[lane_off] = acc[k][tt][reg];
            }
}

// ------------------------------------------------------------------------------------------------ round 4: the one-tap weight gradient as a proper GEMM
// dW[co][k] = sum over pixels of dz[p][co] xm[p][k] for a matrix xm [n hw][Kc] (a first layer's im2col matrix, the gathered matrix of a strided layer).  The one-tap
// instance of the all-taps kernel above gives a wave one 16-column group of k and four MFMAs per step for ten transposed reads (2.5 LDS reads per MFMA: 600 us per
// call on resnet18_style at bs 512, 15 % of its step).  Here a block owns 64 output channels x NT groups of 64 columns: a wave's x fragments feed 4 NT MFMAs per
// 8 + 2 NT reads (1.0 per MFMA at NT = 4), the tiles staged as in the padded-window kernel (128-byte source-swizzled rows, LDS-DMA of eight rows, three stages at
// compile-time distances); every wave issues one dz instruction and NT x instructions per step (its rows 8 w .. 8 w + 7 of every tile).
struct W16gArgs {
    int N, HW, HWp, Kc, Cout; long M, m_per_chunk; size_t wtotal;
    // IMPL: the matrix is never written — column group g = (tap, 64 channels cg of cig) of pixel (ho, wo) is the 128 bytes of the layer's INPUT at
    // ((ho sh - ph + r dh) W + wo sw - pw + s dw) pix_bytes + 128 cg (zeros outside the image: rows by the per-image buffer range, columns by a per-lane test)
    int Wo, W, pix_bytes, sh, sw, ph, pw, dh, dw, S, cig; unsigned magic_wo, x_img;
};
#define W16G_STAGE(CT, NT) ((CT) * 1024 + (NT) * 4096)

// CT: 16-channel output tiles per wave (4 or 8: the block owns 64 or 128 output channels).  What a block moves per step is 32 pixels x (16 CT + 64 NT) values for
// 32 x 16 CT x 64 NT multiply-adds: 51 flop per byte at 64 x 256, 85 at 128 x 256 — the L2 -> LDS stream, not the matrix pipe, is what the 64-channel tile ran at
// (8.4 TB/s summed over the chip at 485 TFLOP/s on 256 -> 512 / stride 2).
template <int GROUPS, int CT, int NT, bool IMPL = false>
__global__ __launch_bounds__(256 * GROUPS, 2) void conv_bf16_wgradg_kernel(W16gArgs a, const u16* __restrict__ xm, const u16* __restrict__ dy, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(1024))) char smem16g[];
    constexpr int STAGE = W16G_STAGE(CT, NT), GROUP = 3 * STAGE, XOFF = CT * 1024, DS = CT / 4;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), t = threadIdx.x & 255, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned St = (unsigned)(size_t)smem16g + (unsigned)(grp * GROUP);
    const int li = lane & 15, lg = lane >> 4;
    const int k0 = blockIdx.y * 64 * NT, co0 = blockIdx.z * 16 * CT;
    const int per_group = (int)(a.m_per_chunk / GROUPS);
    const int mb = (int)blockIdx.x * (int)a.m_per_chunk + grp * per_group;
    int me = mb + per_group; if (me > (int)a.M) me = (int)a.M;
    const int nsteps = per_group / W16_PS;
    const int steps_g = me > mb ? (me - mb + W16_PS - 1) / W16_PS : 0;
    int l_n = __builtin_amdgcn_readfirstlane(mb / a.HWp), l_p0 = __builtin_amdgcn_readfirstlane(mb - (mb / a.HWp) * a.HWp);
    // operand addresses: pixel j0 of each K half; dz tile tt & 3 of the first 64 channels (the second 64: + 4096), this wave's 16 columns of x tile 0 (tile q: + 4096 q)
    const int j0 = 4 * lg + (li >> 2), sub = (li & 3) * 8;
    unsigned aaddr[2][4], baddr[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int ra = j0 + 16 * hf;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) { aaddr[hf][tt] = St + (unsigned)(ra * 128 + ((tt ^ ((ra >> 1) & 3)) << 5) + sub); asm volatile("" : "+v"(aaddr[hf][tt])); }
        baddr[hf] = St + (unsigned)(XOFF + ra * 128 + ((wave ^ ((ra >> 1) & 3)) << 5) + sub);
        asm volatile("" : "+v"(baddr[hf]));
    }
    // staging: this wave's rows 8 wave .. 8 wave + 7 of every 4 KB sub-tile (32 pixels x 128 bytes; the lane's 16-byte piece holds logical piece ^ 2 ((row >> 1) & 3))
    const int row = 8 * wave + (lane >> 3), piece = lane & 7, swz = (piece ^ (((row >> 1) & 3) << 1)) << 4;
    unsigned vdz = (unsigned)((row * a.Cout + co0) * 2 + swz), vx = (unsigned)((row * a.Kc + k0) * 2 + swz);
    asm volatile("" : "+v"(vdz), "+v"(vx));
    const unsigned dz_img = (unsigned)(a.HW * a.Cout * 2), x_img = IMPL ? a.x_img : (unsigned)(a.HW * a.Kc * 2);
    unsigned long long gbase = (unsigned long long)dy + (unsigned long long)l_n * dz_img, xbase = (unsigned long long)xm + (unsigned long long)l_n * x_img;
    const unsigned sdz = St + (unsigned)(wave * 1024);
    // IMPL: the block's NT column groups as (byte offset of the tap and channel group, the tap's column shift)
    int toff[NT], scol[NT];
    if constexpr (IMPL) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int g = blockIdx.y * NT + q, tap = g / a.cig, cg = g - tap * a.cig, r = tap / a.S, sx = tap - r * a.S;
            toff[q] = __builtin_amdgcn_readfirstlane((r * a.dh * a.W + sx * a.dw) * a.pix_bytes + cg * 128);
            scol[q] = __builtin_amdgcn_readfirstlane(sx * a.dw - a.pw);
        }
    }
    auto dma_step = [&](int stg) {
        const bw_i32x4 rg = {(int)__builtin_amdgcn_readfirstlane((unsigned)gbase), (int)(__builtin_amdgcn_readfirstlane((unsigned)(gbase >> 32)) & 0xFFFFu), (int)dz_img, 0x00020000};
        const bw_i32x4 rx = {(int)__builtin_amdgcn_readfirstlane((unsigned)xbase), (int)(__builtin_amdgcn_readfirstlane((unsigned)(xbase >> 32)) & 0xFFFFu), (int)x_img, 0x00020000};
        unsigned sb2 = __builtin_amdgcn_readfirstlane(sdz);
        asm volatile("" : "+s"(sb2));
        const unsigned dst = sb2 + (unsigned)(stg * STAGE);
        const unsigned vg = vdz + (unsigned)(l_p0 * a.Cout * 2);
#pragma unroll
        for (int q = 0; q < DS; ++q) bw_dma16(rg, dst + (unsigned)(q * 4096), vg, (unsigned)(q * 128));
        if constexpr (IMPL) {
            const unsigned pp = (unsigned)(l_p0 + row), ho = __umulhi(pp, a.magic_wo), wo = pp - ho * (unsigned)a.Wo;
            const int iw0 = (int)wo * a.sw;
            const unsigned vb = (unsigned)((((int)ho * a.sh - a.ph) * a.W + iw0 - a.pw) * a.pix_bytes + swz);      // (mod 2^32: rows above the image wrap out of range)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const bool ok = (unsigned)(iw0 + scol[q]) < (unsigned)a.W;
                bw_dma16(rx, dst + (unsigned)(XOFF + q * 4096), ok ? vb + (unsigned)toff[q] : BF_OOB, 0u);
            }
        } else {
            const unsigned vo = vx + (unsigned)(l_p0 * a.Kc * 2);
#pragma unroll
            for (int q = 0; q < NT; ++q) bw_dma16(rx, dst + (unsigned)(XOFF + q * 4096), vo, (unsigned)(q * 128));
        }
        l_p0 += W16_PS;
        if (l_p0 >= a.HWp) { l_p0 = 0; ++l_n; gbase += dz_img; xbase += x_img; }
        l_p0 = __builtin_amdgcn_readfirstlane(l_p0); l_n = __builtin_amdgcn_readfirstlane(l_n);
    };
    f32x4 acc[NT][CT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[q][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    union Frag { bf16x8 v; s16x4 h[2]; };
    auto body = [&](auto stgc, int it) __attribute__((always_inline)) {
        constexpr int STG = decltype(stgc)::value;
        const bool active = it < steps_g;
        if (active) { if (it + 1 < steps_g) w16_wait(DS + NT); else __builtin_amdgcn_s_waitcnt(0x0F70); }
        __syncthreads();
        if (it + 2 < steps_g) dma_step((STG + 2) % 3);
        if (active) {
            constexpr unsigned SO = (unsigned)(STG * STAGE);
            Frag af[CT], bq[2];
#pragma unroll
            for (int tt = 0; tt < CT; ++tt) {
                af[tt].h[0] = tr_read_lds(aaddr[0][tt & 3] + SO + (unsigned)((tt >> 2) * 4096));
                af[tt].h[1] = tr_read_lds(aaddr[1][tt & 3] + SO + (unsigned)((tt >> 2) * 4096));
            }
            bq[0].h[0] = tr_read_lds(baddr[0] + SO); bq[0].h[1] = tr_read_lds(baddr[1] + SO);
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                if (q + 1 < NT) { bq[(q + 1) & 1].h[0] = tr_read_lds(baddr[0] + SO + (unsigned)((q + 1) * 4096)); bq[(q + 1) & 1].h[1] = tr_read_lds(baddr[1] + SO + (unsigned)((q + 1) * 4096)); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < CT; ++tt) acc[q][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt].v, bq[q & 1].v, acc[q][tt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int j = 0; j < 2; ++j) if (j < steps_g) dma_step(j);
    for (int it = 0; it < nsteps; it += 3) {
        body(std::integral_constant<int, 0>{}, it);
        if (it + 1 < nsteps) body(std::integral_constant<int, 1>{}, it + 1);
        if (it + 2 < nsteps) body(std::integral_constant<int, 2>{}, it + 2);
    }
    __syncthreads();
    if constexpr (GROUPS == 2) {
        f32x4* red = reinterpret_cast<f32x4*>(smem16g);
        if (grp == 1) {
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int tt = 0; tt < CT; ++tt) red[(q * CT + tt) * 256 + t] = acc[q][tt];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int tt = 0; tt < CT; ++tt) acc[q][tt] += red[(q * CT + tt) * 256 + t];
    }
    // acc[q][tt][reg]: co = co0 + 16 tt + 4 lg + reg, k = k0 + 64 q + 16 wave + li; slab [co][Kc]
    float* dst = partial + (size_t)blockIdx.x * a.wtotal;
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int tt = 0; tt < CT; ++tt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                dst[(size_t)(co0 + tt * 16 + 4 * lg + reg) * a.Kc + k0 + 64 * q + wave * 16 + li] = acc[q][tt][reg];
}

inline int w16_groups(int tiles);
// the GEMM's tile (CT x NT), wave groups and chunks for cout x Kc over N images of hw pixels
struct W16gPlan { int ct, nt, groups, chunks, implicit; size_t smem; };
inline bool w16g_choose(int n, int hw, int cout, int kc, W16gArgs* a, W16gPlan* p) {
    static int wide = -1;
    if (wide < 0) { const char* e = getenv("DCV_BF16_WGRADG_WIDE"); wide = e ? atoi(e) : 1; }
    const int hwp = (hw + W16_PS - 1) / W16_PS * W16_PS;
    if (hw < 16 || (long)hwp * 2 > (long)hw * 3 || (size_t)hw * cout * 2 >= ((size_t)1 << 30) || kc % 64 != 0 || cout % 64 != 0) return false;
    const int g64 = kc / 64;
    const int NT = g64 % 4 == 0 ? 4 : g64 % 3 == 0 ? 3 : g64 % 2 == 0 ? 2 : 1;
    const int CT = (wide && cout % 128 == 0) ? 8 : 4;
    const long tiles = (long)(g64 / NT) * (cout / (16 * CT));
    a->N = n; a->HW = hw; a->HWp = hwp; a->Kc = kc; a->Cout = cout;
    a->M = (long)n * hwp;
    a->wtotal = (size_t)cout * kc;
    // chunks: the blocks of one chunk (its tiles) read the same pixels — dz once per column group, the matrix once per output-channel tile.  Workgroups go to the
    // eight XCDs round-robin in launch order (x fastest), so a chunk count that is a multiple of 8 puts a chunk's tiles behind ONE L2 and the re-reads stay there
    // (the matrix of a strided layer is taps x its input: 462 MB for 64 -> 128 / stride 2 at 56 x 56 x 512 images; measured on 256 -> 512 / stride 2: 8 chunks of
    // 72 tiles 122 us, 7 chunks — no second round of blocks, but every chunk spread over all XCDs — 166 us).  One or two wave groups: whichever fills more of the chip.
    const size_t stage = (size_t)W16G_STAGE(CT, NT), red_bytes = (size_t)NT * CT * 256 * 16;
    long best_ch = 0; int best_gr = 1; long best_busy = -1;
    for (int gr = 2; gr >= 1; --gr) {
        const size_t smem = gr == 2 ? (6 * stage > red_bytes ? 6 * stage : red_bytes) : 3 * stage;
        if (smem > 160 * 1024) continue;
        const long slots_xcd = (gr == 2 ? 1 : 2) * DCV_NUM_CU / 8;
        long ch = tiles <= slots_xcd ? 8 * (slots_xcd / tiles) : 8;
        const long unit = (long)gr * W16_PS;
        const long max_by_m = a->M / (4 * unit) > 0 ? a->M / (4 * unit) : 1;
        if (ch > max_by_m) ch = max_by_m >= 8 ? max_by_m / 8 * 8 : max_by_m;
        const long max_by_bytes = (long)(((size_t)256 << 20) / (a->wtotal * sizeof(float)));
        if (ch > max_by_bytes) ch = max_by_bytes >= 8 ? max_by_bytes / 8 * 8 : max_by_bytes;
        if (ch < 1) ch = 1;
        long busy = ch * tiles * gr;                             // wave groups at work (of 2 x 256 slots)
        if (busy > 2 * DCV_NUM_CU) busy = 2 * DCV_NUM_CU;
        if (busy > best_busy) { best_busy = busy; best_ch = ch; best_gr = gr; }
    }
    if (best_busy < 0) return false;
    const int gr = best_gr;
    const long unit = (long)gr * W16_PS;
    long mpc = (a->M + best_ch - 1) / best_ch;
    mpc = (mpc + unit - 1) / unit * unit;
    a->m_per_chunk = mpc;
    p->ct = CT; p->nt = NT; p->groups = gr; p->implicit = 0;
    p->chunks = (int)((a->M + mpc - 1) / mpc);
    p->smem = gr == 2 ? (6 * stage > red_bytes ? 6 * stage : red_bytes) : 3 * stage;
    return true;
}
// ... for a "1x1 layer" descriptor (h, w = the pixel map, cin = Kc): a matrix in memory
inline bool w16g_plan(const dcv_conv2d_desc* d, W16gArgs* a, W16gPlan* p) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_BF16_WGRADG"); enabled = e ? atoi(e) : 1; }
    if (!enabled || d->groups != 1 || d->x_layout != DCV_LAYOUT_NHWC) return false;
    if (d->r != 1 || d->s != 1 || d->pad_h != 0 || d->pad_w != 0 || d->stride_h != 1 || d->stride_w != 1) return false;
    if ((size_t)d->h * d->w * d->cin * 2 >= ((size_t)1 << 30)) return false;
    return w16g_choose(d->n, d->h * d->w, d->cout, d->cin, a, p);
}
inline bool fold_rows(const dcv_conv2d_desc* d);
// ... for a layer of several taps read in place (IMPL): strided / dilated / 5x5 layers of 64-aligned channel counts (the gathered matrix of the earlier version of
// this round — written once, read once: 462 + 462 MB for 64 -> 128 / stride 2 at 56 x 56 x 512 images, the layer's input is 205 MB), and the folded first layer
inline bool w16i_plan(const dcv_conv2d_desc* d, W16gArgs* a, W16gPlan* p) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_BF16_WGRAD_IMPLICIT"); enabled = e ? atoi(e) : 1; }
    if (!enabled || d->groups != 1 || d->x_layout != DCV_LAYOUT_NHWC || d->r * d->s < 2) return false;
    const bool fold = fold_rows(d);
    if (!fold && d->cin % 64 != 0) return false;
    const int cinv = fold ? 64 : d->cin, S = fold ? 1 : d->s, taps = d->r * S;
    if (taps > 64 || (size_t)d->h * d->w * d->cin * 2 >= ((size_t)1 << 30)) return false;
    if (!w16g_choose(d->n, d->ho * d->wo, d->cout, taps * cinv, a, p)) return false;
    if ((size_t)a->HWp * (size_t)d->wo >= ((size_t)1 << 32)) return false;
    p->implicit = 1;
    a->Wo = d->wo; a->W = d->w; a->pix_bytes = d->cin * 2; a->sh = d->stride_h; a->sw = d->stride_w; a->ph = d->pad_h; a->pw = fold ? 0 : d->pad_w;
    a->dh = d->dil_h; a->dw = fold ? 1 : d->dil_w; a->S = S; a->cig = cinv / 64;
    a->magic_wo = 0xFFFFFFFFu / (unsigned)d->wo + 1u; a->x_img = (unsigned)((size_t)d->h * d->w * d->cin * 2);
    return true;
}
template <int GROUPS, int CT, bool IMPL>
inline int w16g_launch(int nt, dim3 grid, size_t smem, hipStream_t st, const W16gArgs& a, const u16* xm, const u16* dy, float* partial) {
#define DCV_W16G(NTV) do { static bool conf = false; if (!conf) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgradg_kernel<GROUPS, CT, NTV, IMPL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH; conf = true; } \
        hipLaunchKernelGGL((conv_bf16_wgradg_kernel<GROUPS, CT, NTV, IMPL>), grid, dim3(256 * GROUPS), smem, st, a, xm, dy, partial); } while (0)
    if (nt == 4) DCV_W16G(4); else if (nt == 3) DCV_W16G(3); else if (nt == 2) DCV_W16G(2); else DCV_W16G(1);
#undef DCV_W16G
    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
}

inline bool w16p_plan(const dcv_conv2d_desc* d, W16pArgs* a, int* chunks, size_t* smem, int* groups) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_BF16_WGRAD9P"); enabled = e ? atoi(e) : 1; }
    if (!enabled || d->groups != 1 || d->x_layout != DCV_LAYOUT_NHWC) return false;
    if (d->r != 3 || d->s != 3 || d->pad_h != 1 || d->pad_w != 1 || d->stride_h != 1 || d->stride_w != 1 || d->dil_h != 1 || d->dil_w != 1) return false;
    if (d->cin % 64 != 0 || d->cout % 64 != 0 || d->ho != d->h || d->wo != d->w) return false;
    if (d->w < 4 || d->w > 64) return false;
    const int hw = d->h * d->w;
    if ((size_t)hw * d->cin * 2 >= ((size_t)1 << 30) || (size_t)hw * d->cout * 2 >= ((size_t)1 << 30)) return false;
    const int ips = (hw == 16 && d->w == 4) ? 2 : 1;            // 4 x 4 maps: two images per step (a K half is exactly one image)
    int vw = 4; while (vw < d->w) vw <<= 1;                     // virtual width
    const int sw = vw < 32 ? vw : 32, segs = vw / sw;
    const int rows_step = ips == 2 ? d->h : W16_PS / sw;        // image rows of a step (per image)
    const int vh = (d->h + rows_step - 1) / rows_step * rows_step;
    if (ips == 1 && (long)vh * vw * 2 > (long)hw * 3) return false;      // more than a third of the virtual pixels would be padding (the masked kernel takes those)
    const int xr = (rows_step + 2) * (sw + 2);
    const int nx = (xr + 7) / 8;
    if (4096 + ips * nx * 1024 > W16P_STAGE) return false;
    int gr = w16_groups((d->cin / 64) * (d->cout / 64));
    a->N = d->n; a->H = d->h; a->W = d->w; a->Cin = d->cin; a->Cout = d->cout; a->ips = ips; a->nx = nx; a->sw = sw; a->segs = segs;
    a->HWp = ips == 2 ? W16_PS : vh * vw;
    a->M = (long)((d->n + ips - 1) / ips) * a->HWp;
    a->wtotal = (size_t)d->cout * d->cin * 9;
    *smem = gr == 2 ? ((size_t)2 * W16P_GROUP + 2048 > (size_t)36 * 256 * 16 ? (size_t)2 * W16P_GROUP + 2048 : (size_t)36 * 256 * 16) : (size_t)W16P_GROUP + 1024;
    *groups = gr;
    const long blocks_per_chunk = (long)(d->cin / 64) * (d->cout / 64);
    const long resident = gr == 2 ? DCV_NUM_CU : 2 * DCV_NUM_CU;
    long ch = (resident + blocks_per_chunk - 1) / blocks_per_chunk;
    const long unit = (long)gr * W16_PS;
    const long max_by_m = a->M / (4 * unit) > 0 ? a->M / (4 * unit) : 1;
    if (ch > max_by_m) ch = max_by_m;
    const long max_by_bytes = (long)(((size_t)256 << 20) / (a->wtotal * sizeof(float)));
    if (ch > max_by_bytes) ch = max_by_bytes;
    if (ch < 1) ch = 1;
    long mpc = (a->M + ch - 1) / ch;
    mpc = (mpc + unit - 1) / unit * unit;
    a->m_per_chunk = mpc;
    *chunks = (int)((a->M + mpc - 1) / mpc);
    return true;
}

// wave groups per block (2: in-block sum of two pixel halves, half the partial slabs).  The slabs are what this operator costs on narrow layers — 512 blocks x
// 147 KB written and read again by the ordered sum: half of the 70 us of 64 -> 64 @32x32 at N = 512.  Measured at N = 512 with the DMA staging (round 3), two
// groups against one: 65.7 vs 69.8 us (64 -> 64 @32x32), 64.1 vs 81.5 (256 -> 256 @8x8), ties at 128 -> 128 and 256 -> 256 @16x16, 192 vs 175 (512 -> 512 @8x8),
// 105 vs 101 (512 -> 512 @4x4): two groups up to 16 (co, ci) tiles.  DCV_BF16_WGRAD9_GROUPS = 1 / 2 forces one setting.
inline int w16_groups(int tiles) {
    static int gr = -1;
    if (gr < 0) { const char* e = getenv("DCV_BF16_WGRAD9_GROUPS"); gr = e ? atoi(e) : 0; if (gr != 1 && gr != 2) gr = 0; }
    return gr ? gr : (tiles <= 16 ? 2 : 1);
}

inline bool w16_plan(const dcv_conv2d_desc* d, W16Args* a, int* chunks, size_t* smem, int* groups) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_BF16_WGRAD9"); enabled = e ? atoi(e) : 1; }
    if (!enabled) return false;
    if (d->groups != 1 || d->x_layout != DCV_LAYOUT_NHWC) return false;
    const bool one = d->r == 1 && d->s == 1 && d->pad_h == 0 && d->pad_w == 0;         // 1x1: the one-tap instance (a plain GEMM over the pixels)
    if (d->stride_h != 1 || d->stride_w != 1 || d->dil_h != 1 || d->dil_w != 1) return false;
    if (!one && (d->r != 3 || d->s != 3 || d->pad_h != 1 || d->pad_w != 1)) return false;
    if (d->cin % 64 != 0 || d->cout % 64 != 0 || d->ho != d->h || d->wo != d->w) return false;
    static int ips_on = -1;
    if (ips_on < 0) { const char* e = getenv("DCV_BF16_WGRAD9_PAIRS"); ips_on = e ? atoi(e) : 1; }
    const int hw = d->h * d->w;
    const int ips = (!one && ips_on && hw <= 16) ? 2 : 1;                              // maps of at most 16 pixels: two images per step (the kernel's header)
    const int XR = one ? W16_PS : (ips == 2 ? 16 : W16_PS) + 2 * d->w + 2;             // rows of ONE window
    if (d->w < 2 || (size_t)hw * d->cin * 2 >= ((size_t)1 << 31) || (size_t)hw * d->cout * 2 >= ((size_t)1 << 31)) return false;      // (32-bit buffer offsets inside an image)
    const int hwp = ips == 2 ? W16_PS : (hw + W16_PS - 1) / W16_PS * W16_PS;           // K-axis pixels per image (per PAIR of images with ips == 2)
    if (hwp > 2 * hw * ips) return false;                                              // more than half of the steps' pixels would be padding
    const int nix = (XR + 5) / 6;
    if (6 + ips * nix > 48) return false;                                              // (at most 12 instructions per wave and step: w16_wait)
    const size_t stage_bytes = (size_t)((ips == 2 ? 38 : 37) + ips * (6 * nix + 1)) * W16_LD;
    auto group_bytes_of = [&](int nst) { return (size_t)nst * stage_bytes + (size_t)nst * 16 * 2 * sizeof(unsigned) + 1024; };
    int gr = w16_groups(one ? 1000 : (d->cin / 64) * (d->cout / 64));      // (the one-tap GEMM of a first layer has few, long chunks: 134 us with one group, 211 with two on the ResNet stem)
    int nst = group_bytes_of(3) * gr <= (size_t)80 * 1024 ? 3 : 2;
    if (gr == 2 && 2 * group_bytes_of(nst) > (size_t)150 * 1024) { gr = 1; nst = group_bytes_of(3) <= (size_t)80 * 1024 ? 3 : 2; }
    const size_t group_bytes = group_bytes_of(nst);
    if (group_bytes > (size_t)80 * 1024) return false;
    *smem = gr == 2 ? (2 * group_bytes > (size_t)36 * 256 * 16 ? 2 * group_bytes : (size_t)36 * 256 * 16) : group_bytes;
    *groups = gr;
    a->stages = nst; a->magic_w = 0xFFFFFFFFu / (unsigned)d->w + 1u;
    a->N = d->n; a->H = d->h; a->W = d->w; a->Cin = d->cin; a->Cout = d->cout; a->XR = XR; a->HWp = hwp; a->ips = ips;
    a->M = (long)((d->n + ips - 1) / ips) * hwp;
    a->wtotal = (size_t)d->cout * d->cin * (one ? 1 : 9);
    const long blocks_per_chunk = (long)(d->cin / 64) * (d->cout / 64);
    const long resident = gr == 2 ? DCV_NUM_CU : 2 * DCV_NUM_CU;                       // blocks resident at once (two waves per SIMD either way): one round
    long ch = (resident + blocks_per_chunk - 1) / blocks_per_chunk;
    const long unit = (long)gr * W16_PS;
    const long max_by_m = a->M / (4 * unit) > 0 ? a->M / (4 * unit) : 1;              // at least four steps per group
    if (ch > max_by_m) ch = max_by_m;
    const long max_by_bytes = (long)(((size_t)256 << 20) / (a->wtotal * sizeof(float)));
    if (ch > max_by_bytes) ch = max_by_bytes;
    if (ch < 1) ch = 1;
    long mpc = (a->M + ch - 1) / ch;
    mpc = (mpc + unit - 1) / unit * unit;
    a->m_per_chunk = mpc;
    *chunks = (int)((a->M + mpc - 1) / mpc);
    return true;
}

// ------------------------------------------------------------------------------------------------ prepared filters, casts
// dst[tap][kcout][kcin] bf16 from the fp32 OIHW master filter (fwd), or its transposed + flipped (sub-)filter (data gradient; conv2d_internal.h)
struct Bf16Prep { const float* w; u16* dst; int kcout, kcin, R, S, w_cin, dgrad, r0, rstep, Rw, s0, sstep, Sw, total, fold; };      // fold: layout [r][co][s][ci] (fold_rows)
#define BF_PREP_MAX 24
struct Bf16PrepList { int count; Bf16Prep e[BF_PREP_MAX]; };

__device__ __forceinline__ u16 f32_to_bf16_bits(float v) { const __bf16 h = (__bf16)v; return *reinterpret_cast<const u16*>(&h); }

__global__ __launch_bounds__(256) void prep_bf16_kernel(Bf16PrepList l) {
    const Bf16Prep& e = l.e[blockIdx.y];
    if (e.rstep == 1 && e.sstep == 1 && e.R == e.Rw && e.S == e.Sw && !e.fold) {
        // whole filter: a thread owns one (kernel-cout, kernel-cin) pair, reads its R*S taps (contiguous in OIHW) and writes one element of every tap plane:
        // reads are 4 R S-byte runs, writes of consecutive threads are consecutive bf16 of a plane (the element-wise form below gathers with stride R*S)
        const int pairs = e.kcout * e.kcin, taps = e.R * e.S;
        if ((e.kcin & 3) == 0 && taps <= 9 && (reinterpret_cast<uintptr_t>(e.dst) & 7) == 0) {
            // four consecutive kernel-cin per thread: one 8-byte store per tap plane instead of four 2-byte ones (round 3: the re-layout of all filters of
            // cifar100_deep took 60 us per step for 114 MB)
            for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pairs / 4; i += gridDim.x * blockDim.x) {
                const int ci = (i % (e.kcin / 4)) * 4, co = i / (e.kcin / 4);
                float v[4][9];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float* src = e.dgrad ? e.w + ((size_t)(ci + k) * e.w_cin + co) * taps : e.w + ((size_t)co * e.w_cin + ci + k) * taps;
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) v[k][tp] = tp < taps ? src[e.dgrad ? taps - 1 - tp : tp] : 0.f;
                }
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
                    if (tp < taps) *reinterpret_cast<bf16x4*>(e.dst + ((size_t)tp * e.kcout + co) * e.kcin + ci) = (bf16x4){(__bf16)v[0][tp], (__bf16)v[1][tp], (__bf16)v[2][tp], (__bf16)v[3][tp]};
            }
            return;
        }
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += gridDim.x * blockDim.x) {
            const int ci = i % e.kcin, co = i / e.kcin;
            const float* src = e.dgrad ? e.w + ((size_t)ci * e.w_cin + co) * taps : e.w + ((size_t)co * e.w_cin + ci) * taps;
            for (int tp = 0; tp < taps; ++tp) e.dst[((size_t)tp * e.kcout + co) * e.kcin + ci] = f32_to_bf16_bits(src[e.dgrad ? taps - 1 - tp : tp]);
        }
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < e.total; i += gridDim.x * blockDim.x) {
        const int ci = i % e.kcin; int q = i / e.kcin;
        const int co = q % e.kcout; const int tap = q / e.kcout;
        const int r = tap / e.S, s = tap - r * e.S;
        float v;
        if (!e.dgrad) v = e.w[(((size_t)co * e.w_cin + ci) * e.Rw + r) * e.Sw + s];
        else v = e.w[(((size_t)ci * e.w_cin + co) * e.Rw + (e.r0 + e.rstep * (e.R - 1 - r))) * e.Sw + (e.s0 + e.sstep * (e.S - 1 - s))];
        e.dst[e.fold ? (((size_t)r * e.kcout + co) * e.S + s) * e.kcin + ci : (size_t)i] = f32_to_bf16_bits(v);
    }
}

// Round 4: the forward AND the data-gradient layout of a whole filter from ONE coalesced read of the master weights, tiled through LDS.  prep_bf16_kernel above
// gathers: a thread reads its 4 x R S taps as 36 four-byte loads 144 B apart from its neighbour's (64 cache lines per load instruction; the data-gradient layout
// reads rows Cin R S floats apart) — 43 us per call, 140 us per step of cifar100_deep at bs 512 (6 % of it, profiles/r04) for 110 MB of traffic.  Here a block owns
// a tile of 32 output x 32 input channels: the 32 runs of 32 R S consecutive floats go to LDS with coalesced loads, then every (tap, output channel) row of the
// forward layout [tap][co][ci] and every (tap, input channel) row of the data-gradient layout [R S - 1 - tap][ci][co] leaves as 64 contiguous bytes (8-byte stores).
struct Bf16Tile { const float* w; u16* dst_f; u16* dst_d; int Cout, Cin, taps, tiles_ci, first; };      // first: index of the entry's first tile in the launch
#define BF_TILE_MAX 32
struct Bf16TileList { int count, total; Bf16Tile e[BF_TILE_MAX]; };
#define BF_TILE_LD (32 * 9 + 1)
__global__ __launch_bounds__(256) void prep_bf16_tile_kernel(Bf16TileList l) {
    __shared__ float tile[32 * BF_TILE_LD];
    int ei = 0;
    while (ei + 1 < l.count && (int)blockIdx.x >= l.e[ei + 1].first) ++ei;                 // (uniform: scalar loop)
    const Bf16Tile& e = l.e[ei];
    const int tl = (int)blockIdx.x - e.first, tco = tl / e.tiles_ci, tci = tl - tco * e.tiles_ci;
    const int co0 = tco * 32, ci0 = tci * 32, taps = e.taps, run = 32 * taps, t = threadIdx.x;
    // (row, column) of element i = t + 256 k walk on without a division (run = 32 taps >= 32); six loads in flight per thread: the kernel is a few hundred
    // blocks of one memory round trip after another (12.7 us per call at 1.5 TB/s before)
    {
        int r = t / run, c = t - r * run;
        const int dr = 256 / run, dc = 256 - dr * run, total = 32 * run;
        const float* wrow = e.w + ((size_t)co0 * e.Cin + ci0) * taps;
        const size_t rstride = (size_t)e.Cin * taps;
        for (int i = t; i < total; i += 6 * 256) {
            float v[6]; int at[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                at[k] = r * BF_TILE_LD + c;
                v[k] = i + k * 256 < total ? wrow[(size_t)r * rstride + c] : 0.f;
                r += dr; c += dc;
                if (c >= run) { c -= run; ++r; }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) if (i + k * 256 < total) tile[at[k]] = v[k];
        }
    }
    __syncthreads();
    if (e.dst_f != nullptr)
        for (int i = t; i < taps * 32 * 8; i += 256) {                                      // (tap, co row, group of four ci)
            const int g = i & 7, r = (i >> 3) & 31, tp = i >> 8;
            const float* s0 = tile + r * BF_TILE_LD + (4 * g) * taps + tp;
            *reinterpret_cast<bf16x4*>(e.dst_f + ((size_t)tp * e.Cout + co0 + r) * e.Cin + ci0 + 4 * g) = (bf16x4){(__bf16)s0[0], (__bf16)s0[taps], (__bf16)s0[2 * taps], (__bf16)s0[3 * taps]};
        }
    if (e.dst_d != nullptr)
        for (int i = t; i < taps * 32 * 8; i += 256) {                                      // (tap, ci row, group of four co)
            const int g = i & 7, c = (i >> 3) & 31, tp = i >> 8;
            const float* s0 = tile + (4 * g) * BF_TILE_LD + c * taps + tp;
            *reinterpret_cast<bf16x4*>(e.dst_d + ((size_t)(taps - 1 - tp) * e.Cin + ci0 + c) * e.Cout + co0 + 4 * g) =
                (bf16x4){(__bf16)s0[0], (__bf16)s0[BF_TILE_LD], (__bf16)s0[2 * BF_TILE_LD], (__bf16)s0[3 * BF_TILE_LD]};
        }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float4* __restrict__ x, bf16x4* __restrict__ y, size_t n4, const float* __restrict__ xt, u16* __restrict__ yt, int tail) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        y[i] = (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) yt[threadIdx.x] = f32_to_bf16_bits(xt[threadIdx.x]);
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const uint2* __restrict__ x, float4* __restrict__ y, size_t n4, const u16* __restrict__ xt, float* __restrict__ yt, int tail) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const uint2 v = x[i];
        y[i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) yt[threadIdx.x] = __uint_as_float((unsigned)xt[threadIdx.x] << 16);
}

// ------------------------------------------------------------------------------------------------ weight gradient of the layers the all-taps kernels do not plan
// (round 4: the 3x3 / stride 2 layers of resnet18_style ran their weight gradient per tap on fp32 copies of x and dz — conv_mfma_wgrad_kernel<true>, 539 us a layer
// at bs 512 plus two cast launches, 9 % of the step).  Here x is gathered ONCE into the bf16 matrix [n ho wo][taps cin] (tap-major, the input channels of a tap
// contiguous: 16-byte pieces), the weight gradient then is the plain GEMM dW[co][tap cin] = sum over pixels of dz[pix][co] xcol[pix][tap cin] on the one-tap instance
// of the all-taps kernel, and the ordered slab sum transposes [co][tap][ci] into OIHW.
struct ColArgs { int H, W, Cin, Ho, Wo, R, S, sh, sw, ph, pw, dh, dw; };
__global__ __launch_bounds__(256) void gather_cols_bf16_kernel(ColArgs a, const uint4* __restrict__ x, uint4* __restrict__ out) {
    // block = one output row (n, ho): pieces (wo, tap, 8 channels); x and out as 16-byte pieces
    const int ho = blockIdx.x % a.Ho, n = blockIdx.x / a.Ho;
    const int c8 = a.Cin >> 3, taps = a.R * a.S, per_pix = taps * c8, total = a.Wo * per_pix;
    const uint4* xn = x + (size_t)n * a.H * a.W * c8;
    uint4* orow = out + ((size_t)n * a.Ho + ho) * a.Wo * per_pix;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int wo = i / per_pix, r1 = i - wo * per_pix, tap = r1 / c8, c = r1 - tap * c8;
        const int r = tap / a.S, sx = tap - r * a.S;
        const int hi = ho * a.sh - a.ph + r * a.dh, wi = wo * a.sw - a.pw + sx * a.dw;
        const bool ok = hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
        const uint4 v = xn[ok ? ((size_t)hi * a.W + wi) * c8 + c : 0];               // (unconditional load: no branch per piece)
        orow[i] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
}
// dw[co][ci][tap] = sum over chunks of partial[chunk][co][tap][ci] (deterministic).  Block = 8 float4 columns x 32 chunk lanes (mfma_wgrad9_reduce_kernel's shape:
// one thread walking all chunks of its element was a chain of up to 256 dependent loads — 34 us per call on resnet18_style's strided layers); a lane adds its
// chunks c = cl, cl + 32, ... in order, the 32 lanes are combined through LDS in lane order.  Cin % 4 == 0.
__global__ __launch_bounds__(256) void col_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int chunks, int Cout, int Cin, int taps) {
    __shared__ float4 red[32][8];
    const int v4 = threadIdx.x & 7, cl = threadIdx.x >> 3;
    const size_t wtotal = (size_t)Cout * Cin * taps;
    const size_t e = ((size_t)blockIdx.x * 8 + v4) * 4;              // index in the slab's order [co][tap][ci]: four consecutive input channels
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < wtotal) {
        const float* src = partial + e;
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
    if (cl == 0 && e < wtotal) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 32; ++k) { const float4 v = red[k][v4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const int ci = (int)(e % Cin); const size_t r1 = e / Cin; const int tap = (int)(r1 % taps), co = (int)(r1 / taps);
        float* d = dw + ((size_t)co * Cin + ci) * taps + tap;
        d[0] = t.x; d[taps] = t.y; d[2 * taps] = t.z; d[3 * taps] = t.w;
    }
}

struct PhaseAxis { int r0, taps, pad, count; };
inline PhaseAxis phase_axis(int a, int size, int R, int stride, int pad) {
    PhaseAxis p{};
    p.r0 = (a + pad) % stride;
    p.taps = p.r0 < R ? (R - p.r0 + stride - 1) / stride : 0;
    const int oa = (a + pad - p.r0) / stride;
    p.pad = p.taps - 1 - oa;
    p.count = a < size ? (size - a + stride - 1) / stride : 0;
    return p;
}

// Round 4: the FOLDED first layer.  A layer of few input channels whose S horizontal taps x cin channels are exactly one K step (S cin = 64: the 7x7 / 2 stem of a
// ResNet after space-to-depth — dcv_space_to_depth_bf16: 4x4 taps over 16 channels) needs no im2col matrix: in NHWC the S pixels of a tap row ARE 64 consecutive
// values, so the layer runs as R taps of 64 "channels" over a view of the tensor whose pixels overlap (pixel stride cin, not 64).  Forward: the implicit-GEMM kernel
// with Bf16Args::pix; weight gradient: the implicit instance of the GEMM kernel (w16i_plan).  No padding or dilation along w (every tap of a row is inside the image).
inline bool fold_rows(const dcv_conv2d_desc* d) {
    return d->cin < 64 && d->cin % 8 == 0 && d->cin * d->s == 64 && d->pad_w == 0 && d->dil_w == 1 && d->groups == 1;
}

inline bool desc_ok(const dcv_conv2d_desc* d) {
    if (!d) return false;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cout <= 0 || d->r <= 0 || d->s <= 0) return false;
    if (d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h <= 0 || d->dil_w <= 0) return false;
    const int ho = (d->h + 2 * d->pad_h - d->dil_h * (d->r - 1) - 1) / d->stride_h + 1;
    const int wo = (d->w + 2 * d->pad_w - d->dil_w * (d->s - 1) - 1) / d->stride_w + 1;
    if (ho != d->ho || wo != d->wo || ho <= 0 || wo <= 0) return false;
    if (d->act < DCV_ACT_NONE || d->act > DCV_ACT_TANH) return false;
    return true;
}

// number of (phase) filter pieces of (d, op) and their element offsets inside the prepared buffer
inline int prep_pieces(const dcv_conv2d_desc* d, int op, Bf16Prep* out, int* offs) {
    Bf16Prep e{};
    e.w_cin = d->cin; e.dgrad = op ? 1 : 0;
    e.kcout = op ? d->cin : d->cout; e.kcin = op ? d->cout : d->cin;
    e.Rw = d->r; e.Sw = d->s; e.fold = (op == 0 && fold_rows(d)) ? 1 : 0;
    if (op == 0 || (d->stride_h == 1 && d->stride_w == 1)) {
        e.R = d->r; e.S = d->s; e.r0 = e.s0 = 0; e.rstep = e.sstep = 1;
        e.total = d->cout * d->cin * d->r * d->s;
        out[0] = e; offs[0] = 0;
        return 1;
    }
    int n = 0, off = 0;
    for (int pa = 0; pa < d->stride_h; ++pa)
        for (int pb = 0; pb < d->stride_w; ++pb) {
            const PhaseAxis ph = phase_axis(pa, d->h, d->r, d->stride_h, d->pad_h), pw = phase_axis(pb, d->w, d->s, d->stride_w, d->pad_w);
            Bf16Prep q = e;
            q.R = ph.taps; q.S = pw.taps; q.r0 = ph.r0; q.s0 = pw.r0; q.rstep = d->stride_h; q.sstep = d->stride_w;
            q.total = d->cout * d->cin * ph.taps * pw.taps;
            out[n] = q; offs[n] = off; off += q.total; ++n;
        }
    return n;
}

// Samples per launch when the tensor a kernel reads (or the pixel count) exceeds what its 32-bit byte offsets address: the largest slice of the batch whose input stays
// below 2 GiB and whose pixel count is a multiple of 128 (so that the slices' statistics rows tile the whole array); 0: no such slice.  Round 4: the stem of
// resnet18_style as im2col + 1x1 at the metric's 512 images per GPU reads a 2.5 GB matrix.
inline size_t slice_limit() {                                       // DCV_BF16_SLICE_BYTES: tests force slicing at small sizes
    static size_t lim = 0;
    if (lim == 0) { const char* e = getenv("DCV_BF16_SLICE_BYTES"); lim = e ? (size_t)atoll(e) : ((size_t)1 << 31); if (lim == 0 || lim > ((size_t)1 << 31)) lim = (size_t)1 << 31; }
    return lim;
}
inline int batch_slice(const dcv_conv2d_desc* d, int op) {
    const size_t in1 = (op == 1 ? (size_t)d->ho * d->wo * d->cout : (size_t)d->h * d->w * d->cin) * 2;
    const size_t out_pix = op == 1 ? (size_t)d->h * d->w : (size_t)d->ho * d->wo;
    size_t ns = (slice_limit() - 1) / in1;
    if (ns > (size_t)d->n) ns = (size_t)d->n;
    while (ns > 0 && (ns * out_pix) % 128 != 0) --ns;
    return (int)ns;
}

inline bool supported(const dcv_conv2d_desc* d, int op) {
    if (!desc_ok(d) || d->groups != 1 || d->x_layout != DCV_LAYOUT_NHWC) return false;
    if (d->r * d->s > 64) return false;                                               // tap validity is a 64-bit mask
    const int kcin = op == 1 ? d->cout : d->cin, kcout = op == 1 ? d->cin : d->cout;
    if ((kcin % BF_BK != 0 && !(op == 0 && fold_rows(d))) || kcout % 64 != 0) return false;
    // 32-bit buffer offsets with headroom for the out-of-range marker: per LAUNCH — a larger batch runs as slices of the batch (batch_slice below)
    const size_t xin1 = (op == 1 ? (size_t)d->ho * d->wo * d->cout : (size_t)d->h * d->w * d->cin) * 2;
    const size_t wbytes = (size_t)d->cout * d->cin * d->r * d->s * 2;
    if (xin1 >= ((size_t)1 << 30) || wbytes >= ((size_t)1 << 31)) return false;
    if ((size_t)d->n * xin1 >= slice_limit() && batch_slice(d, op) <= 0) return false;
    if (op == 1) {
        if (d->stride_h == 1 && d->stride_w == 1) {
            if (d->dil_h * (d->r - 1) < d->pad_h || d->dil_w * (d->s - 1) < d->pad_w) return false;
        } else if (d->dil_h != 1 || d->dil_w != 1 || d->stride_h > 4 || d->stride_w > 4) return false;
    }
    return true;
}

// the window kernel's plan: 3x3 / stride 1 / pad 1 / no dilation on the whole tensor (not a phase of a strided data gradient), enough pixels to give every CU a
// block, and two windows + two filter stages within the CU's LDS.  DCV_BF16_HALO=0 switches it off, =2 forces it for every shape it can run.
inline bool halo_plan(const Bf16Args& a, int bn, int* win_rows, size_t* smem) {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("DCV_BF16_HALO"); mode = e ? atoi(e) : 1; }
    if (mode == 0 || a.phase || a.R != 3 || a.S != 3 || a.sh != 1 || a.sw != 1 || a.ph != 1 || a.pw != 1 || a.dh != 1 || a.dw != 1) return false;
    if (a.Hin != a.Hout || a.Win != a.Wout) return false;
    // measured at N = 512: kernel by kernel (tools/bf16_bench.py) +3 - 4 % on 128 / 256 channels at 16x16 and 8x8, a tie on 512 -> 512 @8x8, -8 % on 64 -> 64 @32x32 and
    // -19 % on 512 -> 512 @4x4 in isolation; inside the captured training step (tools/gpu_halo_cmp.sh, cifar100_deep bs 512, alternating runs on one box) every
    // 3x3 layer on this kernel is the fastest setting: 2.99 - 3.01 ms against 3.04 - 3.05 (only >= 128 channels and >= 8 columns) and 3.07 (off).  So only the
    // pixel count decides (every CU a 256-pixel tile).
    if (mode != 2 && a.M < 8192) return false;
    const int rows = (256 + 2 * a.Win + 2 + 63) / 64 * 64;
    if (rows > BH_WROWS || rows < 256 || a.magic_w == 0u || a.M >= ((long)1 << 31) || a.x_bytes >= 0x80000000u) return false;      // (W <= 56; at least four window pieces: the counted waits of the kernel assume them)
    *win_rows = rows; *smem = (size_t)BH_BST + (size_t)3 * bn * 128;
    return true;
}

struct MomArgs { const u16* yprev; const float* mean; const float* rstd; const u16* addend; };
template <bool OUT_BF16>
int launch(const Bf16Args& a0, const u16* x, const u16* wb, const float* bias, void* y, double* stats, hipStream_t st, const MomArgs* mom = nullptr) {
    Bf16Args a = a0;
    // n / d = mulhi(n, floor((2^32 - 1) / d) + 1) is exact while n d < 2^32 (0: the kernels divide in 64 bits)
    const bool small = a.Wout >= 2 && a.Hout >= 2 && (size_t)a.M * (size_t)(a.Wout > a.Hout ? a.Wout : a.Hout) < ((size_t)1 << 32);
    a.magic_w = small ? 0xFFFFFFFFu / (unsigned)a.Wout + 1u : 0u; a.magic_h = small ? 0xFFFFFFFFu / (unsigned)a.Hout + 1u : 0u;
    static bool configured[2] = {false, false};
    bool wide = a.Cout % 128 == 0;
    {
        int win_rows = 0; size_t hs = 0;
        // few pixels (the deep layers of a CIFAR-sized model at 8x8 / 4x4): 256-pixel x 128-channel tiles leave CUs without a block (512 -> 512 @4x4, N = 512: 128 blocks
        // on 256 CUs); the 64-channel-wide instance has twice the blocks.  DCV_BF16_HALO_WIDE_MIN_BLOCKS: the 128-wide instance needs at least that many tiles.
        static int wide_min = -1;
        if (wide_min < 0) { const char* e = getenv("DCV_BF16_HALO_WIDE_MIN_BLOCKS"); wide_min = e ? atoi(e) : 192; }
        const bool wide_halo = wide && (long)dcv_ceil_div((size_t)a.M, 256) * (a.Cout / 128) >= wide_min;
        if (halo_plan(a, wide_halo ? 128 : 64, &win_rows, &hs)) {
            wide = wide_halo;
            const Bf16Args& ah = a;                               // (Win == Wout, Hin == Hout here: the reciprocals are the window's)
            static bool hconf[2] = {false, false};
            // persistent blocks: one per CU, each walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...
            const unsigned hgy = (unsigned)(a.Cout / (wide ? 128 : 64)), htiles = (unsigned)dcv_ceil_div((size_t)a.M, 256);
            unsigned hgx = wide ? htiles : DCV_NUM_CU / hgy; if (hgx < 1) hgx = 1; if (hgx > htiles) hgx = htiles;      // (the 128-wide instance: one tile per block)
            dim3 hgrid(hgx, hgy);
            if (hgrid.y > 65535) return DCV_ERR_UNSUPPORTED;
            // the 128-wide instance with several channel tiles: a 1-D grid in XCD-aware order (the kernel's header; DCV_BF16_HALO_XCD=0: the 2-D grid)
            static int xcd = -1;
            if (xcd < 0) { const char* e = getenv("DCV_BF16_HALO_XCD"); xcd = e ? atoi(e) : 1; }
            if (wide && xcd && hgy > 1 && 8 * hgy <= DCV_NUM_CU) { a.xcd_tiles = (int)htiles; hgrid = dim3((htiles + 7) / 8 * 8 * hgy, 1); }
            if constexpr (OUT_BF16) {
                // 64 input channels: the filters-in-registers kernel (DCV_BF16_HALO_RF=0: off)
                static int rf = -1;
                if (rf < 0) { const char* e = getenv("DCV_BF16_HALO_RF"); rf = e ? atoi(e) : 1; }
                if (rf && a.Cin == 64) {
                    static bool rconf[2] = {false, false};
                    const unsigned gy = (unsigned)(a.Cout / 64);
                    unsigned gx = DCV_NUM_CU / gy; if (gx < 1) gx = 1; if (gx > htiles) gx = htiles;
                    const size_t rs = (size_t)2 * BH_WSTRIDE + 8192;                   // two windows + the ninth tap's fragments
                    dcv_count_launch(DCV_FAMILY_BF16_WINDOW);
                    if (mom != nullptr) {
                        if (!rconf[1]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_halo_rf_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH; rconf[1] = true; }
                        hipLaunchKernelGGL((conv_bf16_halo_rf_kernel<true>), dim3(gx, gy), dim3(256), rs, st, ah, x, wb, bias, reinterpret_cast<u16*>(y), stats, win_rows, mom->yprev, mom->mean, mom->rstd, mom->addend);
                    } else {
                        if (!rconf[0]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_halo_rf_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH; rconf[0] = true; }
                        hipLaunchKernelGGL((conv_bf16_halo_rf_kernel<false>), dim3(gx, gy), dim3(256), rs, st, ah, x, wb, bias, reinterpret_cast<u16*>(y), stats, win_rows, (const u16*)nullptr, (const float*)nullptr, (const float*)nullptr, (const u16*)nullptr);
                    }
                    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
                }
                if (mom != nullptr) {
                    static bool mconf[2] = {false, false};
                    dcv_count_launch(DCV_FAMILY_BF16_WINDOW);
                    if (wide) {
                        if (!mconf[0]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_halo_kernel<2, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH; mconf[0] = true; }
                        hipLaunchKernelGGL((conv_bf16_halo_kernel<2, true, false, true>), hgrid, dim3(512), hs, st, ah, x, wb, bias, y, stats, win_rows, mom->yprev, mom->mean, mom->rstd, mom->addend);
                    } else {
                        if (!mconf[1]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_halo_kernel<1, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH; mconf[1] = true; }
                        hipLaunchKernelGGL((conv_bf16_halo_kernel<1, true, true, true>), hgrid, dim3(512), hs, st, ah, x, wb, bias, y, stats, win_rows, mom->yprev, mom->mean, mom->rstd, mom->addend);
                    }
                    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
                }
            }
            if (wide) {
                if (!hconf[0]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_halo_kernel<2, OUT_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH; hconf[0] = true; }
                dcv_count_launch(DCV_FAMILY_BF16_WINDOW);
                hipLaunchKernelGGL((conv_bf16_halo_kernel<2, OUT_BF16>), hgrid, dim3(512), hs, st, ah, x, wb, bias, y, stats, win_rows);
            } else {
                if (!hconf[1]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_halo_kernel<1, OUT_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH; hconf[1] = true; }
                dcv_count_launch(DCV_FAMILY_BF16_WINDOW);
                hipLaunchKernelGGL((conv_bf16_halo_kernel<1, OUT_BF16>), hgrid, dim3(512), hs, st, ah, x, wb, bias, y, stats, win_rows);
            }
            return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
        }
    }
    if (mom != nullptr) return DCV_ERR_UNSUPPORTED;                  // (only the window kernel has the moments epilogue)
    if constexpr (OUT_BF16) {
        // a folded first layer (pixel stride below 64 values) on a window staged once per tile (DCV_BF16_FOLDWIN=0: off)
        static int fwin = -1;
        if (fwin < 0) { const char* e = getenv("DCV_BF16_FOLDWIN"); fwin = e ? atoi(e) : 1; }
        const int hw1 = a.Hout * a.Wout;
        // (the tile's pixels + R - 1 rows of the tensor + the 3 extra pixels of a row's last tap + the row ends a tile crosses)
        const int need_px = (BF_BM / a.Wout + 2) * a.sh * a.Win + (a.R - 1) * a.dh * a.Win + (64 / a.pix) + a.sw * a.Wout;
        const int win_px = (need_px + 127) / 128 * 128;
        if (fwin && a.pix < 64 && a.Cin == 64 && a.S == 1 && a.R <= 4 && !a.phase && a.ph == 0 && a.pw == 0 && hw1 % BF_BM == 0 && a.M % BF_BM == 0 && a.Cout % 64 == 0 &&
            a.M < ((long)1 << 31) && win_px <= BFW_MAX_PX && a.stat_rows == 0) {
            static bool wconf = false;
            const size_t ws2 = (size_t)2 * win_px * a.pix * 2 + 8192;
            if (!wconf) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_foldwin_kernel<DCV_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_foldwin_kernel<DCV_ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_foldwin_kernel<-1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH;
                wconf = true;
            }
            const unsigned gy = (unsigned)(a.Cout / 64), ntl = (unsigned)(a.M / BF_BM);
            unsigned gx = 2 * DCV_NUM_CU / gy; if (gx < 1) gx = 1; if (gx > ntl) gx = ntl;
            if (gy > 65535) return DCV_ERR_UNSUPPORTED;
            dcv_count_launch(DCV_FAMILY_BF16_IGEMM);
            u16* y16 = reinterpret_cast<u16*>(y);
            if (a.act == DCV_ACT_NONE) hipLaunchKernelGGL((conv_bf16_foldwin_kernel<DCV_ACT_NONE>), dim3(gx, gy), dim3(256), ws2, st, a, x, wb, bias, y16, stats, win_px);
            else if (a.act == DCV_ACT_RELU) hipLaunchKernelGGL((conv_bf16_foldwin_kernel<DCV_ACT_RELU>), dim3(gx, gy), dim3(256), ws2, st, a, x, wb, bias, y16, stats, win_px);
            else hipLaunchKernelGGL((conv_bf16_foldwin_kernel<-1>), dim3(gx, gy), dim3(256), ws2, st, a, x, wb, bias, y16, stats, win_px);
            return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
        }
        // at most four K steps: the filter-in-registers kernel (DCV_BF16_RFK=0: off)
        static int rfk = -1;
        if (rfk < 0) { const char* e = getenv("DCV_BF16_RFK"); rfk = e ? atoi(e) : 1; }
        const int nk = a.R * a.S * (a.Cin / BF_BK);
        if (rfk && nk <= 4 && a.R * a.S <= 16 && !a.phase && small && a.M < ((long)1 << 31) && a.Cout % 64 == 0) {
            static bool kconf = false;
            const size_t ks = (size_t)RFK_STAGES * BF_BM * 128 + 4096;
            if (!kconf) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_rfk_kernel<DCV_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ks) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_rfk_kernel<DCV_ACT_RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ks) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_rfk_kernel<-1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ks) != hipSuccess) return DCV_ERR_LAUNCH;
                kconf = true;
            }
            const unsigned gy = (unsigned)(a.Cout / 64), ntl = (unsigned)dcv_ceil_div((size_t)a.M, BF_BM);
            unsigned gx = 2 * DCV_NUM_CU / gy; if (gx < 1) gx = 1; if (gx > ntl) gx = ntl;
            if (gy > 65535) return DCV_ERR_UNSUPPORTED;
            dcv_count_launch(DCV_FAMILY_BF16_IGEMM);
            if (a.act == DCV_ACT_NONE) hipLaunchKernelGGL((conv_bf16_rfk_kernel<DCV_ACT_NONE>), dim3(gx, gy), dim3(256), ks, st, a, x, wb, bias, reinterpret_cast<u16*>(y), stats);
            else if (a.act == DCV_ACT_RELU) hipLaunchKernelGGL((conv_bf16_rfk_kernel<DCV_ACT_RELU>), dim3(gx, gy), dim3(256), ks, st, a, x, wb, bias, reinterpret_cast<u16*>(y), stats);
            else hipLaunchKernelGGL((conv_bf16_rfk_kernel<-1>), dim3(gx, gy), dim3(256), ks, st, a, x, wb, bias, reinterpret_cast<u16*>(y), stats);
            return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
        }
    }
    const size_t smem = wide ? (size_t)2 * (BF_BM + 128) * 128 : (size_t)2 * (BF_BM + 64) * 128;
    dim3 grid((unsigned)dcv_ceil_div((size_t)a.M, BF_BM), (unsigned)(a.Cout / (wide ? 128 : 64)));
    if (grid.y > 65535) return DCV_ERR_UNSUPPORTED;
    {   // several channel tiles: the 1-D grid in XCD-aware order (DCV_BF16_HALO_XCD=0: the 2-D grid)
        static int xcd = -1;
        if (xcd < 0) { const char* e = getenv("DCV_BF16_HALO_XCD"); xcd = e ? atoi(e) : 1; }
        if (xcd && grid.y > 1 && 8 * grid.y <= 2 * DCV_NUM_CU && (size_t)(grid.x + 7) / 8 * 8 * grid.y < ((size_t)1 << 31)) { a.xcd_tiles = (int)grid.x; grid = dim3((grid.x + 7) / 8 * 8 * grid.y, 1); }
    }
    dcv_count_launch(DCV_FAMILY_BF16_IGEMM);
    if (wide) {
        if (!configured[0]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<2, 2, OUT_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return DCV_ERR_LAUNCH;
            configured[0] = true;
        }
        hipLaunchKernelGGL((conv_bf16_kernel<2, 2, OUT_BF16>), grid, dim3(256), smem, st, a, x, wb, bias, y, stats);
    } else {
        if (!configured[1]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<4, 1, OUT_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return DCV_ERR_LAUNCH;
            configured[1] = true;
        }
        hipLaunchKernelGGL((conv_bf16_kernel<4, 1, OUT_BF16>), grid, dim3(256), smem, st, a, x, wb, bias, y, stats);
    }
    return (hipGetLastError() == hipSuccess) ? DCV_OK : DCV_ERR_LAUNCH;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// space-to-depth of a strided first layer's fp32 input into bf16 NHWC: out[n][i][j][(a sw + b) cin + c] = x[n][c][i sh + a - ph][j sw + b - pw], zeros outside the
// image and in the channels >= sh sw cin.  A thread writes one 16-byte piece (8 channels) of one pixel.
struct S2dArgs { int H, W, Cin, sh, sw, ph, pw, Hs, Ws, CP, nchw; unsigned total; };
__global__ __launch_bounds__(256) void space_to_depth_bf16_kernel(S2dArgs a, const float* __restrict__ x, uint4* __restrict__ out) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.total) return;
    const unsigned pieces = (unsigned)a.CP >> 3, piece = i % pieces, pixel = i / pieces;
    const unsigned j = pixel % (unsigned)a.Ws, r1 = pixel / (unsigned)a.Ws, ii = r1 % (unsigned)a.Hs, n = r1 / (unsigned)a.Hs;
    const int real = a.sh * a.sw * a.Cin;
    union { uint4 u; __bf16 h[8]; } o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = (int)piece * 8 + k;
        float v = 0.f;
        if (e < real) {
            const int c = e % a.Cin, ab = e / a.Cin, pa = ab / a.sw, pb = ab - pa * a.sw;
            const int ih = (int)ii * a.sh + pa - a.ph, iw = (int)j * a.sw + pb - a.pw;
            if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                v = a.nchw ? x[(((size_t)n * a.Cin + c) * a.H + ih) * a.W + iw] : x[(((size_t)n * a.H + ih) * a.W + iw) * a.Cin + c];
        }
        o.h[k] = (__bf16)v;
    }
    out[i] = o.u;
}

// The ResNet stem's instance (NCHW input, three channels, stride 2: sixteen output channels = (a, b, c) in 2 x 2 x 3 + four zeros) with a thread per OUTPUT PIXEL:
// twelve loads whose lanes walk a plane's row at a stride of two floats (the b = 0 / b = 1 loads of a (c, a) pair cover the same lines), two 16-byte stores of
// consecutive lanes' consecutive pixels, no division (grid = (column blocks, output rows, images)).  The general kernel above decodes (c, a, b) per element with
// integer divisions and moves the same bytes at 2.3 TB/s: 233 us for the 308 + 217 MB of resnet18_style's batch of 512.
__global__ __launch_bounds__(128) void space_to_depth_rgb2_bf16_kernel(S2dArgs a, const float* __restrict__ x, uint4* __restrict__ out) {
    const int j = blockIdx.x * 128 + threadIdx.x, i = blockIdx.y, n = blockIdx.z;
    if (j >= a.Ws) return;
    const float* xn = x + (size_t)n * 3 * a.H * a.W;
    const int iw0 = 2 * j - a.pw;
    float v[12];                                                    // v[(pa * 2 + pb) * 3 + c]
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
        const int ih = 2 * i + pa - a.ph;
        const bool rok = ih >= 0 && ih < a.H;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* row = xn + ((size_t)c * a.H + (rok ? ih : 0)) * a.W;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int iw = iw0 + pb;
                const bool ok = rok && iw >= 0 && iw < a.W;
                const float q = row[ok ? iw : 0];
                v[(pa * 2 + pb) * 3 + c] = ok ? q : 0.f;
            }
        }
    }
    union { uint4 u[2]; __bf16 h[16]; } o;
#pragma unroll
    for (int e = 0; e < 12; ++e) o.h[e] = (__bf16)v[e];
#pragma unroll
    for (int e = 12; e < 16; ++e) o.h[e] = (__bf16)0.f;
    uint4* dst = out + (((size_t)n * a.Hs + i) * a.Ws + j) * 2;
    dst[0] = o.u[0]; dst[1] = o.u[1];
}

}  // namespace

struct Im2colArgs { int H, W, Cin, Ho, Wo, R, S, sh, sw, ph, pw, dh, dw, K, kpad, nchw; unsigned M; int TH, PH, PW, tiles_w, tiles_h; };

// Block = one image's tile of TH x 32 output pixels.  The input patch of the tile ((TH - 1) sh + (R - 1) dh + 1 rows x 31 sw + (S - 1) dw + 1 columns x Cin
// channels, zeros outside the image) goes to LDS once with coalesced loads; a thread then owns ONE 8-element piece of the k axis — its eight patch offsets
// (ci PH PW + r dh PW + s dw) sit in registers — and walks over the tile's pixels: per element one add and one ds_read_b32, per piece one 16-byte store (the lanes
// of a pixel write its kpad row contiguously).  Round 3: the first version decoded (ci, r, s) per element with carries, bounds tests and a gather from global memory:
// ~25 instructions per element, 293 us for the 308 MB of the ResNet stem at bs 64.
__global__ __launch_bounds__(256) void im2col_bf16_kernel(Im2colArgs a, const float* __restrict__ x, u16* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float im_patch[];            // [Cin][PH][PW] + one zero
    const int t = threadIdx.x;
    int b = blockIdx.x;
    const int tw = b % a.tiles_w; b /= a.tiles_w;
    const int th = b % a.tiles_h; const int n = b / a.tiles_h;
    const int ho0 = th * a.TH, wo0 = tw * 32;
    const int hi0 = ho0 * a.sh - a.ph, wi0 = wo0 * a.sw - a.pw;
    const int HW = a.H * a.W, PHW = a.PH * a.PW, psize = a.Cin * PHW;
    const float* xn = x + (size_t)n * a.Cin * HW;
    for (int e = t; e < psize; e += 256) {
        const int ci = e / PHW, r2 = e - ci * PHW, py = r2 / a.PW, px = r2 - py * a.PW;
        const int hi = hi0 + py, wi = wi0 + px;
        const bool ok = hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
        const int off = a.nchw ? ci * HW + hi * a.W + wi : (hi * a.W + wi) * a.Cin + ci;
        const float g = xn[ok ? off : 0];                                       // unconditional (a valid address either way): no branch per element
        im_patch[e] = ok ? g : 0.f;
    }
    if (t == 0) im_patch[psize] = 0.f;
    // this thread's piece of the k axis and its pixel lane
    const int KP = a.kpad >> 3, PL = 256 / KP;                                   // pieces per pixel, pixel lanes
    const int kp = t % KP, lane_p = t / KP;
    int koff[8];
    {
        const int RS = a.R * a.S;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * kp + j;
            const int ci = k / RS, rs = k - ci * RS, r = rs / a.S, sx = rs - r * a.S;
            koff[j] = k < a.K ? ci * PHW + r * a.dh * a.PW + sx * a.dw : -1;
        }
    }
    __syncthreads();
    if (lane_p >= PL) return;
    const int npix = a.TH * 32;
    for (int pl = lane_p; pl < npix; pl += PL) {
        const int py = pl >> 5, px = pl & 31;
        const int ho = ho0 + py, wo = wo0 + px;
        if (ho >= a.Ho || wo >= a.Wo) continue;
        const int base = py * a.sh * a.PW + px * a.sw;
        union { bf16x8 h; uint4 u; } o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.h[j] = (__bf16)im_patch[koff[j] < 0 ? psize : base + koff[j]];
        *reinterpret_cast<uint4*>(out + ((size_t)((size_t)n * a.Ho + ho) * a.Wo + wo) * a.kpad + 8 * kp) = o.u;
    }
}

// the gathered-matrix weight gradient's plan: the layer as a 1x1 layer over [n][ho][wo][taps cin]
inline bool col_plan(const dcv_conv2d_desc* d, dcv_conv2d_desc* d1, W16Args* a, int* chunks, size_t* smem, int* groups, size_t* col_bytes) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DCV_BF16_WGRAD_COL"); enabled = e ? atoi(e) : 1; }
    if (!enabled || !desc_ok(d) || d->groups != 1 || d->x_layout != DCV_LAYOUT_NHWC) return false;
    const int taps = d->r * d->s;
    if (taps < 2 || taps > 25 || d->cin % 64 != 0 || d->cout % 64 != 0) return false;
    const size_t cb = (size_t)d->n * d->ho * d->wo * taps * d->cin * 2;
    if (cb > ((size_t)6 << 30) || (size_t)d->h * d->w * d->cin * 2 >= ((size_t)1 << 31)) return false;
    *d1 = *d;
    d1->h = d->ho; d1->w = d->wo; d1->cin = taps * d->cin; d1->r = d1->s = 1; d1->stride_h = d1->stride_w = 1; d1->pad_h = d1->pad_w = 0; d1->dil_h = d1->dil_w = 1;
    if (!w16_plan(d1, a, chunks, smem, groups)) return false;
    *col_bytes = dcv_align_up(cb, 256);
    return true;
}

// the GEMM weight-gradient launch over a pixel matrix (partial slabs [chunk][co][Kc])
inline int wgradg_run(const W16gArgs& ag, const W16gPlan& p, const u16* xm, const u16* dy, float* partial, hipStream_t st) {
    dim3 grid((unsigned)p.chunks, (unsigned)(ag.Kc / (64 * p.nt)), (unsigned)(ag.Cout / (16 * p.ct)));
    if (grid.y > 65535 || grid.z > 65535) return DCV_ERR_UNSUPPORTED;
    dcv_count_launch(DCV_FAMILY_BF16_WGRAD9);
    if (p.implicit) {
        if (p.ct == 8) return p.groups == 2 ? w16g_launch<2, 8, true>(p.nt, grid, p.smem, st, ag, xm, dy, partial) : w16g_launch<1, 8, true>(p.nt, grid, p.smem, st, ag, xm, dy, partial);
        return p.groups == 2 ? w16g_launch<2, 4, true>(p.nt, grid, p.smem, st, ag, xm, dy, partial) : w16g_launch<1, 4, true>(p.nt, grid, p.smem, st, ag, xm, dy, partial);
    }
    if (p.ct == 8) return p.groups == 2 ? w16g_launch<2, 8, false>(p.nt, grid, p.smem, st, ag, xm, dy, partial) : w16g_launch<1, 8, false>(p.nt, grid, p.smem, st, ag, xm, dy, partial);
    return p.groups == 2 ? w16g_launch<2, 4, false>(p.nt, grid, p.smem, st, ag, xm, dy, partial) : w16g_launch<1, 4, false>(p.nt, grid, p.smem, st, ag, xm, dy, partial);
}

extern "C" {

int dcv_conv2d_bf16_supported(const dcv_conv2d_desc* d, int op) {
    if (op == 2) {
        W16Args a{}; W16pArgs ap{}; int chunks, groups; size_t smem, cb; dcv_conv2d_desc d1; W16gArgs ag{}; W16gPlan gp{};
        return (desc_ok(d) && (dcv_mfma_conv2d_wgrad_in16_supported(d) || w16_plan(d, &a, &chunks, &smem, &groups) || w16p_plan(d, &ap, &chunks, &smem, &groups) ||
                               w16i_plan(d, &ag, &gp) || col_plan(d, &d1, &a, &chunks, &smem, &groups, &cb))) ? 1 : 0;
    }
    return (op == 0 || op == 1) && supported(d, op) ? 1 : 0;
}

size_t dcv_conv2d_bf16_wgrad_workspace_bytes(const dcv_conv2d_desc* d) {
    if (!dcv_conv2d_bf16_supported(d, 2)) return 0;
    size_t need = dcv_mfma_conv2d_wgrad_in16_supported(d) ? dcv_mfma_conv2d_ws(d, 2) : 0;
    W16Args a{}; int chunks, groups; size_t smem;
    if (w16_plan(d, &a, &chunks, &smem, &groups)) { const size_t n16 = dcv_align_up((size_t)chunks * a.wtotal * sizeof(float), 256); if (n16 > need) need = n16; }
    W16gArgs ag{}; W16gPlan gp{};
    if (w16g_plan(d, &ag, &gp)) { const size_t n16 = dcv_align_up((size_t)gp.chunks * ag.wtotal * sizeof(float), 256); if (n16 > need) need = n16; }
    W16pArgs ap{};
    if (w16p_plan(d, &ap, &chunks, &smem, &groups)) { const size_t n16 = dcv_align_up((size_t)chunks * ap.wtotal * sizeof(float), 256); if (n16 > need) need = n16; }
    if ((fold_rows(d) || !dcv_mfma_conv2d_wgrad_in16_supported(d)) && w16i_plan(d, &ag, &gp)) { const size_t n16 = dcv_align_up((size_t)gp.chunks * ag.wtotal * sizeof(float), 256); if (n16 > need) need = n16; }
    dcv_conv2d_desc d1; size_t cb;
    if (need == 0 && col_plan(d, &d1, &a, &chunks, &smem, &groups, &cb)) {
        size_t slabs = (size_t)chunks * a.wtotal * sizeof(float);
        if (w16g_plan(&d1, &ag, &gp) && (size_t)gp.chunks * ag.wtotal * sizeof(float) > slabs) slabs = (size_t)gp.chunks * ag.wtotal * sizeof(float);
        need = cb + dcv_align_up(slabs, 256);
    }
    return need;
}

int dcv_conv2d_bf16_wgrad(const dcv_conv2d_desc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && x && dy && dw);
    if (!dcv_conv2d_bf16_supported(d, 2)) return DCV_ERR_UNSUPPORTED;
    W16Args a{}; int chunks, groups; size_t smem;
    W16pArgs ap{};
    if (w16p_plan(d, &ap, &chunks, &smem, &groups)) {            // the padded-window kernel (maps 4 / 8 / 16 / 32 wide)
        if (!ws || ws_bytes < (size_t)chunks * ap.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
        DCV_CHECK_ARG(aligned16(x) && aligned16(dy) && aligned16(ws));
        dim3 grid((unsigned)chunks, (unsigned)(d->cin / 64), (unsigned)(d->cout / 64));
        if (grid.y > 65535 || grid.z > 65535) return DCV_ERR_UNSUPPORTED;
        static bool pconf = false;
        if (!pconf) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9p_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9p_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9p_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9p_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH;
            pconf = true;
        }
        float* partial = reinterpret_cast<float*>(ws);
        const u16* xs = reinterpret_cast<const u16*>(x); const u16* dys = reinterpret_cast<const u16*>(dy);
        dcv_count_launch(DCV_FAMILY_BF16_WGRAD9);
        if (ap.ips == 2) {
            if (groups == 2) hipLaunchKernelGGL((conv_bf16_wgrad9p_kernel<2, 2>), grid, dim3(512), smem, dcv_stream(stream), ap, xs, dys, partial);
            else hipLaunchKernelGGL((conv_bf16_wgrad9p_kernel<1, 2>), grid, dim3(256), smem, dcv_stream(stream), ap, xs, dys, partial);
        } else {
            if (groups == 2) hipLaunchKernelGGL((conv_bf16_wgrad9p_kernel<2, 1>), grid, dim3(512), smem, dcv_stream(stream), ap, xs, dys, partial);
            else hipLaunchKernelGGL((conv_bf16_wgrad9p_kernel<1, 1>), grid, dim3(256), smem, dcv_stream(stream), ap, xs, dys, partial);
        }
        if (hipGetLastError() != hipSuccess) return DCV_ERR_LAUNCH;
        return dcv_mfma_wgrad9_reduce(partial, dw, chunks, d->cout, d->cin, dcv_stream(stream));
    }
    W16gArgs ag{}; W16gPlan gp{};
    if (w16g_plan(d, &ag, &gp)) {                                // a 1x1 layer over a pixel matrix (a first layer's im2col matrix): the GEMM
        if (!ws || ws_bytes < (size_t)gp.chunks * ag.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
        DCV_CHECK_ARG(aligned16(x) && aligned16(dy) && aligned16(ws));
        float* partial = reinterpret_cast<float*>(ws);
        const int rc = wgradg_run(ag, gp, reinterpret_cast<const u16*>(x), reinterpret_cast<const u16*>(dy), partial, dcv_stream(stream));
        if (rc != DCV_OK) return rc;
        return dcv_mfma_wgrad_reduce_plain(partial, dw, gp.chunks, ag.wtotal, dcv_stream(stream));
    }
    if (w16_plan(d, &a, &chunks, &smem, &groups)) {
        if (!ws || ws_bytes < (size_t)chunks * a.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
        DCV_CHECK_ARG(aligned16(x) && aligned16(dy) && aligned16(ws));
        dim3 grid((unsigned)chunks, (unsigned)(d->cin / 64), (unsigned)(d->cout / 64));
        if (grid.y > 65535 || grid.z > 65535) return DCV_ERR_UNSUPPORTED;
        static bool configured = false;
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH;
            configured = true;
        }
        float* partial = reinterpret_cast<float*>(ws);
        const u16* xs = reinterpret_cast<const u16*>(x); const u16* dys = reinterpret_cast<const u16*>(dy);
        const bool one = d->r == 1;
        dcv_count_launch(DCV_FAMILY_BF16_WGRAD9);
        if (groups == 2) { if (one) hipLaunchKernelGGL((conv_bf16_wgrad9_kernel<2, 1>), grid, dim3(512), smem, dcv_stream(stream), a, xs, dys, partial);
                           else hipLaunchKernelGGL((conv_bf16_wgrad9_kernel<2, 9>), grid, dim3(512), smem, dcv_stream(stream), a, xs, dys, partial); }
        else { if (one) hipLaunchKernelGGL((conv_bf16_wgrad9_kernel<1, 1>), grid, dim3(256), smem, dcv_stream(stream), a, xs, dys, partial);
               else hipLaunchKernelGGL((conv_bf16_wgrad9_kernel<1, 9>), grid, dim3(256), smem, dcv_stream(stream), a, xs, dys, partial); }
        if (hipGetLastError() != hipSuccess) return DCV_ERR_LAUNCH;
        if (one) return dcv_mfma_wgrad_reduce_plain(partial, dw, chunks, a.wtotal, dcv_stream(stream));      // slabs [co][ci] = dw's own order
        return dcv_mfma_wgrad9_reduce(partial, dw, chunks, d->cout, d->cin, dcv_stream(stream));
    }
    if ((fold_rows(d) || !dcv_mfma_conv2d_wgrad_in16_supported(d)) && w16i_plan(d, &ag, &gp)) {      // several taps read in place
        if (!ws || ws_bytes < (size_t)gp.chunks * ag.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
        DCV_CHECK_ARG(aligned16(x) && aligned16(dy) && aligned16(ws));
        float* partial = reinterpret_cast<float*>(ws);
        const int rc = wgradg_run(ag, gp, reinterpret_cast<const u16*>(x), reinterpret_cast<const u16*>(dy), partial, dcv_stream(stream));
        if (rc != DCV_OK) return rc;
        hipLaunchKernelGGL(col_wgrad_reduce_kernel, dim3((unsigned)dcv_ceil_div(ag.wtotal, 32)), dim3(256), 0, dcv_stream(stream), partial, dw, gp.chunks, d->cout, d->cin, d->r * d->s);
        DCV_RETURN_LAUNCH();
    }
    if (!dcv_mfma_conv2d_wgrad_in16_supported(d)) {
        dcv_conv2d_desc d1; size_t cb;
        if (!col_plan(d, &d1, &a, &chunks, &smem, &groups, &cb)) return DCV_ERR_UNSUPPORTED;
        if (!ws || ws_bytes < cb + (size_t)chunks * a.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
        DCV_CHECK_ARG(aligned16(x) && aligned16(dy) && aligned16(ws));
        u16* xcol = reinterpret_cast<u16*>(ws);
        float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + cb);
        const ColArgs ca{d->h, d->w, d->cin, d->ho, d->wo, d->r, d->s, d->stride_h, d->stride_w, d->pad_h, d->pad_w, d->dil_h, d->dil_w};
        hipLaunchKernelGGL(gather_cols_bf16_kernel, dim3((unsigned)(d->n * d->ho)), dim3(256), 0, dcv_stream(stream), ca, reinterpret_cast<const uint4*>(x), reinterpret_cast<uint4*>(xcol));
        if (w16g_plan(&d1, &ag, &gp)) {
            if (ws_bytes < cb + (size_t)gp.chunks * ag.wtotal * sizeof(float)) return DCV_ERR_WORKSPACE;
            const int rc = wgradg_run(ag, gp, xcol, reinterpret_cast<const u16*>(dy), partial, dcv_stream(stream));
            if (rc != DCV_OK) return rc;
            hipLaunchKernelGGL(col_wgrad_reduce_kernel, dim3((unsigned)dcv_ceil_div(ag.wtotal, 32)), dim3(256), 0, dcv_stream(stream), partial, dw, gp.chunks, d->cout, d->cin, d->r * d->s);
            DCV_RETURN_LAUNCH();
        }
        dim3 grid((unsigned)chunks, (unsigned)(d1.cin / 64), (unsigned)(d1.cout / 64));
        if (grid.y > 65535 || grid.z > 65535) return DCV_ERR_UNSUPPORTED;
        static bool cconf = false;
        if (!cconf) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wgrad9_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DCV_ERR_LAUNCH;
            cconf = true;
        }
        dcv_count_launch(DCV_FAMILY_BF16_WGRAD9);
        const u16* dys = reinterpret_cast<const u16*>(dy);
        if (groups == 2) hipLaunchKernelGGL((conv_bf16_wgrad9_kernel<2, 1>), grid, dim3(512), smem, dcv_stream(stream), a, xcol, dys, partial);
        else hipLaunchKernelGGL((conv_bf16_wgrad9_kernel<1, 1>), grid, dim3(256), smem, dcv_stream(stream), a, xcol, dys, partial);
        hipLaunchKernelGGL(col_wgrad_reduce_kernel, dim3((unsigned)dcv_ceil_div(a.wtotal, 32)), dim3(256), 0, dcv_stream(stream), partial, dw, chunks, d->cout, d->cin, d->r * d->s);
        DCV_RETURN_LAUNCH();
    }
    dcv_count_launch(DCV_FAMILY_BF16_WGRAD_TAP);
    return dcv_mfma_conv2d_wgrad_in16(d, x, dy, dw, ws, ws_bytes, dcv_stream(stream));
}

size_t dcv_conv2d_bf16_filter_bytes(const dcv_conv2d_desc* d, int op) {
    if ((op != 0 && op != 1) || !supported(d, op)) return 0;
    return dcv_align_up((size_t)d->cout * d->cin * d->r * d->s * 2, 256);
}

int dcv_conv2d_bf16_prepare_filters(int count, const dcv_conv2d_desc* const* descs, const int* ops, const float* const* w, void* const* filters, void* stream) {
    DCV_CHECK_ARG(count >= 0 && (count == 0 || (descs && ops && w && filters)));
    Bf16PrepList l{};
    int max_total = 0;
    auto flush = [&]() {
        if (l.count == 0) return;
        int gx = (max_total + 255) / 256; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
        hipLaunchKernelGGL(prep_bf16_kernel, dim3((unsigned)gx, (unsigned)l.count), dim3(256), 0, dcv_stream(stream), l);
        l.count = 0; max_total = 0;
    };
    // whole filters of at most nine taps with 32-aligned channel counts: the tiled kernel, the two layouts of one weight from one read (DCV_BF16_PREP_TILED=0: off)
    static int tiled = -1;
    if (tiled < 0) { const char* e = getenv("DCV_BF16_PREP_TILED"); tiled = e ? atoi(e) : 1; }
    Bf16TileList tlist{};
    auto flush_tiles = [&]() {
        if (tlist.count == 0) return;
        hipLaunchKernelGGL(prep_bf16_tile_kernel, dim3((unsigned)tlist.total), dim3(256), 0, dcv_stream(stream), tlist);
        tlist.count = 0; tlist.total = 0;
    };
    for (int k = 0; k < count; ++k) {
        DCV_CHECK_ARG((ops[k] == 0 || ops[k] == 1) && w[k] && filters[k]);
        if (!supported(descs[k], ops[k])) return DCV_ERR_UNSUPPORTED;
        const dcv_conv2d_desc* dk = descs[k];
        if (tiled && dk->r * dk->s <= 9 && dk->cin % 32 == 0 && dk->cout % 32 == 0 && (ops[k] == 0 || (dk->stride_h == 1 && dk->stride_w == 1)) &&
            (reinterpret_cast<uintptr_t>(filters[k]) & 7) == 0) {
            int slot = -1;
            for (int q = 0; q < tlist.count; ++q)
                if (tlist.e[q].w == w[k] && tlist.e[q].Cout == dk->cout && tlist.e[q].Cin == dk->cin && tlist.e[q].taps == dk->r * dk->s &&
                    (ops[k] == 0 ? tlist.e[q].dst_f == nullptr : tlist.e[q].dst_d == nullptr)) { slot = q; break; }
            if (slot < 0) {
                if (tlist.count == BF_TILE_MAX) flush_tiles();
                slot = tlist.count++;
                Bf16Tile& e = tlist.e[slot];
                e = Bf16Tile{w[k], nullptr, nullptr, dk->cout, dk->cin, dk->r * dk->s, dk->cin / 32, tlist.total};
                tlist.total += (dk->cout / 32) * (dk->cin / 32);
            }
            if (ops[k] == 0) tlist.e[slot].dst_f = reinterpret_cast<u16*>(filters[k]); else tlist.e[slot].dst_d = reinterpret_cast<u16*>(filters[k]);
            continue;
        }
        Bf16Prep pieces[DCV_PREP_PIECES]; int offs[DCV_PREP_PIECES];
        const int np = prep_pieces(descs[k], ops[k], pieces, offs);
        for (int q = 0; q < np; ++q) {
            Bf16Prep e = pieces[q];
            if (e.total == 0) continue;
            e.w = w[k]; e.dst = reinterpret_cast<u16*>(filters[k]) + offs[q];
            l.e[l.count++] = e;
            if (e.total > max_total) max_total = e.total;
            if (l.count == BF_PREP_MAX) flush();
        }
    }
    flush();
    flush_tiles();
    DCV_RETURN_LAUNCH();
}

int dcv_conv2d_bf16_fwd_stats_layout(const dcv_conv2d_desc* d, int* rows, int* rows_per_sample) {
    DCV_CHECK_ARG(rows && rows_per_sample);
    *rows = 0; *rows_per_sample = 0;
    if (!supported(d, 0)) return DCV_OK;
    const long M = (long)d->n * d->ho * d->wo;
    *rows = (int)dcv_ceil_div((size_t)M, BF_BM);
    if ((d->ho * d->wo) % BF_BM == 0) *rows_per_sample = d->ho * d->wo / BF_BM;
    return DCV_OK;
}

int dcv_conv2d_bf16_fwd(const dcv_conv2d_desc* d, const void* x, const void* filter, const float* bias, void* y, int y_dtype, double* stats, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && x && filter && y && (y_dtype == DCV_DTYPE_F32 || y_dtype == DCV_DTYPE_BF16));
    if (!supported(d, 0)) return DCV_ERR_UNSUPPORTED;
    DCV_CHECK_ARG(aligned16(x) && aligned16(filter) && aligned16(y) && (!bias || aligned16(bias)));
    Bf16Args a{};
    a.N = d->n; a.Hin = d->h; a.Win = d->w; a.Cin = d->cin; a.pix = d->cin; a.Hout = d->ho; a.Wout = d->wo; a.Cout = d->cout;
    a.R = d->r; a.S = d->s; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    if (fold_rows(d)) { a.Cin = 64; a.S = 1; }                   // a tap row = 64 consecutive values of the tensor: S pixels x cin channels
    a.act = d->act; a.slope = d->act_slope; a.M = (long)d->n * d->ho * d->wo;
    a.w_bytes = (unsigned)((size_t)d->cout * d->cin * d->r * d->s * 2);
    const u16* xs = reinterpret_cast<const u16*>(x); const u16* ws = reinterpret_cast<const u16*>(filter);
    const size_t in1 = (size_t)d->h * d->w * d->cin, out1 = (size_t)d->ho * d->wo * d->cout;      // elements per sample
    if ((size_t)d->n * in1 * 2 < slice_limit()) {
        a.x_bytes = (unsigned)((size_t)d->n * in1 * 2);
        return y_dtype == DCV_DTYPE_BF16 ? launch<true>(a, xs, ws, bias, y, stats, dcv_stream(stream)) : launch<false>(a, xs, ws, bias, y, stats, dcv_stream(stream));
    }
    // the batch in slices (batch_slice): same kernels, the slices' statistics rows side by side in the one array dcv_conv2d_bf16_fwd_stats_layout describes
    const int ns = batch_slice(d, 0);
    if (ns <= 0) return DCV_ERR_UNSUPPORTED;
    const int rows_total = (int)dcv_ceil_div((size_t)d->n * d->ho * d->wo, BF_BM);
    const size_t yel = y_dtype == DCV_DTYPE_BF16 ? 2 : 4;
    for (int n0 = 0; n0 < d->n; n0 += ns) {
        const int nn = d->n - n0 < ns ? d->n - n0 : ns;
        Bf16Args s = a;
        s.N = nn; s.M = (long)nn * d->ho * d->wo; s.x_bytes = (unsigned)((size_t)nn * in1 * 2);
        s.stat_rows = rows_total; s.stat_row0 = (int)((size_t)n0 * d->ho * d->wo / BF_BM);
        void* ys = reinterpret_cast<char*>(y) + (size_t)n0 * out1 * yel;
        const int rc = y_dtype == DCV_DTYPE_BF16 ? launch<true>(s, xs + (size_t)n0 * in1, ws, bias, ys, stats, dcv_stream(stream)) : launch<false>(s, xs + (size_t)n0 * in1, ws, bias, ys, stats, dcv_stream(stream));
        if (rc != DCV_OK) return rc;
    }
    return DCV_OK;
}

int dcv_conv2d_bf16_dgrad(const dcv_conv2d_desc* d, const void* dy, const void* filter, void* dx, int dx_dtype, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && dy && filter && dx && (dx_dtype == DCV_DTYPE_F32 || dx_dtype == DCV_DTYPE_BF16));
    if (!supported(d, 1)) return DCV_ERR_UNSUPPORTED;
    DCV_CHECK_ARG(aligned16(dy) && aligned16(filter) && aligned16(dx));
    hipStream_t st = dcv_stream(stream);
    const u16* dys = reinterpret_cast<const u16*>(dy); const u16* ws = reinterpret_cast<const u16*>(filter);
    if ((size_t)d->n * d->ho * d->wo * d->cout * 2 >= slice_limit()) {      // the batch in slices (batch_slice)
        const int ns = batch_slice(d, 1);
        if (ns <= 0) return DCV_ERR_UNSUPPORTED;
        const size_t in1 = (size_t)d->ho * d->wo * d->cout, out1 = (size_t)d->h * d->w * d->cin, oel = dx_dtype == DCV_DTYPE_BF16 ? 2 : 4;
        for (int n0 = 0; n0 < d->n; n0 += ns) {
            dcv_conv2d_desc ds = *d;
            ds.n = d->n - n0 < ns ? d->n - n0 : ns;
            const int rc = dcv_conv2d_bf16_dgrad(&ds, dys + (size_t)n0 * in1, filter, reinterpret_cast<char*>(dx) + (size_t)n0 * out1 * oel, dx_dtype, stream);
            if (rc != DCV_OK) return rc;
        }
        return DCV_OK;
    }
    Bf16Args a{};
    a.N = d->n; a.Hin = d->ho; a.Win = d->wo; a.Cin = d->cout; a.pix = d->cout; a.Cout = d->cin;
    a.sh = 1; a.sw = 1; a.dh = d->dil_h; a.dw = d->dil_w; a.act = DCV_ACT_NONE; a.slope = 0.f;
    a.x_bytes = (unsigned)((size_t)d->n * d->ho * d->wo * d->cout * 2);
    if (d->stride_h == 1 && d->stride_w == 1) {
        a.Hout = d->h; a.Wout = d->w; a.R = d->r; a.S = d->s;
        a.ph = d->dil_h * (d->r - 1) - d->pad_h; a.pw = d->dil_w * (d->s - 1) - d->pad_w; a.M = (long)d->n * d->h * d->w;
        a.w_bytes = (unsigned)((size_t)d->cout * d->cin * d->r * d->s * 2);
        return dx_dtype == DCV_DTYPE_BF16 ? launch<true>(a, dys, ws, nullptr, dx, nullptr, st) : launch<false>(a, dys, ws, nullptr, dx, nullptr, st);
    }
    Bf16Prep pieces[DCV_PREP_PIECES]; int offs[DCV_PREP_PIECES];
    const int np = prep_pieces(d, 1, pieces, offs);
    bool any_empty = false;
    for (int q = 0; q < np; ++q) any_empty = any_empty || pieces[q].total == 0;
    const size_t esz = dx_dtype == DCV_DTYPE_BF16 ? 2 : 4;
    if (any_empty && hipMemsetAsync(dx, 0, (size_t)d->n * d->h * d->w * d->cin * esz, st) != hipSuccess) return DCV_ERR_LAUNCH;
    int q = 0;
    for (int pa = 0; pa < d->stride_h; ++pa)
        for (int pb = 0; pb < d->stride_w; ++pb, ++q) {
            const PhaseAxis ph = phase_axis(pa, d->h, d->r, d->stride_h, d->pad_h), pw = phase_axis(pb, d->w, d->s, d->stride_w, d->pad_w);
            if (ph.taps == 0 || pw.taps == 0 || ph.count == 0 || pw.count == 0) continue;
            Bf16Args p = a;
            p.Hout = ph.count; p.Wout = pw.count; p.R = ph.taps; p.S = pw.taps; p.ph = ph.pad; p.pw = pw.pad; p.dh = p.dw = 1;
            p.M = (long)d->n * ph.count * pw.count;
            p.phase = 1; p.oh_mul = d->stride_h; p.ow_mul = d->stride_w; p.oh_off = pa; p.ow_off = pb; p.Hfull = d->h; p.Wfull = d->w;
            p.w_bytes = (unsigned)((size_t)pieces[q].total * 2);
            const int rc = dx_dtype == DCV_DTYPE_BF16 ? launch<true>(p, dys, ws + offs[q], nullptr, dx, nullptr, st) : launch<false>(p, dys, ws + offs[q], nullptr, dx, nullptr, st);
            if (rc != DCV_OK) return rc;
        }
    return DCV_OK;
}

// dcv_conv2d_bf16_dgrad (stride 1) whose statistics epilogue leaves the moments of dx the BatchNorm-only layer below needs (the kernel's header); rows of 128 pixels
int dcv_conv2d_bf16_dgrad_moments(const dcv_conv2d_desc* d, const void* dy, const void* filter, void* dx, const void* addend, const void* y_prev, const float* mean,
                                  const float* rstd, double* moments, int moments_rows, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && dy && filter && dx && (addend || y_prev) && (!y_prev || (mean && rstd && moments)));
    if (!supported(d, 1) || d->stride_h != 1 || d->stride_w != 1) return DCV_ERR_UNSUPPORTED;
    DCV_CHECK_ARG(aligned16(dy) && aligned16(filter) && aligned16(dx) && (!y_prev || (aligned16(y_prev) && aligned16(mean) && aligned16(rstd))) && (!addend || aligned16(addend)));
    const long M = (long)d->n * d->h * d->w;
    if (y_prev && moments_rows != (int)dcv_ceil_div((size_t)M, 128)) return DCV_ERR_INVALID_ARG;
    if ((size_t)d->n * d->ho * d->wo * d->cout * 2 >= slice_limit()) return DCV_ERR_UNSUPPORTED;
    Bf16Args a{};
    a.N = d->n; a.Hin = d->ho; a.Win = d->wo; a.Cin = d->cout; a.pix = d->cout; a.Cout = d->cin;
    a.sh = 1; a.sw = 1; a.dh = d->dil_h; a.dw = d->dil_w; a.act = DCV_ACT_NONE; a.slope = 0.f;
    a.x_bytes = (unsigned)((size_t)d->n * d->ho * d->wo * d->cout * 2);
    a.Hout = d->h; a.Wout = d->w; a.R = d->r; a.S = d->s;
    a.ph = d->dil_h * (d->r - 1) - d->pad_h; a.pw = d->dil_w * (d->s - 1) - d->pad_w; a.M = M;
    a.w_bytes = (unsigned)((size_t)d->cout * d->cin * d->r * d->s * 2);
    const MomArgs mom{reinterpret_cast<const u16*>(y_prev), mean, rstd, reinterpret_cast<const u16*>(addend)};
    return launch<true>(a, reinterpret_cast<const u16*>(dy), reinterpret_cast<const u16*>(filter), nullptr, dx, y_prev ? moments : nullptr, dcv_stream(stream), &mom);
}

// im2col of a small-cin first layer: out[m][k] (bf16, row stride kpad) = x[n][ci][ho sh - ph + r dh][wo sw - pw + s dw] for k = (ci R + r) S + s (the OIHW
// filter's own flattening), 0 outside the image and for k >= cin R S.  Thread = (pixel, 8 consecutive k): one 16-byte store; the taps' offsets come from a
// table in the kernel arguments (no division per element).
int dcv_im2col_bf16(const dcv_conv2d_desc* d, const float* x, void* out, int kpad, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && x && out && aligned16(out));
    const int K = d->cin * d->r * d->s;
    if (d->groups != 1 || kpad < K || kpad > DCV_IM2COL_MAX_K || kpad % 64 != 0) return DCV_ERR_UNSUPPORTED;
    const size_t M = (size_t)d->n * d->ho * d->wo;
    if ((size_t)d->cin * d->h * d->w >= ((size_t)1 << 31) || M >= ((size_t)1 << 31)) return DCV_ERR_UNSUPPORTED;      // 32-bit indices inside an image and of pixels (the matrix itself is addressed in 64 bits: 2.5 GB for the resnet18_style stem at bs 512)
    Im2colArgs a{};
    a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Ho = d->ho; a.Wo = d->wo; a.R = d->r; a.S = d->s; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w;
    a.dh = d->dil_h; a.dw = d->dil_w; a.K = K; a.kpad = kpad; a.M = (unsigned)M; a.nchw = d->x_layout == DCV_LAYOUT_NCHW ? 1 : 0;
    // tile height: as many output rows as keep the patch within 48 KB of LDS (at most 8)
    a.PW = 31 * a.sw + (a.S - 1) * a.dw + 1;
    int th = 8;
    for (; th >= 1; --th) { a.PH = (th - 1) * a.sh + (a.R - 1) * a.dh + 1; if (((size_t)a.Cin * a.PH * a.PW + 1) * sizeof(float) <= (size_t)48 * 1024) break; }
    if (th < 1) return DCV_ERR_UNSUPPORTED;
    if (th > d->ho) { th = d->ho; a.PH = (th - 1) * a.sh + (a.R - 1) * a.dh + 1; }
    a.TH = th; a.tiles_w = (d->wo + 31) / 32; a.tiles_h = (d->ho + th - 1) / th;
    const size_t blocks = (size_t)d->n * a.tiles_h * a.tiles_w;
    if (blocks >= ((size_t)1 << 31)) return DCV_ERR_UNSUPPORTED;
    const size_t smem = ((size_t)a.Cin * a.PH * a.PW + 1) * sizeof(float);
    hipLaunchKernelGGL(im2col_bf16_kernel, dim3((unsigned)blocks), dim3(256), smem, dcv_stream(stream), a, x, reinterpret_cast<u16*>(out));
    DCV_RETURN_LAUNCH();
}

int dcv_space_to_depth_bf16(const dcv_conv2d_desc* d, const float* x, void* out, int hs, int ws, int cp, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && x && out && aligned16(out));
    if (d->groups != 1 || d->dil_h != 1 || d->dil_w != 1 || cp % 8 != 0 || cp > 64 || d->stride_h * d->stride_w * d->cin > cp || hs <= 0 || ws <= 0) return DCV_ERR_UNSUPPORTED;
    const size_t total = (size_t)d->n * hs * ws * (cp / 8);
    if (total >= ((size_t)1 << 32) - 256) return DCV_ERR_UNSUPPORTED;
    S2dArgs a{d->h, d->w, d->cin, d->stride_h, d->stride_w, d->pad_h, d->pad_w, hs, ws, cp, d->x_layout == DCV_LAYOUT_NCHW ? 1 : 0, (unsigned)total};
    static int rgb2 = -1;
    if (rgb2 < 0) { const char* e = getenv("DCV_S2D_RGB2"); rgb2 = e ? atoi(e) : 1; }
    if (rgb2 && a.nchw && d->cin == 3 && d->stride_h == 2 && d->stride_w == 2 && cp == 16 && hs <= 65535 && d->n <= 65535)
        hipLaunchKernelGGL(space_to_depth_rgb2_bf16_kernel, dim3((unsigned)dcv_ceil_div(ws, 128), (unsigned)hs, (unsigned)d->n), dim3(128), 0, dcv_stream(stream), a, x, reinterpret_cast<uint4*>(out));
    else
        hipLaunchKernelGGL(space_to_depth_bf16_kernel, dim3((unsigned)dcv_ceil_div(total, 256)), dim3(256), 0, dcv_stream(stream), a, x, reinterpret_cast<uint4*>(out));
    DCV_RETURN_LAUNCH();
}

int dcv_cast_f32_to_bf16(const float* x, void* y, size_t n, void* stream) {
    DCV_CHECK_ARG(x && y && aligned16(x) && (reinterpret_cast<uintptr_t>(y) & 7) == 0);
    if (n == 0) return DCV_OK;
    const size_t n4 = n / 4; const int tail = (int)(n - 4 * n4);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(dcv_stream_grid(n4 ? n4 : 1, 256)), dim3(256), 0, dcv_stream(stream), reinterpret_cast<const float4*>(x), reinterpret_cast<bf16x4*>(y), n4,
                       x + 4 * n4, reinterpret_cast<u16*>(y) + 4 * n4, tail);
    DCV_RETURN_LAUNCH();
}

int dcv_cast_bf16_to_f32(const void* x, float* y, size_t n, void* stream) {
    DCV_CHECK_ARG(x && y && aligned16(y) && (reinterpret_cast<uintptr_t>(x) & 7) == 0);
    if (n == 0) return DCV_OK;
    const size_t n4 = n / 4; const int tail = (int)(n - 4 * n4);
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(dcv_stream_grid(n4 ? n4 : 1, 256)), dim3(256), 0, dcv_stream(stream), reinterpret_cast<const uint2*>(x), reinterpret_cast<float4*>(y), n4,
                       reinterpret_cast<const u16*>(x) + 4 * n4, y + 4 * n4, tail);
    DCV_RETURN_LAUNCH();
}

}  // extern "C"
