// mfma16_probe.hip — what bounds the fp32 implicit-GEMM loop of conv2d_mfma.hip (v_mfma_f32_16x16x4_f32, 256-thread blocks, 256 x 64 tile, K chunks of 16)?
//   mode 0: 64 MFMAs per iteration on 16 accumulator tiles, operands in registers (the matrix pipe's rate at the clock the chip holds)
//   mode 1: + the 8 ds_read_b128 fragment reads of a chunk
//   mode 2: + 5 global float4 loads issued before the MFMAs, 5 ds_write_b128 + one barrier after them (the kernel's loop, addresses advancing by 64 B)
//   mode 3: mode 2 with the MFMAs interleaved with the other instructions by sched_group_barrier
// each with 1, 2, 3 resident blocks per CU (dynamic LDS pads the block to 160 KB / B).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma16_probe.hip -o build/probes/mfma16_probe && build/probes/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(i4 rsrc, unsigned lds_addr, unsigned voffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voffset), "s"(rsrc) : "memory");
}

constexpr int ITER = 576;            // chunks per tile in the probe (16 tiles' worth of the 64 -> 64 3x3 layer)
constexpr int LD = 20;

// WPE: waves per SIMD promised to the compiler: with a register budget <= 256 it keeps the accumulators in VGPRs (v_mfma ... vgprcd), otherwise in AGPRs
template <int MODE, int WPE>
__global__ __launch_bounds__(256, WPE) void loop(const float* __restrict__ x, float* __restrict__ out, size_t xfloats, int iters, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                       // [2][256 * LD]
    float* Bs = lds + 2 * 256 * LD;        // [2][64 * LD]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4, q = t & 3;
    // mode 8 = mode 1 on pseudo-random operands (mode 1's fragments are a 128-value ramp): the matrix pipe's power, and with it the clock the chip holds, depends on the data
    for (int i = t; i < 2 * 320 * LD; i += 256) { unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; lds[i] = MODE == 8 ? ((float)(h & 0xFFFFFF) * (1.f / 8388608.f) - 1.f) : 0.001f * (float)(i & 127); }
    __syncthreads();
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    f4 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i] = (f4){1.f + lane, 2.f, 3.f, 4.f}; bf[i] = (f4){0.5f, 0.25f * lane, 1.f, 2.f}; }
    const float* ap[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ap[j] = x + (((size_t)blockIdx.x * 256 + (t >> 2) + 64 * j) * 64 + 4 * q) % (xfloats - 64 * ITER * 16);
    const float* bp = x + 4 * q + (size_t)(t >> 2) * 576;
    f4 va[4], vb;
    const long long rt0 = wall_clock64(), cc0 = (long long)__builtin_readcyclecounter();
    unsigned junk[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    for (int c = 0; c < iters; ++c) {
        const int buf = c & 1;
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { va[j] = *reinterpret_cast<const f4*>(ap[j]); ap[j] += 16; }
            vb = *reinterpret_cast<const f4*>(bp); bp += 16; if ((c & 31) == 31) bp -= 512;
        }
        if ((MODE >= 1 && MODE <= 3) || MODE == 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f4*>(&As[buf * 256 * LD + (wave * 64 + i * 16 + li) * LD + 4 * lg]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f4*>(&Bs[buf * 64 * LD + (j * 16 + li) * LD + 4 * lg]);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
        if (MODE == 6) {      // 4 LDS-DMA instructions per 64 MFMAs, L2-hot addresses, waited for every 16 iterations (an untracked asynchronous load into REGISTERS from inline
            // assembly is not an option: the compiler reuses the destination registers before the data arrives)
            const unsigned long long xa = (unsigned long long)x;
            const i4 rs = {(int)(unsigned)xa, (int)(unsigned)((xa >> 32) & 0xFFFFu), 1 << 30, 0x00020000};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                dma16(rs, (unsigned)(size_t)lds + (unsigned)(wave * 4096 + v * 1024), (unsigned)(((blockIdx.x & 255) * 64 + v * 4 + (c & 3)) * 1024 + lane * 16));
            }
            if ((c & 15) == 15) __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        if (MODE == 4 || MODE == 5) {      // 64 (mode 4) / 256 (mode 5) independent VALU instructions per 64 MFMAs: do they cost matrix-pipe time?
            constexpr int NV = MODE == 4 ? 8 : 32;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("v_add_u32 %0, %0, 1" : "+v"(junk[u]));
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f4*>(&As[(buf ^ 1) * 256 * LD + ((t >> 2) + 64 * j) * LD + 4 * q]) = va[j];
            *reinterpret_cast<f4*>(&Bs[(buf ^ 1) * 64 * LD + (t >> 2) * LD + 4 * q]) = vb;
            if (MODE == 3) {
                // per chunk: 64 MFMA, 5 VMEM reads, 8 DS reads, 5 DS writes: spread them
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      // 8 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 DS write
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // some VALU
                }
            }
            __syncthreads();
        }
    }
    const long long rt1 = wall_clock64(), cc1 = (long long)__builtin_readcyclecounter();
    if (blockIdx.x == 0 && t == 0 && clk) { clk[0] = rt1 - rt0; clk[1] = cc1 - cc0; }
    float s = (float)(junk[0] + junk[1] + junk[2] + junk[3] + junk[4] + junk[5] + junk[6] + junk[7]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + t] = s;
}

static long long* g_clk = nullptr;
template <int MODE, int WPE = 1>
int run(const float* x, float* out, size_t xfloats, int per_cu, int iters = ITER, int rounds = 4) {
    const size_t smem = per_cu == 1 ? 150 * 1024 : (per_cu == 2 ? 78 * 1024 : 51200);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop<MODE, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int grid = 256 * per_cu * rounds;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((loop<MODE, WPE>), dim3(grid), dim3(256), smem, 0, x, out, xfloats, iters, g_clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = iters == ITER ? 5 : 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((loop<MODE, WPE>), dim3(grid), dim3(256), smem, 0, x, out, xfloats, iters, g_clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double flops = (double)grid * 4 * iters * 64 * 2.0 * 1024;
    printf("mode %d (waves/SIMD promised %d), %d block(s)/CU, %d chunks per block, %d round(s): %.1f us, %.1f TF (%.2f of 157.3)\n", MODE, WPE, per_cu, iters, rounds, ms * 1e3, flops / ms * 1e-9, flops / ms * 1e-9 / 157.3);
    long long h[2]; CK(hipMemcpy(h, g_clk, 16, hipMemcpyDeviceToHost));
    printf("    block 0: %lld ticks of 100 MHz, %lld of the cycle counter -> %.0f MHz\n", h[0], h[1], h[0] ? 100.0 * (double)h[1] / (double)h[0] : 0.0);
    return 0;
}

int main() {
    const size_t xfloats = (size_t)512 * 1024 * 64;     // the 64-channel 32x32 activation of 512 images
    float *x, *out;
    CK(hipMalloc(&g_clk, 16)); CK(hipMemset(g_clk, 0, 16));
    CK(hipMalloc(&x, xfloats * 4)); CK(hipMemset(x, 0, xfloats * 4)); CK(hipMalloc(&out, (size_t)256 * 3 * 8 * 256 * 4));      // the largest grid below: 3 blocks per CU x 8 rounds
    // short kernels: one tile's worth of MFMAs per block (36 chunks), one round of blocks, 50 launches back to back
    for (int b = 1; b <= 3; ++b) { if (run<6, 3>(x, out, xfloats, b)) return 1; if (run<1, 3>(x, out, xfloats, b)) return 1; if (run<8, 3>(x, out, xfloats, b)) return 1; }
    for (int b = 1; b <= 3; ++b) { if (run<4, 3>(x, out, xfloats, b)) return 1; if (run<5, 3>(x, out, xfloats, b)) return 1; }
    for (int b = 1; b <= 3; ++b) { if (run<0, 3>(x, out, xfloats, b, 36, 1)) return 1; if (run<1, 3>(x, out, xfloats, b, 36, 1)) return 1; if (run<0, 3>(x, out, xfloats, b, 36, 8)) return 1; }
    for (int b = 1; b <= 3; ++b) {
        if (run<0>(x, out, xfloats, b)) return 1;
        if (run<1>(x, out, xfloats, b)) return 1;
        if (run<2>(x, out, xfloats, b)) return 1;
        if (run<3>(x, out, xfloats, b)) return 1;
        if (run<0, 3>(x, out, xfloats, b)) return 1;
        if (run<1, 3>(x, out, xfloats, b)) return 1;
        if (run<2, 3>(x, out, xfloats, b)) return 1;
    }
    return 0;
}
