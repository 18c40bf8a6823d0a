// conv4x4_probe.hip — where does the time of the LDS -> LDS 4 -> 4 5x5 convolution on v_mfma_f32_4x4x1_16b_f32 go (rs_conv_c4_co4, resident.hip)?
// One 1024-thread block per CU, two samples of 32 x 32 x 4 in LDS, the convolution + barrier repeated REPS times; variants leave parts out:
//   bit 0: no MFMAs   bit 1: no x reads   bit 2: no filter reads   bit 3: no epilogue (activation, stores, sums)   bit 4: taps fully pipelined (as built) vs rolled rows
//   hipcc --offload-arch=gfx950 -O3 -I deepcv_amd/csrc -I include tools/probes/conv4x4_probe.hip -o tools/bin/conv4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int XIN = 1296, YS = 4096, K = 5, XW = 36, REPS = 200;
__device__ __forceinline__ void lbarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int CTRL> __device__ __forceinline__ float dppm(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false)); }

template <int SKIP>
__device__ __forceinline__ void conv(const f4* xin, float* yt, const float* wl, float* wsum, float bias) {
    int t = threadIdx.x; asm volatile("" : "+v"(t));
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), s = wave >> 3, w8 = wave & 7, j = lane & 3;
    const f4* wl4 = reinterpret_cast<const f4*>(wl) + j;
    const f4* xs = xin + s * XIN;
    float sum = 0.f, sq = 0.f;
    const int g0 = w8, p0 = g0 * 64 + lane, p1 = p0 + 512;
    const f4* x0 = xs + (p0 >> 5) * XW + (p0 & 31);
    const f4* x1 = xs + (p1 >> 5) * XW + (p1 & 31);
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    f4 wv[2], a0[2], a1[2];
    const f4 one = {1.f, 0.5f, 0.25f, 2.f};
    wv[0] = (SKIP & 4) ? one : wl4[0]; a0[0] = (SKIP & 2) ? one : x0[0]; a1[0] = (SKIP & 2) ? one : x1[0];
#pragma unroll
    for (int tap = 0; tap < K * K; ++tap) {
        const int cur = tap & 1, nxt = cur ^ 1;
        if (tap + 1 < K * K) {
            const int ky = (tap + 1) / K, kx = (tap + 1) - ky * K;
            wv[nxt] = (SKIP & 4) ? one : wl4[(tap + 1) * 4]; a0[nxt] = (SKIP & 2) ? one : x0[ky * XW + kx]; a1[nxt] = (SKIP & 2) ? one : x1[ky * XW + kx];
        }
        if (!(SKIP & 1)) {
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[cur].x, wv[cur].x, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[cur].x, wv[cur].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[cur].y, wv[cur].y, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[cur].y, wv[cur].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[cur].z, wv[cur].z, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[cur].z, wv[cur].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[cur].w, wv[cur].w, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[cur].w, wv[cur].w, acc1, 0, 0, 0);
        } else { acc0 += a0[cur] * wv[cur].x; acc1 += a1[cur] * wv[cur].y; }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!(SKIP & 8)) {
        float* y0 = yt + s * YS + (g0 * 64 + (lane & ~3)) * 4 + j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            { const float v = fmaxf(acc0[i] + bias, 0.01f * (acc0[i] + bias)); y0[i * 4] = v; sum += v; sq += v * v; }
            { const float v = fmaxf(acc1[i] + bias, 0.01f * (acc1[i] + bias)); y0[(512 + i) * 4] = v; sum += v; sq += v * v; }
        }
        sum += dppm<0x124>(sum); sq += dppm<0x124>(sq);
        sum += dppm<0x128>(sum); sq += dppm<0x128>(sq);
        sum += __shfl_xor(sum, 16, 64); sq += __shfl_xor(sq, 16, 64);
        sum += __shfl_xor(sum, 32, 64); sq += __shfl_xor(sq, 32, 64);
        if (lane < 4) { wsum[(wave * 2) * 16 + lane] = sum; wsum[(wave * 2 + 1) * 16 + lane] = sq; }
    } else if (acc0[0] + acc1[1] + acc0[2] + acc1[3] == 1234.5f) wsum[0] = 1.f;
}

template <int SKIP>
__global__ __launch_bounds__(1024) void probe(float* out, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f4* xin = reinterpret_cast<f4*>(smem);
    float* yt = reinterpret_cast<float*>(smem + 2 * XIN * 16);
    float* wl = yt + 2 * YS;
    float* wsum = wl + 400;
    for (int i = threadIdx.x; i < 2 * XIN; i += 1024) xin[i] = (f4){seed + i * 1e-3f, seed, -seed, 0.5f};
    for (int i = threadIdx.x; i < 400; i += 1024) wl[i] = seed * 1e-2f + i * 1e-4f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < REPS; ++r) { conv<SKIP>(xin, yt, wl, wsum, seed); lbarrier(); }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(t1 - t0) * 0.01f / REPS;
    if (yt[threadIdx.x] == 1234.5f) out[300] = wsum[3];
}


// variant B: lanes 0 - 31 = row r0, lanes 32 - 63 = row r0 + 2 of the wave's four rows; group 1 = group 0 one row further down, so x[row + ky][col + kx] serves group 0 at
// filter row ky and group 1 at filter row ky - 1: 6 x 5 x-reads for both groups instead of 2 x 25.  DEPTH filter-column steps of operands are in flight.
template <int SKIP, int DEPTH>
__device__ __forceinline__ void convB(const f4* xin, float* yt, const float* wl, float* wsum, float bias) {
    int t = threadIdx.x; asm volatile("" : "+v"(t));
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), s = wave >> 3, w8 = wave & 7, j = lane & 3;
    const f4* wl4 = reinterpret_cast<const f4*>(wl) + j;
    const int r0 = w8 * 4 + (lane >> 5) * 2, c = lane & 31;
    const f4* x0 = xin + s * XIN + r0 * XW + c;
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    // step n = (row ry of the 6 x-rows, kx): x vector a[n]; group 0 uses it with filter row ry (ry < 5), group 1 with filter row ry - 1 (ry >= 1)
    constexpr int NS = 6 * K;
    f4 a[DEPTH + 1], w0[DEPTH + 1], w1[DEPTH + 1];
    auto request = [&](int n, int slot) {
        const int ry = n / K, kx = n - ry * K;
        a[slot] = x0[ry * XW + kx];
        if (ry < K) w0[slot] = wl4[(ry * K + kx) * 4];
        if (ry >= 1 && ry == K) w1[slot] = wl4[((ry - 1) * K + kx) * 4];       // (rows 1 .. 4: group 1's weights are the ones group 0 used one row earlier — re-read only for the last row)
    };
    f4 wprev[K];                                                                 // the previous row's K weight vectors, for group 1
#pragma unroll
    for (int n = 0; n < DEPTH; ++n) request(n, n % (DEPTH + 1));
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int slot = n % (DEPTH + 1), ry = n / K, kx = n - ry * K;
        if (n + DEPTH < NS) request(n + DEPTH, (n + DEPTH) % (DEPTH + 1));
        const f4 av = a[slot];
        if (ry < K) {
            const f4 wv = w0[slot];
            if (!(SKIP & 1)) {
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, wv.x, acc0, 0, 0, 0); acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, wv.y, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, wv.z, acc0, 0, 0, 0); acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, wv.w, acc0, 0, 0, 0);
            } else acc0 += av * wv.x;
        }
        if (ry >= 1) {
            const f4 wv = ry == K ? w1[slot] : wprev[kx];
            if (!(SKIP & 1)) {
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, wv.x, acc1, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, wv.y, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, wv.z, acc1, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, wv.w, acc1, 0, 0, 0);
            } else acc1 += av * wv.y;
        }
        if (ry < K) wprev[kx] = w0[slot];
        __builtin_amdgcn_sched_barrier(0);
    }
    float sum = 0.f, sq = 0.f;
    if (!(SKIP & 8)) {
        // lane (b, j): channel j of pixels (row r0 [+ 1], columns 4 (c / 4) + i)
        float* y0 = yt + s * YS + (r0 * 32 + (c & ~3)) * 4 + j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            { const float v = fmaxf(acc0[i] + bias, 0.01f * (acc0[i] + bias)); y0[i * 4] = v; sum += v; sq += v * v; }
            { const float v = fmaxf(acc1[i] + bias, 0.01f * (acc1[i] + bias)); y0[(32 + i) * 4] = v; sum += v; sq += v * v; }
        }
        sum += dppm<0x124>(sum); sq += dppm<0x124>(sq);
        sum += dppm<0x128>(sum); sq += dppm<0x128>(sq);
        sum += __shfl_xor(sum, 16, 64); sq += __shfl_xor(sq, 16, 64);
        sum += __shfl_xor(sum, 32, 64); sq += __shfl_xor(sq, 32, 64);
        if (lane < 4) { wsum[(wave * 2) * 16 + lane] = sum; wsum[(wave * 2 + 1) * 16 + lane] = sq; }
    } else if (acc0[0] + acc1[1] + acc0[2] + acc1[3] == 1234.5f) wsum[0] = 1.f;
}

template <int SKIP, int DEPTH>
__global__ __launch_bounds__(1024) void probeB(float* out, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f4* xin = reinterpret_cast<f4*>(smem);
    float* yt = reinterpret_cast<float*>(smem + 2 * XIN * 16);
    float* wl = yt + 2 * YS;
    float* wsum = wl + 400;
    for (int i = threadIdx.x; i < 2 * XIN; i += 1024) xin[i] = (f4){seed + i * 1e-3f, seed, -seed, 0.5f};
    for (int i = threadIdx.x; i < 400; i += 1024) wl[i] = seed * 1e-2f + i * 1e-4f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < REPS; ++r) { convB<SKIP, DEPTH>(xin, yt, wl, wsum, seed); lbarrier(); }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(t1 - t0) * 0.01f / REPS;
    if (yt[threadIdx.x] == 1234.5f) out[300] = wsum[3];
}

template <int SKIP, int DEPTH> int runB(float* out, const char* what) {
    const size_t sm = 2 * XIN * 16 + 2 * YS * 4 + 400 * 4 + 16 * 2 * 16 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probeB<SKIP, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    float h[256];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((probeB<SKIP, DEPTH>), dim3(256), dim3(1024), sm, 0, out, 1.f); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    float mn = 1e9f, mx = 0.f, av = 0.f; for (int i = 0; i < 256; ++i) { mn = fminf(mn, h[i]); mx = fmaxf(mx, h[i]); av += h[i] / 256; }
    printf("B %-42s %.2f us per convolution + barrier (min %.2f, max %.2f over the blocks)\n", what, av, mn, mx);
    return 0;
}

template <int SKIP> int run(float* out, const char* what) {
    const size_t sm = 2 * XIN * 16 + 2 * YS * 4 + 400 * 4 + 16 * 2 * 16 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<SKIP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    float h[256];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe<SKIP>, dim3(256), dim3(1024), sm, 0, out, 1.f); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    float mn = 1e9f, mx = 0.f, av = 0.f; for (int i = 0; i < 256; ++i) { mn = fminf(mn, h[i]); mx = fmaxf(mx, h[i]); av += h[i] / 256; }
    printf("%-44s %.2f us per convolution + barrier (min %.2f, max %.2f over the blocks)\n", what, av, mn, mx);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 4096));
    if (run<0>(out, "as built")) return 1;
    if (run<1>(out, "no MFMAs")) return 1;
    if (run<2>(out, "no x reads")) return 1;
    if (run<4>(out, "no filter reads")) return 1;
    if (run<6>(out, "no LDS reads at all")) return 1;
    if (run<8>(out, "no epilogue")) return 1;
    if (run<14>(out, "MFMAs only")) return 1;
    if (run<15>(out, "nothing (loop + barrier)")) return 1;
    if (runB<0, 1>(out, "shared rows, depth 1")) return 1;
    if (runB<0, 2>(out, "shared rows, depth 2")) return 1;
    if (runB<0, 3>(out, "shared rows, depth 3")) return 1;
    if (runB<0, 5>(out, "shared rows, depth 5")) return 1;
    if (runB<8, 3>(out, "shared rows, depth 3, no epilogue")) return 1;
    if (runB<1, 3>(out, "shared rows, depth 3, no MFMAs")) return 1;
    return 0;
}
