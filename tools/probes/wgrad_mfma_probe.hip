// wgrad_mfma_probe.hip — the 4 -> 4 5x5 weight gradient of the LDS-resident backward launch on the matrix cores, in isolation: is it faster than the VALU
// units (11.2 us per layer)?  dW[co][ci][ky][kx] = sum_p x~[ci][p + kx - 2] dz~[co][p - (ky - 2) XW] over the flat zero-padded 36 x 36 tiles: rows of the
// MFMA = (kx, ci), columns = (ky, co), K = the 1296 padded positions; taps 0..3 x 0..3 in one 16 x 16 tile, the fifth row / column of taps in three more
// (39 % useful MACs).  Unit = (tile, sample[, half of K]): one accumulator, one ds_read_b32 per operand and MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/wgrad_mfma_probe.hip -o build/probes/wgrad_mfma_probe && build/probes/wgrad_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int XW = 36, TILE = 36 * 36;             // float4 per sample tile

constexpr int UN = 12;
template <int KSPLIT>
__global__ __launch_bounds__(1024) void k(const float* xsrc, const float* dsrc, float* out, int iters) {
    extern __shared__ float4 lds[];
    float4* xt = lds; float4* dzt = lds + 2 * TILE;
    for (int i = threadIdx.x; i < 2 * TILE; i += 1024) { xt[i] = reinterpret_cast<const float4*>(xsrc)[i]; dzt[i] = reinterpret_cast<const float4*>(dsrc)[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = wave & 3, s = (wave >> 2) & 1, half = wave >> 3;
    if (half >= KSPLIT) return;
    const float* xf = reinterpret_cast<const float*>(xt + s * TILE);
    const float* df = reinterpret_cast<const float*>(dzt + s * TILE);
    const int i16 = lane & 15, kq = lane >> 4;
    const int kx = (tile & 1) ? 4 : (i16 >> 2), ci = i16 & 3;          // A row (kx, ci)
    const int ky = (tile & 2) ? 4 : (i16 >> 2), co = i16 & 3;          // B column (ky, co)
    const int aoff = (kq + kx - 2) * 4 + ci, boff = (kq - (ky - 2) * XW) * 4 + co;
    constexpr int G = TILE / 4;                                         // 324 groups of 4 positions
    const int g0 = half * (G / KSPLIT), g1 = g0 + G / KSPLIT;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        int g = g0;
        auto masked = [&](int g) {
            const int ia = g * 16 + aoff, ib = g * 16 + boff;
            const bool va = ia >= 0 && ia < TILE * 4, vb = ib >= 0 && ib < TILE * 4;
            float a = xf[va ? ia : 0], b = df[vb ? ib : 0];
            a = va ? a : 0.f; b = vb ? b : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        };
        for (; g < g1 && g < 18; ++g) masked(g);
        const int gm = g1 < G - 18 ? g1 : G - 18;
        for (; g + UN <= gm; g += UN) {                                  // the operands of UN groups requested before their MFMAs
            float a[UN], b[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) { a[u] = xf[(g + u) * 16 + aoff]; b[u] = df[(g + u) * 16 + boff]; }
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
        }
        for (; g < gm; ++g) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[g * 16 + aoff], df[g * 16 + boff], acc, 0, 0, 0); }
        for (; g < g1; ++g) masked(g);
    }
    // D[row 4 kq + r][col i16]: row = (kx_r, ci_r), col = (ky, co)
    float* o = out + ((size_t)blockIdx.x * 2 + s) * KSPLIT * 400 + (size_t)half * 400;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r, rkx = (tile & 1) ? 4 : (row >> 2), rci = row & 3;
        const bool use = ((tile & 1) ? (row >> 2) == 0 : true) && ((tile & 2) ? (i16 >> 2) == 0 : true);
        if (use) o[((co * 4 + rci) * 5 + ky) * 5 + rkx] = acc[r];
    }
}

int main() {
    std::vector<float> x(2 * TILE * 4, 0.f), d(2 * TILE * 4, 0.f);
    srand(1);
    for (int s = 0; s < 2; ++s) for (int r = 2; r < 34; ++r) for (int c = 2; c < 34; ++c) for (int ch = 0; ch < 4; ++ch) {
        x[((s * TILE) + r * XW + c) * 4 + ch] = (rand() % 2001 - 1000) * 1e-3f; d[((s * TILE) + r * XW + c) * 4 + ch] = (rand() % 2001 - 1000) * 1e-3f; }
    float *dx, *dd, *dout; CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dd, d.size() * 4)); CK(hipMalloc(&dout, 256 * 2 * 2 * 400 * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, d.data(), d.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE * 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE * 16));
    // numerics (one iteration)
    for (int ks = 1; ks <= 2; ++ks) {
        CK(hipMemset(dout, 0, 256 * 2 * 2 * 400 * 4));
        if (ks == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 4 * TILE * 16, 0, dx, dd, dout, 1); else hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 4 * TILE * 16, 0, dx, dd, dout, 1);
        CK(hipDeviceSynchronize());
        std::vector<float> o(2 * ks * 400); CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int s = 0; s < 2; ++s) for (int co = 0; co < 4; ++co) for (int ci = 0; ci < 4; ++ci) for (int ky = 0; ky < 5; ++ky) for (int kx = 0; kx < 5; ++kx) {
            double ref = 0;
            for (int r = 2; r < 34; ++r) for (int c = 2; c < 34; ++c) ref += (double)d[((s * TILE) + r * XW + c) * 4 + co] * x[((s * TILE) + (r + ky - 2) * XW + c + kx - 2) * 4 + ci];
            double got = 0; for (int h = 0; h < ks; ++h) got += o[(s * ks + h) * 400 + ((co * 4 + ci) * 5 + ky) * 5 + kx];
            worst = fmax(worst, fabs(got - ref));
        }
        printf("KSPLIT %d: max abs error %.3e\n", ks, worst);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200;
    for (int ks = 1; ks <= 2; ++ks) for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (ks == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 4 * TILE * 16, 0, dx, dd, dout, iters); else hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 4 * TILE * 16, 0, dx, dd, dout, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("KSPLIT %d (%d waves): %.2f us per weight gradient of two samples (loop only)\n", ks, 8 * ks, ms * 1e3 / iters);
    }
    return 0;
}
