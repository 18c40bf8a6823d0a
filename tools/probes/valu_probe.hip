// valu_probe.hip — what does the VALU of a CU deliver to a 1024-thread block (4 waves per SIMD), in the instruction forms the LDS-resident kernels use?
//   (a) v_pk_fma_f32 with a 64-bit SGPR operand pair (filter weights from the scalar cache)   (b) v_pk_fma_f32 on VGPRs only   (c) v_fma_f32 with an SGPR operand
// 8 independent accumulator pairs per thread, ITER x 80 instructions; 256 blocks.  Prints FMA / clock / CU at an assumed 2.4 GHz and the TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_probe.hip -o build/probes/valu_probe && build/probes/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(4))) const float cfloat;
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 2000;

template <int MODE>
__global__ __launch_bounds__(1024) void k(const float* w, float* out, float seed) {
    cfloat* ws = (cfloat*)(unsigned long long)w;
    f2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f2){seed + i, seed - i};
    f2 x = (f2){seed * 0.5f + threadIdx.x, seed * 0.25f};
    f2 wv[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) wv[i] = (f2){w[threadIdx.x % 64 + i], w[threadIdx.x % 32 + i]};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { const f2 s = (f2){ws[(j * 8 + i) * 2 % 64], ws[((j * 8 + i) * 2 + 1) % 64]}; acc[i] = __builtin_elementwise_fma(x, s, acc[i]); }
                else if (MODE == 1) acc[i] = __builtin_elementwise_fma(x, wv[j], acc[i]);
                else { acc[i].x = fmaf(x.x, ws[(j * 8 + i) % 64], acc[i].x); }
            }
        }
        x.x += 1e-9f;
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].y;
    if (r == 12345.678f) out[0] = r;
}

int main() {
    float *w, *out; CK(hipMalloc(&w, 4096)); CK(hipMalloc(&out, 64)); CK(hipMemset(w, 0, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, w, out, 1.f);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, w, out, 1.f);
            else hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, w, out, 1.f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double fma = (double)ITER * 80 * (mode == 2 ? 1 : 2) * 1024 * 256;
            if (rep) printf("mode %d (%s): %.1f us, %.1f TFLOP/s, %.1f FMA/clk/CU at 2.4 GHz\n", mode, mode == 0 ? "pk_fma, SGPR pair" : mode == 1 ? "pk_fma, VGPR" : "fma, SGPR",
                            ms * 1e3, 2 * fma / (ms * 1e-3) / 1e12, fma / 256 / (ms * 1e-3 * 2.4e9));
        }
    }
    return 0;
}
