// mfma4x4_probe.hip — v_mfma_f32_4x4x1_16b_f32 (16 independent 4 x 4 outer products per wave instruction, K = 1) as the engine of small-channel convolutions:
//   (1) operand layout: A lane l -> block l / 4, row l % 4;  B lane l -> block l / 4, column l % 4;  D register r, lane l -> block l / 4, row r, column l % 4 ?
//   (2) rate: a 1024-thread block (4 waves per SIMD) issuing MFMAs back to back on 2 alternating accumulator tiles, with and without the LDS reads a
//       convolution pairs with them (one ds_read_b128 per 4 MFMAs + one per 4 MFMAs for the weights)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma4x4_probe.hip -o build/probes/mfma4x4_probe && build/probes/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout(const float* a, const float* b, float* d) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[r * 64 + threadIdx.x] = acc[r];
}

constexpr int ITER = 4000;
template <int MODE>
__global__ __launch_bounds__(1024) void rate(float* out, float seed) {
    __shared__ f4 tile[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) tile[i] = (f4){seed + i, seed, seed * 2, 1.f};
    __syncthreads();
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    const f4* p = tile + (threadIdx.x & 63);
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, acc1, 0, 0, 0);
            }
        } else {
            const f4 x = p[(it & 31) * 64], w = p[2048 + (it & 31) * 64];          // one x vector and one weight vector per 4 + 4 MFMAs
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, w.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, w.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, w.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, w.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, w.w, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, w.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, w.x, acc1, 0, 0, 0);
        }
    }
    const float r = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
    if (r == 12345.678f) out[0] = r;
}

int main() {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + 3 * l; }
    float *a, *b, *d, *out; CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&d, 1024)); CK(hipMalloc(&out, 64));
    CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, a, b, d);
    CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) { const int blk = l / 4, j = l % 4; if (hd[r * 64 + l] != ha[blk * 4 + r] * hb[blk * 4 + j]) ++bad; }
    printf("layout D[reg r][lane l] = A[4 (l / 4) + r] * B[l]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, hd[l], hd[64 + l], hd[128 + l], hd[192 + l]);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(1024), 0, 0, out, 1.f); else hipLaunchKernelGGL(rate<1>, dim3(256), dim3(1024), 0, 0, out, 1.f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double mfma_per_simd = (double)ITER * 8 * 4;      // 4 waves per SIMD
            if (rep) printf("mode %d (%s): %.1f us, %.2f clk per MFMA per SIMD at 2.4 GHz (%.2f at 2.1), %.1f MAC/clk/CU at 2.4 GHz\n", mode, mode ? "with 2 ds_read_b128 per 8" : "bare",
                            ms * 1e3, ms * 1e-3 * 2.4e9 / mfma_per_simd, ms * 1e-3 * 2.1e9 / mfma_per_simd, mfma_per_simd * 4 * 256 / (ms * 1e-3 * 2.4e9));
        }
    return 0;
}
