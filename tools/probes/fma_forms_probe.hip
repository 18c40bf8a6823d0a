// fma_forms_probe.hip — why do the VALU convolution / weight-gradient phases of the LDS-resident kernels run at a third of the FMA rate valu_probe.hip measures?
//   mode 0: v_pk_fma_f32, natural register pairs          mode 1: v_pk_fma_f32 with a broadcast operand (op_sel_hi:[1,0,1]: what acc[co pair] += x * d[co pair] compiles to)
//   mode 2: the weight-gradient loop itself: per iteration 6 ds_read_b128 (sliding window + dz), s_waitcnt, 80 FMAs into 80 accumulators (broadcast forms)
//   mode 3: the same loop with the accumulators paired over the INPUT channel (natural pairs of x, dz duplicated by v_mov)
// NW of the block's 16 waves run the loop (the weight gradient keeps 10 busy).  Prints FMA / clock / CU at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fma_forms_probe.hip -o build/probes/fma_forms_probe && build/probes/fma_forms_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 2000;

template <int MODE>
__global__ __launch_bounds__(1024) void k(const float* w, float* out, float seed, int nw) {
    __shared__ float4 tile[2 * 1296];
    for (int i = threadIdx.x; i < 2 * 1296; i += 1024) tile[i] = make_float4(w[i % 64], w[(i + 1) % 64], seed, seed * 0.5f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= nw) return;
    float r = 0.f;
    if (MODE <= 1) {
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
                    if (MODE == 0) acc[i] = __builtin_elementwise_fma(x, wv[j], acc[i]);
                    else acc[i] = __builtin_elementwise_fma((f2){x.x, x.x}, wv[j], acc[i]);
                }
            }
            x.x += 1e-9f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += acc[i].x + acc[i].y;
    } else {
        float acc[5][4][4];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][b][c] = 0.f;
        const float4* xs = tile + wave % 5 * 36;
        const float4* ds = tile + 1296 + 2 * 36 + 2;
        for (int it = 0; it < ITER; ++it) {
            const int px = (it & 15) * 64 + lane, rr = px >> 5, c = px & 31, a = rr * 36 + c;
            const float4 d4 = ds[a];
            const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const float4 x4 = xs[a + kx];
                const float x[4] = {x4.x, x4.y, x4.z, x4.w};
                if (MODE == 2) {
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                        for (int co = 0; co < 4; ++co) acc[kx][ci][co] += x[ci] * d[co];
                } else {
                    // acc[kx][co][ci]: pairs over ci are the natural pairs (x.x, x.y), (x.z, x.w); dz duplicated into both halves
#pragma unroll
                    for (int co = 0; co < 4; ++co) {
                        const f2 dd = (f2){d[co], d[co]};
                        f2 a01 = (f2){acc[kx][co][0], acc[kx][co][1]}, a23 = (f2){acc[kx][co][2], acc[kx][co][3]};
                        a01 = __builtin_elementwise_fma((f2){x4.x, x4.y}, dd, a01);
                        a23 = __builtin_elementwise_fma((f2){x4.z, x4.w}, dd, a23);
                        acc[kx][co][0] = a01.x; acc[kx][co][1] = a01.y; acc[kx][co][2] = a23.x; acc[kx][co][3] = a23.y;
                    }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c) r += acc[a][b][c];
    }
    if (r == 12345.678f) out[0] = r;
}

int main() {
    float *w, *out; CK(hipMalloc(&w, 4096)); CK(hipMalloc(&out, 64)); CK(hipMemset(w, 0, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"pk_fma natural pairs", "pk_fma broadcast operand", "wgrad loop, pairs over co (broadcast x)", "wgrad loop, pairs over ci (dz duplicated)"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int nw : {16, 10, 4}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, w, out, 1.f, nw);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, w, out, 1.f, nw);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, w, out, 1.f, nw);
                else hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, w, out, 1.f, nw);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double fma = (double)ITER * (mode <= 1 ? 160 : 80) * 64 * nw * 256;
                if (rep) printf("mode %d (%s), %2d waves: %8.1f us, %6.1f FMA/clk/CU, %.0f clk per wave-iteration\n", mode, names[mode], nw, ms * 1e3,
                                fma / 256 / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / ITER);
            }
        }
    }
    return 0;
}
