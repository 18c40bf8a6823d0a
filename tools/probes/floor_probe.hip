// floor_probe.hip — what does a kernel of the stock step cost when it does (almost) nothing?  Chains of identical kernels captured into one hipGraph,
// timed with HIP events around the replay; per-kernel time = replay time / chain length.  Variants isolate grid size, LDS footprint, dirty bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/floor_probe.hip -o gpurun_out/floor_probe && gpurun_out/floor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

extern __shared__ char dyn_smem[];
struct Big { int v[48]; };   // 192 B of by-value arguments, like TileArgs + pointers

__global__ __launch_bounds__(256) void k_empty(float* p, int n) { if (n < 0) p[0] = 1.f; }
__global__ __launch_bounds__(256) void k_empty_big(Big b, float* p, int n) { if (n < 0) p[0] = (float)b.v[7]; }
__global__ __launch_bounds__(256) void k_lds(float* p, int n) { if (n < 0) { dyn_smem[threadIdx.x] = 1; __syncthreads(); p[0] = dyn_smem[3]; } }
// each block writes `per_block` floats (coalesced float4)
__global__ __launch_bounds__(256) void k_write(float* p, int per_block_v4) {
    float4* q = reinterpret_cast<float4*>(p) + (size_t)blockIdx.x * per_block_v4;
    for (int i = threadIdx.x; i < per_block_v4; i += 256) q[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void k_copy(const float* __restrict__ s, float* __restrict__ d, int per_block_v4) {
    const float4* a = reinterpret_cast<const float4*>(s) + (size_t)blockIdx.x * per_block_v4;
    float4* q = reinterpret_cast<float4*>(d) + (size_t)blockIdx.x * per_block_v4;
    for (int i = threadIdx.x; i < per_block_v4; i += 256) q[i] = a[i];
}
// tiny dependent reduction: C blocks, each sums `n` doubles (like the finalize kernels)
__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ part, int n, float* out) {
    __shared__ double red[4];
    double s = 0; for (int i = threadIdx.x; i < n; i += 256) s += part[(size_t)blockIdx.x * n + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

template <typename F>
static double chain_us(hipStream_t st, int len, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < len; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const int reps = 10;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3 / (reps * (double)len);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t MB8 = 8u << 20;
    float *a, *b; double* part; float* outp;
    CK(hipMalloc(&a, 4 * MB8)); CK(hipMalloc(&b, 4 * MB8)); CK(hipMalloc(&part, 16 * 1024 * sizeof(double))); CK(hipMalloc(&outp, 1024));
    CK(hipMemset(a, 0, 4 * MB8)); CK(hipMemset(b, 0, 4 * MB8)); CK(hipMemset(part, 0, 16 * 1024 * sizeof(double)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int L = 100;
    Big big{};
    printf("chain of %d identical kernels in one hipGraph; us per kernel\n", L);
    for (int blocks : {4, 16, 256, 512, 1024, 2048})
        printf("empty            grid %5d            : %6.2f\n", blocks, chain_us(st, L, [&](int) { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, st, a, 1); }));
    printf("empty, 192 B args grid  1024            : %6.2f\n", chain_us(st, L, [&](int) { hipLaunchKernelGGL(k_empty_big, dim3(1024), dim3(256), 0, st, big, a, 1); }));
    for (int kb : {8, 16, 32, 48, 64})
        for (int blocks : {512, 1024})
            printf("empty            grid %5d  LDS %2d KB  : %6.2f\n", blocks, kb, chain_us(st, L, [&](int) { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), kb * 1024, st, a, 1); }));
    for (int blocks : {256, 512, 1024, 2048}) {
        const int pb = (int)(MB8 / 16 / blocks);
        printf("write 8 MB       grid %5d            : %6.2f\n", blocks, chain_us(st, L, [&](int i) { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, st, (i & 1) ? a : b, pb); }));
    }
    for (int blocks : {256, 512, 1024, 2048}) {
        const int pb = (int)(MB8 / 16 / blocks);
        printf("copy 8 MB->8 MB (ping-pong, dependent) grid %5d : %6.2f\n", blocks, chain_us(st, L, [&](int i) { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, (i & 1) ? a : b, (i & 1) ? b : a, pb); }));
    }
    // the pattern of the norm chain: [copy 8 MB] -> [finalize 4 blocks over 1024 partials] -> [copy 8 MB] ...
    printf("alternating copy(1024 blocks) / finalize(4 blocks x 1024 doubles): %6.2f per PAIR\n",
           2 * chain_us(st, L, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_finalize, dim3(4), dim3(256), 0, st, part, 1024, outp);
                                           else hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, st, a, b, (int)(MB8 / 16 / 1024)); }));
    printf("finalize alone (4 blocks x 1024 doubles): %6.2f\n", chain_us(st, L, [&](int) { hipLaunchKernelGGL(k_finalize, dim3(4), dim3(256), 0, st, part, 1024, outp); }));
    printf("finalize alone (16 blocks x 512 doubles): %6.2f\n", chain_us(st, L, [&](int) { hipLaunchKernelGGL(k_finalize, dim3(16), dim3(256), 0, st, part, 512, outp); }));
    // eager (no graph) reference for the empty kernel
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, a, 1);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, a, 1);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty grid 1024, eager launches (no graph): %6.2f\n", ms);
    }
    return 0;
}
