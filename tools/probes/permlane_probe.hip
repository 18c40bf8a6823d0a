// what v_permlane32_swap / v_permlane16_swap and the row rotations of the DPP really do on gfx950 (rb_sum_blocks24, resident.hip, was written from the
// LLVM documentation and summed the wrong lanes): prints, per lane, both results of swap(a = lane, b = 100 + lane) and row_ror:8 / row_ror:4 of the lane id
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
    const int l = threadIdx.x;
    const auto r32 = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(100 + l), false, false);
    const auto r16 = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)(100 + l), false, false);
    o[l * 6 + 0] = (int)r32[0]; o[l * 6 + 1] = (int)r32[1]; o[l * 6 + 2] = (int)r16[0]; o[l * 6 + 3] = (int)r16[1];
    o[l * 6 + 4] = __builtin_amdgcn_update_dpp(0, l, 0x128, 0xf, 0xf, false);
    o[l * 6 + 5] = __builtin_amdgcn_update_dpp(0, l, 0x124, 0xf, 0xf, false);
}
int main() {
    int* d; hipMalloc(&d, 64 * 6 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[64 * 6]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: swap32 -> (%3d, %3d)  swap16 -> (%3d, %3d)  ror8 %2d  ror4 %2d\n", l, h[l * 6], h[l * 6 + 1], h[l * 6 + 2], h[l * 6 + 3], h[l * 6 + 4], h[l * 6 + 5]);
    return 0;
}
