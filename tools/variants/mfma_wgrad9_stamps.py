# time stamps (wall_clock64, 100 MHz) of every block of conv_mfma_wgrad9_kernel<32, false, false, false> (64 -> 64: one block per chunk): written over the first floats
# of the block's slab (results wrong on purpose): [entry & 0xFFFFFF, prologue, loop, epilogue] in ticks; tools/kbench.py wgrad6430 with DCV_KBENCH_WSTAMPS=<blocks>
SRC = 'conv2d_mfma.hip'
VARIANTS = {
    'gstamps': [
        ("    extern __shared__ __attribute__((aligned(16))) float smem9[];\n    const int XR = a.XR;", "    extern __shared__ __attribute__((aligned(16))) float smem9[];\n    const long long T0 = wall_clock64();\n    const int XR = a.XR;"),
        ("    __syncthreads();\n    int it = 0;\n    for (long m = mb; m < me; m += PS, ++it) {", "    __syncthreads();\n    const long long T1 = wall_clock64(), C1 = (long long)__builtin_readcyclecounter();\n    int it = 0;\n    for (long m = mb; m < me; m += PS, ++it) {"),
        ("    float* dst = partial + (size_t)blockIdx.x * a.wtotal;\n    if constexpr (W32) {", "    float* dst = partial + (size_t)blockIdx.x * a.wtotal;\n    const long long T2 = wall_clock64(), C2 = (long long)__builtin_readcyclecounter();\n    if constexpr (W32) {"),
        ("                dst[((size_t)k * a.Cout + co) * a.Cin + ci] = acc32[k][r];\n            }\n        return;", "                dst[((size_t)k * a.Cout + co) * a.Cin + ci] = acc32[k][r];\n            }\n        __builtin_amdgcn_s_waitcnt(0);\n        __syncthreads();\n        if (t == 0) { const long long T3 = wall_clock64(); dst[0] = (float)(T0 & 0xFFFFFF); dst[1] = (float)(T1 - T0); dst[2] = (float)(T2 - T1); dst[3] = (float)(T3 - T2); dst[4] = (float)(C2 - C1) / (float)(T2 - T1); }\n        return;"),
    ],
}
