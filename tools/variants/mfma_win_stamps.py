# time stamps (wall_clock64: 100 MHz) of every block of conv_mfma_win_kernel: entry, main loop start, main loop end, exit -> the stats buffer (tools/kbench.py <case>p with DCV_KBENCH_STAMPS=<blocks>)
SRC = 'conv2d_mfma.hip'
VARIANTS = {
    'winstamps': [
        ("    extern __shared__ __attribute__((aligned(16))) char mw_lds[];\n", "    extern __shared__ __attribute__((aligned(16))) char mw_lds[];\n    const long long T0 = wall_clock64();\n"),
        ("    f32x4 wa[4], wb2[4];", "    const long long Ta = wall_clock64();\n    f32x4 wa[4], wb2[4];"),
        ("    wload(wa, 0, 0);\n    MF_WAIT_VMCNT0();\n    __syncthreads();", "    wload(wa, 0, 0);\n    const long long Tb = wall_clock64();\n    MF_WAIT_VMCNT0();\n    const long long Tc = wall_clock64();\n    __syncthreads();"),
        ("    int chunk = 0;\n    for (; chunk + 1 < cchunks; chunk += 2)", "    const long long T1 = wall_clock64(), C1 = (long long)__builtin_readcyclecounter();\n    int chunk = 0;\n    for (; chunk + 1 < cchunks; chunk += 2)"),
        ("    float ssum[4][4], ssq[4][4];\n    float4 bv[4];\n", "    const long long T2 = wall_clock64(), C2 = (long long)__builtin_readcyclecounter();\n    float ssum[4][4], ssq[4][4];\n    float4 bv[4];\n"),
        ("    if (stats != nullptr) {\n        // BatchNorm statistics partials of this 256-pixel tile",
         "    if (stats != nullptr && t == 0) { __builtin_amdgcn_s_waitcnt(0); const long long T3 = wall_clock64(); double* o = stats + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8; o[0] = (double)T0; o[1] = (double)T1; o[2] = (double)T2; o[3] = (double)T3; o[4] = (double)Ta; o[5] = (double)Tb; o[6] = (double)Tc; o[7] = (double)(C2 - C1) / (double)(T2 - T1); }\n    if (false) {\n        // BatchNorm statistics partials of this 256-pixel tile"),
    ],
}
