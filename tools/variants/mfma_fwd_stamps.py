# time stamps (wall_clock64: 100 MHz) of every block of conv_mfma_kernel: entry, main loop start, main loop end, exit -> the stats buffer (tools/kbench.py <case>p with DCV_KBENCH_STAMPS=1)
SRC = 'conv2d_mfma.hip'
VARIANTS = {
    'stamps': [
        ("    constexpr int NT = BN / 16;                                   // accumulator tiles along N per wave\n    constexpr int BM = 64 * BMR;\n    __shared__ __attribute__((aligned(16))) float As[2][BM * MF_LD];",
         "    constexpr int NT = BN / 16;\n    constexpr int BM = 64 * BMR;\n    const long long T0 = wall_clock64();\n    __shared__ __attribute__((aligned(16))) float As[2][BM * MF_LD];"),
        ("    const int li = lane & 15, lg = lane >> 4;\n    for (int c = 0; c < nchunks; ++c) {\n        const int buf = c & 1;\n        const bool has_next = (c + 1) < nchunks;",
         "    const int li = lane & 15, lg = lane >> 4;\n    const long long T1 = wall_clock64();\n    for (int c = 0; c < nchunks; ++c) {\n        const int buf = c & 1;\n        const bool has_next = (c + 1) < nchunks;"),
        ("    // ---- epilogue: acc[i][j][reg] -> pixel m0 + wave*16*BMR + i*16 + 4*lg + reg, channel n0 + j*16 + li\n    float ssum[NT], ssq[NT];",
         "    const long long T2 = wall_clock64();\n    float ssum[NT], ssq[NT];"),
        ("    if (stats != nullptr) {\n        // BatchNorm statistics partials of this 128-pixel tile",
         "    if (stats != nullptr && t == 0) { __builtin_amdgcn_s_waitcnt(0); const long long T3 = wall_clock64(); double* o = stats + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4; o[0] = (double)T0; o[1] = (double)T1; o[2] = (double)T2; o[3] = (double)T3; }\n    if (false) {\n        // BatchNorm statistics partials of this 128-pixel tile"),
    ],
}
