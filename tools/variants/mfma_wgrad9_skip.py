# phase-skip variants of conv_mfma_wgrad9_kernel (wrong results on purpose)
SRC = 'conv2d_mfma.hip'
NOLOAD = ("        if (has_next) load_step(m + PS);\n        const float* gs = Gs + buf * PS * WG_LD;", "        const float* gs = Gs + buf * PS * WG_LD;")
NOSTORE = ("        if (has_next) store_step(buf ^ 1, m + PS);\n        __syncthreads();\n    }\n    // partial[chunk][tap][co][ci]", "        __syncthreads();\n    }\n    // partial[chunk][tap][co][ci]")
NOBAR = ("        if (has_next) store_step(buf ^ 1, m + PS);\n        __syncthreads();\n    }\n    // partial[chunk][tap][co][ci]", "    }\n    // partial[chunk][tap][co][ci]")
NOSLAB = ("                dst[((size_t)k * a.Cout + co) * a.Cin + ci] = acc[k][j][reg];", "                if (acc[k][j][reg] == 123.456f) dst[((size_t)k * a.Cout + co) * a.Cin + ci] = acc[k][j][reg];")
VARIANTS = {
    'g_base': [],
    'g_noload': [NOLOAD],
    'g_nostage': [NOLOAD, NOSTORE],
    'g_nostage_nobar': [NOLOAD, NOBAR],
    'g_noslab': [NOSLAB],
}
