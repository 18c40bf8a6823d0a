# phase-skip variants of conv_bf16_halo_kernel's K loop (wrong results on purpose): what does a tap wait for?
SRC = 'conv2d_bf16.hip'
NOB = ("            if (tap + 2 < 9) stage_filter((tap + 2) % 3, chunk, tap + 2);\n            else if (more) stage_filter((tap + 2) % 3, nchunk, tap - 7);\n", "")
NOW = ("            if (more && tap < npieces) stage_window_piece(PAR ^ 1, nwv, nchunk, tap);               // (the other window buffer: last read in the previous chunk)\n", "")
NOBAR = ("            __syncthreads();                                       // everybody's pieces have landed; every wave is done reading filter stage (tap + 2) % 3\n", "")
NOREADS = ("                for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(lds + PAR * BH_WSTRIDE + (ks ? (aaddr[i][tap] ^ 64u) : aaddr[i][tap]));\n#pragma unroll\n                for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(lds + (tap % 3) * B_BYTES + j * 2048 + (ks ? bad1 : bad0));",
           "                for (int i = 0; i < TI; ++i) { union { bf16x8 h; unsigned u[4]; } q; q.u[0] = aaddr[i][tap]; q.u[1] = bad0; q.u[2] = ks; q.u[3] = tap; af[i] = q.h; }\n#pragma unroll\n                for (int j = 0; j < 4; ++j) { union { bf16x8 h; unsigned u[4]; } q; q.u[0] = bad1 + j; q.u[1] = bad0; q.u[2] = ks; q.u[3] = tap; bf[j] = q.h; }")
VARIANTS = {
    'k_base': [],
    'k_nofilterdma': [NOB],
    'k_nodma': [NOB, NOW],
    'k_nodma_nobar': [NOB, NOW, NOBAR],
    'k_noreads': [NOREADS],
    'k_mfmaonly': [NOB, NOW, NOBAR, NOREADS],
}
