# round 4: phase-skip variants of conv_bf16_halo_kernel's rolling-pipeline K loop (wrong results on purpose): what does a tap wait for?
SRC = 'conv2d_bf16.hip'
NOB = ("            if (tap + 2 < 9) stage_filter((tap + 2) % 3, chunk, tap + 2);\n            else if (more) stage_filter((tap + 2) % 3, nchunk, tap - 7);\n", "")
NOW = ("            if (more && tap < npieces) stage_window_piece(PAR ^ 1, nwv, nchunk, tap);               // (the other window buffer: last read in the previous chunk)\n", "")
NOBAR = ("            __syncthreads();                                       // everybody's pieces have landed; every wave is done reading filter stage (tap + 2) % 3\n", "")
JUNK = "{ union { bf16x8 h; unsigned u[4]; } q; q.u[0] = aaddr[0][tap]; q.u[1] = bad0; q.u[2] = (unsigned)%s; q.u[3] = tap; %s = q.h; }"
NOREADS = [
    ("for (int j = 0; j < 4; ++j) fb0[j] = *reinterpret_cast<const bf16x8*>(lds + (tap % 3) * B_BYTES + j * 2048 + bad0);", "for (int j = 0; j < 4; ++j) " + JUNK % ("j", "fb0[j]")),
    ("for (int i = 0; i < TI; ++i) fa0[i] = *reinterpret_cast<const bf16x8*>(lds + PAR * BH_WSTRIDE + aaddr[i][tap]);", "for (int i = 0; i < TI; ++i) " + JUNK % ("i + 7", "fa0[i]")),
    ("for (int j = 0; j < 4; ++j) fb1[j] = *reinterpret_cast<const bf16x8*>(lds + (tap % 3) * B_BYTES + j * 2048 + bad1);", "for (int j = 0; j < 4; ++j) " + JUNK % ("j + 11", "fb1[j]")),
    ("for (int i = 0; i < TI; ++i) fa1[i] = *reinterpret_cast<const bf16x8*>(lds + PAR * BH_WSTRIDE + (aaddr[i][tap] ^ k64));", "for (int i = 0; i < TI; ++i) " + JUNK % ("i + 17", "fa1[i]")),
]
# no MFMAs: the operands stay live (an empty asm takes them), the accumulators get one add per block so the epilogue still has values
NOMFMA = [("            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);\n    };",
           "            for (int j = 0; j < 4; ++j) { asm volatile(\"\" :: \"v\"(fa[i]), \"v\"(fb[j])); }\n    };")]
VARIANTS = {
    'k_base': [],
    'k_nofilterdma': [NOB],
    'k_nodma': [NOB, NOW],
    'k_nodma_nobar': [NOB, NOW, NOBAR],
    'k_noreads': NOREADS,
    'k_mfmaonly': [NOB, NOW, NOBAR] + NOREADS,
    'k_nomfma': NOMFMA,
    'k_nomfma_nodma': NOMFMA + [NOB, NOW],
}
