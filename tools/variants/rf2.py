# what the tap loop of conv_bf16_halo_rf_kernel waits for: fragment reads off (stale registers), every tap's B from vector registers off
SRC = 'conv2d_bf16.hip'
VARIANTS = {
    'rf_noread': [("            for (int i = 0; i < TI; ++i) fa1[i] = *reinterpret_cast<const bf16x8*>(wbase + (aaddr[i][tap] ^ k64) + 2048 * i);",
                   "            for (int i = 0; i < TI; ++i) if (k64 == 0u) fa1[i] = *reinterpret_cast<const bf16x8*>(wbase + (aaddr[i][tap] ^ k64) + 2048 * i);"),
                  ("                for (int i = 0; i < TI; ++i) fa0[i] = *reinterpret_cast<const bf16x8*>(wbase + aaddr[i][tap + 1] + 2048 * i);",
                   "                for (int i = 0; i < TI; ++i) if (k64 == 0u) fa0[i] = *reinterpret_cast<const bf16x8*>(wbase + aaddr[i][tap + 1] + 2048 * i);")],
    'rf_notile0': [("        rf_mfma_drain();\n", "        rf_mfma_drain();\n        if (m0 >= 0) continue;\n")],
}
