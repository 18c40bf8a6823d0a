# conv_mfma_wgrad9_kernel with LDS-DMA staging: is the cost of the staging loads their traffic (L2 / HBM -> CU) or the mechanism (instructions, LDS writes)?  (wrong results on purpose)
SRC = 'conv2d_mfma.hip'
SAME = ("        dp0 += PS; if (dp0 >= a.HWp) { dp0 = 0; ++dn; }\n", "")
NOX = ("            if (4 * k + wave < nix) __builtin_amdgcn_raw_ptr_buffer_load_lds(rxi,", "            if (4 * k + wave < nix && k > 100) __builtin_amdgcn_raw_ptr_buffer_load_lds(rxi,")
NOG = ("            if (4 * k + wave < nig) __builtin_amdgcn_raw_ptr_buffer_load_lds(rgi,", "            if (4 * k + wave < nig && k > 100) __builtin_amdgcn_raw_ptr_buffer_load_lds(rgi,")
VARIANTS = {
    'd_base': [],
    'd_same': [SAME],
    'd_nox': [NOX],
    'd_nodma': [NOX, NOG],
}
