# phase-skip variants of the 32x32x2 weight-gradient loop (wrong results on purpose)
SRC = 'conv2d_mfma.hip'
NOX = ("            if (4 * k + wave < nix) __builtin_amdgcn_raw_ptr_buffer_load_lds(rxi,", "            if (4 * k + wave < nix && k > 100) __builtin_amdgcn_raw_ptr_buffer_load_lds(rxi,")
NOG = ("            if (4 * k + wave < nig) __builtin_amdgcn_raw_ptr_buffer_load_lds(rgi,", "            if (4 * k + wave < nig && k > 100) __builtin_amdgcn_raw_ptr_buffer_load_lds(rgi,")
NOBAR = ("        else asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");     // this wave's pieces of the next step have landed\n        __syncthreads();\n", "")
NOB = ("                            bq[r * 3 + sx][0] = xp[0]; bq[r * 3 + sx][1] = xp[WG_LD];", "                            bq[r * 3 + sx][0] = (float)row; bq[r * 3 + sx][1] = (float)(r + sx); (void)xp;")
NOA = ("                    const float av[2] = {gp[0], gp[WG_LD]};                                       // A'[i = co][k = pixel]\n                    const float4 mf = *reinterpret_cast<const float4*>(cf + row * 2);            // (row is even)",
       "                    const float av[2] = {(float)row, 1.f}; (void)gp;\n                    const float4 mf = make_float4(1.f, 0.f, 1.f, 1.f);")
VARIANTS = {
    'h_base': [],
    'h_nodma': [NOX, NOG],
    'h_nodma_nobar': [NOX, NOG, NOBAR],
    'h_noreads': [NOA, NOB],
    'h_mfmaonly': [NOX, NOG, NOBAR, NOA, NOB],
}
