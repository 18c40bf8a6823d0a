# A/B: the eight-wave window kernel's MOM epilogue with its yprev / addend rows requested into L2 at the start of the tile (as conv_bf16_halo_rf_kernel does)
SRC = 'conv2d_bf16.hip'
_PF = '''    unsigned pfy = 0u, pfa = 0u;
    auto prefetch_rows = [&](int m0v) __attribute__((always_inline)) {
        if constexpr (MOM) {
            const int row = t / WN, mp = m0v + row < (int)a.M ? m0v + row : m0v;
            if (yprev != nullptr) { const u16* q = yprev + (size_t)mp * a.Cout + n0 + 64 * (t % WN); asm volatile("global_load_dword %0, %1, off" : "=v"(pfy) : "v"(q)); }
            if (addend != nullptr) { const u16* q = addend + (size_t)mp * a.Cout + n0 + 64 * (t % WN); asm volatile("global_load_dword %0, %1, off" : "=v"(pfa) : "v"(q)); }
        }
    };
'''
_A = "    auto finish_tile = [&](int dead) __attribute__((always_inline)) {\n        float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};\n        uint2 yq[MOM ? TI : 1][4], aq[MOM ? TI : 1][4];"
_B = "    unsigned wv_cur = wv_of(m0), wv_nxt = (PERSIST && tile + gstep < ntiles) ? wv_of(m0 + gstep * BMH) : 0u;\n    for (int p = 0; p < npieces; ++p) stage_window_piece(0, wv_cur, 0, p);"
_C = "                if (more) { set_tile(m0); wv_cur = wv_nxt; wv_nxt = tile + gstep < ntiles ? wv_of(m0 + gstep * BMH) : 0u; fresh = true; }"
_D = "    if (pend) mma_set(fa1, fb1);\n    if constexpr (!PERSIST) finish_tile((cchunks - 1) & 1);         // (one tile: its epilogue once, behind the loop)\n}"
VARIANTS = {
    'halo_base': [],
    'halo_pf': [(_A, _PF + _A),
                (_B, _B.replace("    for (int p = 0;", "    prefetch_rows(m0);\n    for (int p = 0;")),
                (_C, _C.replace("set_tile(m0); ", "set_tile(m0); prefetch_rows(m0); ")),
                (_D, _D.replace("\n}", "\n    if constexpr (MOM) asm volatile(\"\" :: \"v\"(pfy), \"v\"(pfa));\n}"))],
}
