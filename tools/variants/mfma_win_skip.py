# phase-skip variants of conv_mfma_win_kernel's main loop (wrong results on purpose): what does a solo wave wait for?
SRC = 'conv2d_mfma.hip'
NOW = ("            if (tap < 8) wload(wn, chunk, tap + 1);                // requested a whole tap (64 MFMAs) before their first use\n            else if (chunk + 1 < cchunks) wload(wn, chunk + 1, 0);\n",
       "            if (tap == 100) wload(wn, chunk, tap + 1);\n")
NOX = ("            if (tap < 8) xload(xn, tap + 1);                       // (the next chunk's window is only complete behind the barrier)\n", "            if (tap == 100) xload(xn, tap + 1);\n")
NOBAR = ("        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");           // this wave's pieces of the next window have landed ...\n        __syncthreads();                                           // ... and everybody's; every wave is done reading this one\n", "")
NODMA = ("            if (tap == 0 && chunk + 1 < cchunks) stage((chunk + 1) & 1, chunk + 1);\n", "")
VARIANTS = {
    'w_base': [],
    'w_nofilter': [NOW],
    'w_nox': [NOX],
    'w_nobar': [NOBAR],
    'w_nodma': [NODMA],
    'w_mfmaonly': [NOW, NOX, NOBAR, NODMA],
}
