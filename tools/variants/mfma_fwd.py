# phase-skip variants of conv_mfma_kernel (fp32 implicit GEMM): what does each part of the loop / epilogue cost on 64 -> 64 @32x32 N = 512?
SRC = 'conv2d_mfma.hip'
SEL = ("v[j] = aokf[j] != 0.f ? g : make_float4(0.f, 0.f, 0.f, 0.f);", "v[j] = g;")
TAP = ("if (lr < a.R) set_tap(lr, ls);", "{ _Pragma(\"unroll\") for (int j = 0; j < BMR; ++j) aptr[j] -= a.Cin; }")
STO = ("                    y[off * a.Cout + co] = o;\n", "                    if (o == 123.456f) y[off * a.Cout + co] = o;\n")
VARIANTS = {
    'base': [],
    'nosel': [SEL],
    'notap': [SEL, TAP],
    'nostore': [STO],
    'notap_nostore': [SEL, TAP, STO],
}
