# prefetch depth of the 4x4x1 convolution's tap loop (rs_conv_c4_co4_32): 2 slots ahead in the product; variants 5 and 10
SRC = 'resident.hip'
VARIANTS = {
    'convdepth5': [("    constexpr int XW = 32 + 2 * (K / 2), NS = (K + 1) * K, DEPTH = 2;", "    constexpr int XW = 32 + 2 * (K / 2), NS = (K + 1) * K, DEPTH = 5;")],
    'convdepth10': [("    constexpr int XW = 32 + 2 * (K / 2), NS = (K + 1) * K, DEPTH = 2;", "    constexpr int XW = 32 + 2 * (K / 2), NS = (K + 1) * K, DEPTH = 10;")],
}
