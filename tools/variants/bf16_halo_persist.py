# the 128-channel-wide instance of conv_bf16_halo_kernel persistent over tiles like the 64-wide one (dropped earlier on 256 -> 256 @8x8: registers); re-tried
# on the 16 x 16 and 8 x 8 shapes of the bench
SRC = 'conv2d_bf16.hip'
VARIANTS = {
    'halopersist2': [
        ("template <int WN, bool OUT_BF16, bool PERSIST = (WN == 1)>", "template <int WN, bool OUT_BF16, bool PERSIST = true>"),
        ("            unsigned hgx = wide ? htiles : DCV_NUM_CU / hgy; if (hgx < 1) hgx = 1;", "            unsigned hgx = DCV_NUM_CU / hgy; if (hgx < 1) hgx = 1;"),
    ],
}
