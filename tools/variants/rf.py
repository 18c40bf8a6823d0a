# phase-skip variants of the filters-in-registers window kernel (conv_bf16_halo_rf_kernel): what each part of a tile costs
SRC = 'conv2d_bf16.hip'
VARIANTS = {
    'rf_base': [],
    'rf_noepi': [("        else finish_tile(par, std::integral_constant<int, -1>{}, std::false_type{});\n    }\n}", "        else if (m0 < 0) finish_tile(par, std::integral_constant<int, -1>{}, std::false_type{});\n    }\n}"),
                 ("        if (m0 + BMH <= (int)a.M && a.act == DCV_ACT_NONE) finish_tile(", "        if (m0 < -1 && a.act == DCV_ACT_NONE) finish_tile("),
                 ("        else if (m0 + BMH <= (int)a.M && a.act == DCV_ACT_RELU) finish_tile(", "        else if (m0 < -1 && a.act == DCV_ACT_RELU) finish_tile(")],
    'rf_nostats': [("        if (stats != nullptr) {\n            // statistics rows per 128 pixels: the tile's halves = waves 0, 1 and waves 2, 3 (same butterfly", "        if (stats != nullptr && m0 < 0) {\n            // statistics rows per 128 pixels: the tile's halves = waves 0, 1 and waves 2, 3 (same butterfly")],
    'rf_noset': [("        if (!first) set_tile(m0);\n", "")],
    'rf_onetap': [("        for (int tap = 0; tap < 9; ++tap) {\n#pragma unroll\n            for (int i = 0; i < TI; ++i) fa1[i] = *reinterpret_cast<const bf16x8*>(wbase",
                   "        for (int tap = 0; tap < 1; ++tap) {\n#pragma unroll\n            for (int i = 0; i < TI; ++i) fa1[i] = *reinterpret_cast<const bf16x8*>(wbase")],
    'rf_nodma': [("        if (tile + gstep < ntiles) stage_window(par ^ 1, wv_of(m0 + gstep * BMH));\n", "")],
}
