// conv2d_mfma.hip — MFMA implicit-GEMM convolution back-end for wide layers (placeholder dispatcher until the kernels land).
#include "conv2d_internal.h"

bool dcv_mfma_conv2d_supported(const dcv_conv2d_desc*, int) { return false; }
size_t dcv_mfma_conv2d_ws(const dcv_conv2d_desc*, int) { return 0; }
int dcv_mfma_conv2d_fwd(const dcv_conv2d_desc*, const float*, const float*, const float*, float*, void*, size_t, hipStream_t) { return DCV_ERR_UNSUPPORTED; }
int dcv_mfma_conv2d_dgrad(const dcv_conv2d_desc*, const float*, const float*, float*, void*, size_t, hipStream_t) { return DCV_ERR_UNSUPPORTED; }
int dcv_mfma_conv2d_wgrad(const dcv_conv2d_desc*, const float*, const float*, float*, void*, size_t, hipStream_t) { return DCV_ERR_UNSUPPORTED; }
