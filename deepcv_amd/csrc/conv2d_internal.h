// conv2d_internal.h — internal (non-exported) entry points of the convolution back-ends, dispatched by conv2d.hip.
#pragma once
#include "common.h"

// direct (register-blocked, LDS-resident filter) back-end: any shape, tuned for the small-channel stock layers
size_t dcv_direct_conv2d_fwd_ws(const dcv_conv2d_desc* d);
size_t dcv_direct_conv2d_dgrad_ws(const dcv_conv2d_desc* d);
int dcv_direct_conv2d_fwd(const dcv_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, hipStream_t st);
int dcv_direct_conv2d_dgrad(const dcv_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes, hipStream_t st);
size_t dcv_direct_conv2d_wgrad_ws(const dcv_conv2d_desc* d);
bool dcv_direct_conv2d_wgrad_fuses_bias(const dcv_conv2d_desc* d);
int dcv_direct_conv2d_wgrad(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, float* db, bool* db_done, void* ws, size_t ws_bytes, hipStream_t st);

// MFMA implicit-GEMM back-end (fp32 inputs, v_mfma_f32_16x16x4_f32): wide layers (conv2d_mfma.hip)
bool dcv_mfma_conv2d_supported(const dcv_conv2d_desc* d, int op /*0 fwd, 1 dgrad, 2 wgrad*/);
size_t dcv_mfma_conv2d_ws(const dcv_conv2d_desc* d, int op);
int dcv_mfma_conv2d_fwd(const dcv_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, hipStream_t st);
int dcv_mfma_conv2d_dgrad(const dcv_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes, hipStream_t st);
int dcv_mfma_conv2d_wgrad(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st);

// shared with elementwise.hip
extern "C" size_t dcv_act_bwd_workspace_bytes(size_t rows, int channels);
extern "C" int dcv_channel_sum(const float* x, float* out, size_t rows, int channels, void* ws, size_t ws_bytes, void* stream);
