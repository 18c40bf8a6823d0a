/* deepcv_hip.h — C-ABI of libdeepcv_hip.so: hand-written HIP kernels (gfx950 / MI355X) for DeepCV's
 * image-classifier training hot path.
 *
 * The reference (PaulEmmanuelSotir/DeepCV) is 100 % Python and has NO native layer; every
 * FLOP of its hot path is dispatched to third-party PyTorch ATen ops through `torch.nn` modules.  Each entry
 * point below therefore cites the reference line that *triggers* the ATen op it replaces (SURVEY.md §2.3
 * rows K1..K12).  How a reference maintainer would bind these (ctypes stubs behind the submodule-creator
 * registry) is shown in INTEGRATION.md; `deepcv_amd/_native.py` is that binding.
 *
 * Conventions
 *   - plain device pointers + explicit dims; no torch types.  `stream` is a `hipStream_t` passed as `void*`
 *     (0 = the null stream).  Launchers never allocate, never synchronise, never throw: they validate the
 *     arguments on the host, enqueue kernels on `stream` and return 0, or a negative `DCV_ERR_*` code
 *     (`dcv_strerror`).  They are re-entrant and safe to capture into a hipGraph.
 *   - activations are fp32 **NHWC** (channels-last): element (n, h, w, c) at ((n*H + h)*W + w)*C + c.
 *   - parameters keep the reference's `state_dict` layouts: conv weight [Cout][Cin/groups][R][S] (OIHW),
 *     linear weight [out][in], per-channel vectors [C].
 *   - reductions are fixed-order (per-block partial slabs + ordered second stage): every kernel is
 *     bit-reproducible run to run; no floating-point atomics anywhere.
 *   - `ws`/`ws_bytes`: caller-provided scratch; query the size with the matching `*_workspace_bytes`.
 */
#ifndef DEEPCV_HIP_H
#define DEEPCV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCV_OK 0
#define DCV_ERR_INVALID_ARG (-1)   /* NULL pointer, non-positive dim, inconsistent shapes */
#define DCV_ERR_UNSUPPORTED (-2)   /* valid request this build has no kernel for (e.g. groups > 1) */
#define DCV_ERR_WORKSPACE (-3)     /* ws_bytes smaller than *_workspace_bytes() */
#define DCV_ERR_LAUNCH (-4)        /* hipGetLastError() != hipSuccess after enqueue */

/* fused epilogue / elementwise activation codes (reference: `act_fn()` in nn.py:553) */
#define DCV_ACT_NONE 0
#define DCV_ACT_RELU 1
#define DCV_ACT_LEAKY_RELU 2
#define DCV_ACT_SIGMOID 3
#define DCV_ACT_TANH 4

#define DCV_POOL_AVG 0
#define DCV_POOL_MAX 1

#define DCV_LAYOUT_NHWC 0
#define DCV_LAYOUT_NCHW 1

const char* dcv_strerror(int code);
int dcv_version(void);

/* Launch counters of the convolution kernel families (host-side, per process; for tests and diagnostics): which hand-written kernel a layer runs on is a
 * plan made inside the library (shape, pixel count, alignment), so a parity test at the bench sizes reads these around its step to assert that the family
 * the bench times is the one it compared.  counts[f] = launches of family f since the library was loaded; fills min(n, DCV_KERNEL_FAMILIES) entries and
 * returns DCV_KERNEL_FAMILIES.  No reference counterpart (the reference dispatches through ATen). */
enum {
    DCV_FAMILY_BF16_IGEMM = 0,      /* conv_bf16_kernel: bf16 implicit GEMM (strided layers, small maps, 1x1) */
    DCV_FAMILY_BF16_WINDOW = 1,     /* conv_bf16_halo_kernel: bf16 3x3 window kernel */
    DCV_FAMILY_BF16_WGRAD9 = 2,     /* conv_bf16_wgrad9_kernel: bf16 all-taps weight gradient (and its one-tap GEMM instance) */
    DCV_FAMILY_BF16_WGRAD_TAP = 3,  /* dcv_conv2d_bf16_wgrad's fallback: the fp32-staging weight gradient kernels reading bf16 tensors (layers wgrad9 does not plan) */
    DCV_FAMILY_F32_IGEMM = 4,       /* conv_mfma_kernel: fp32 implicit GEMM */
    DCV_FAMILY_F32_HALO = 5,        /* conv_mfma_halo_kernel: narrow (16 / 32 channel) fp32 layers */
    DCV_FAMILY_F32_WINDOW = 6,      /* conv_mfma_win_kernel: fp32 3x3 window kernel */
    DCV_FAMILY_F32_WGRAD9 = 7,      /* conv_mfma_wgrad9_kernel: fp32 all-taps weight gradient */
    DCV_FAMILY_F32_WGRAD_TAP = 8,   /* conv_mfma_wgrad_kernel<false>: fp32 per-tap weight gradient */
    DCV_KERNEL_FAMILIES = 9
};
int dcv_launch_counts(unsigned long long* counts, int n);
/* Fills multiprocessor (CU) count, wavefront size and the gcn arch name of HIP device `device`. */
int dcv_device_info(int device, int* cu_count, int* wave_size, char* arch, int arch_len);

/* ---------------------------------------------------------------------------------------------------
 * K1-K3  conv2d forward / data-gradient / weight-gradient
 * replaces torch.nn.Conv2d created at submodule_creators.py:251 (fwd) and its autograd (ignite_training.py:253)
 * ------------------------------------------------------------------------------------------------- */
#define DCV_MATH_FP32 0
#define DCV_MATH_BF16 1
typedef struct {
    int n, h, w, cin;            /* input  N x H x W x Cin                                     */
    int cout, r, s;              /* filter Cout x Cin x R x S (OIHW)                           */
    int stride_h, stride_w;
    int pad_h, pad_w;
    int dil_h, dil_w;
    int groups;                  /* only 1 is supported                                        */
    int ho, wo;                  /* output spatial dims; must equal the conv formula           */
    int x_layout;                /* DCV_LAYOUT_NHWC, or DCV_LAYOUT_NCHW for the forward input  */
    int act;                     /* forward: fused activation applied after bias (DCV_ACT_*)   */
    float act_slope;             /* negative slope for DCV_ACT_LEAKY_RELU                      */
    int math;                    /* DCV_MATH_FP32 (default), or DCV_MATH_BF16: the MFMA kernels round x / dy / w to bf16 on the fly and accumulate in fp32
                                    (BASELINE configs 3 and 5; tensors stay fp32 in memory; layers not served by an MFMA kernel ignore it) */
} dcv_conv2d_desc;

size_t dcv_conv2d_fwd_workspace_bytes(const dcv_conv2d_desc* d);
size_t dcv_conv2d_dgrad_workspace_bytes(const dcv_conv2d_desc* d);
size_t dcv_conv2d_wgrad_workspace_bytes(const dcv_conv2d_desc* d);
/* y[N,Ho,Wo,Cout] = act(conv(x, w) + bias);  bias may be NULL */
int dcv_conv2d_fwd(const dcv_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                   void* ws, size_t ws_bytes, void* stream);
/* dx[N,H,W,Cin] = conv_transpose(dy[N,Ho,Wo,Cout], w)   (dy is the gradient w.r.t. the PRE-activation output) */
int dcv_conv2d_dgrad(const dcv_conv2d_desc* d, const float* dy, const float* w, float* dx,
                     void* ws, size_t ws_bytes, void* stream);
/* dw[Cout,Cin,R,S] = x (*) dy ; db[Cout] = sum dy (db may be NULL).  x honours d->x_layout. */
int dcv_conv2d_wgrad(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, float* db,
                     void* ws, size_t ws_bytes, void* stream);
/* Deferred second stage.  Weight gradients are produced in two stages: per-block partial slabs, then an ordered sum of the slabs.  On small layers the second
 * stage is a ~5 us launch whose result nobody reads before the optimizer (or the gradient all-reduce), so a training step may postpone it and sum the slabs
 * of ALL its layers in ONE launch: the *_deferred variants run stage one only and describe the pending sums as up to two segments (weights, bias);
 * dcv_slab_reduce executes any number of segments in one launch (dst[i] = sum over k < chunks of src[k * stride + i], fixed order: the arithmetic of
 * the immediate entry points).  `*n_segments == 0` on return: the back-end summed immediately (nothing is pending).  Until the segments have been
 * reduced the workspace `ws` must stay untouched. */
typedef struct dcv_slab_segment {
    const float* src;
    float* dst;
    unsigned long long count;      /* entries of dst */
    unsigned long long stride;     /* entries between consecutive slabs of src */
    int chunks;                    /* slabs */
    int tr_a, tr_b;                /* 0, 0: dst[i] = sum.  Else the sums are [..][tr_a][tr_b] matrices written transposed: element ((o * tr_a + p) * tr_b + c) goes to
                                    * dst[(o * tr_b + c) * tr_a + p] (weight gradient of a Linear fed with NHWC-flattened activations, see dcv_linear_bwd_deferred) */
    int reserved;
} dcv_slab_segment;
#define DCV_MAX_SLAB_SEGMENTS 32   /* per launch; dcv_slab_reduce splits longer lists */
int dcv_conv2d_wgrad_deferred(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, float* db,
                              void* ws, size_t ws_bytes, dcv_slab_segment* segments /* [2] */, int* n_segments, void* stream);
int dcv_slab_reduce(int n, const dcv_slab_segment* segments, void* stream);
/* Stage one of SEVERAL weight gradients (the layers of one backward pass, collected by the caller) in as few launches as possible: the stock layers'
 * kernels run as block ranges of one multi-problem launch.  Appends every pending sum to `segments` (capacity 2 * count, *n_segments is the running
 * count); gradients of jobs whose back-end sums immediately are complete on return.  Each job needs its own workspace. */
typedef struct dcv_wgrad_job {
    dcv_conv2d_desc desc;
    const float* x;
    const float* dy;
    float* dw;
    float* db;                     /* nullable */
    void* ws;
    size_t ws_bytes;               /* >= dcv_conv2d_wgrad_workspace_bytes(&desc) */
} dcv_wgrad_job;
int dcv_conv2d_wgrad_batch(int count, const dcv_wgrad_job* jobs, dcv_slab_segment* segments, int* n_segments, void* stream);
/* 1 when dcv_conv2d_wgrad produces db from the dy tiles it already stages (no extra pass over dy, no extra launch): callers which can get the bias
 * gradient elsewhere too (dcv_bn_gn_act_bwd, dcv_act_bwd) use this to pick the cheaper source. */
int dcv_conv2d_wgrad_fuses_bias(const dcv_conv2d_desc* d);

/* Prepared filters.  The forward / data-gradient kernels read the OIHW filter re-laid out for the kernel which serves the layer (scalar-cache tiles for the
 * direct kernels, K-contiguous rows for the MFMA kernels); dcv_conv2d_fwd / dcv_conv2d_dgrad do that re-layout on every call.  A training loop whose
 * weights change once per optimisation step keeps the re-laid filters instead and refreshes ALL of them with one launch after the optimizer step
 * (the reference re-reads `Conv2d.weight` on every forward, submodule_creators.py:251; here that read is paid once per step).
 *   dcv_conv2d_filter_bytes     size of the prepared filter of (d, op), op 0 = forward, 1 = data gradient; 0 when that kernel reads OIHW directly.
 *                               op 2 = the weight [O][C*H*W] of a Linear behind Flatten (classification head, submodule_creators.py:268-269 after
 *                               parameters.yml:87) re-laid as [O][H*W][C]: the dcv_linear_* entry points then read the NHWC activations in place instead
 *                               of a transposed copy (descriptor: h, w, cin of the activations, cout = O, r = h, s = w, stride 1, no padding; the weight
 *                               gradient comes back in [O][C*H*W] order through dcv_slab_segment.tr_a / tr_b)
 *                               op 3 = the forward filter of a layer of the LDS-resident network (dcv_resident_forward): [R][S][cin rounded up to 4][cout]
 *                               op 4 = its data-gradient filter (dcv_resident_backward): transposed + flipped [R][S][cout4][cin4], or the matrix-core
 *                               B fragments [16][4][4][9] for 3x3 layers with 16 output channels on rows of 16 pixels (and not 4 input channels: 4 <- 16 runs on the VALU from the first layout)
 *   dcv_conv2d_prepare_filters  re-lays out `count` filters (entries whose size is 0 are skipped; filters[k] may then be NULL)
 *   dcv_conv2d_*_prepared       as dcv_conv2d_fwd / dcv_conv2d_dgrad with the filter prepared for exactly this descriptor (w still required: fallback paths) */
size_t dcv_conv2d_filter_bytes(const dcv_conv2d_desc* d, int op);
int dcv_conv2d_prepare_filters(int count, const dcv_conv2d_desc* const* descs, const int* ops, const float* const* w, void* const* filters, void* stream);
/* `stats` (nullable): the kernel's epilogue also emits per-block sums of y and y^2 per output channel, [2][Cout][rows] doubles with
 * rows = dcv_conv2d_fwd_stats_rows(d) (0: this kernel cannot, pass NULL) — the BatchNorm statistics partials of dcv_bn_gn_fwd_train for free */
int dcv_conv2d_fwd_stats_rows(const dcv_conv2d_desc* d);
/* The same query with the alignment of the rows to samples: *rows_per_sample > 0 means no row mixes samples and sample n owns rows
 * [n * rows_per_sample, (n + 1) * rows_per_sample) — per-(sample, channel) sums, what the GroupNorm statistics of dcv_norm_chain_fwd are computed from. */
int dcv_conv2d_fwd_stats_layout(const dcv_conv2d_desc* d, int* rows, int* rows_per_sample);
int dcv_conv2d_fwd_prepared(const dcv_conv2d_desc* d, const float* x, const float* w, const void* filter, const float* bias, float* y, double* stats, void* stream);
int dcv_conv2d_dgrad_prepared(const dcv_conv2d_desc* d, const float* dy, const float* w, const void* filter, float* dx, void* stream);
/* dx = data gradient + addend, in the data-gradient kernel's epilogue: `addend` ([N][H][W][Cin] like dx) is the gradient which reaches the layer's input over
 * another edge of the graph (a `_from` link of the reference's architecture grammar, nn_spec.py:218-243), so that autograd's separate accumulation kernel
 * disappears.  DCV_ERR_UNSUPPORTED (nothing launched) when the kernel serving this layer has no such epilogue: add separately then. */
int dcv_conv2d_dgrad_prepared_add(const dcv_conv2d_desc* d, const float* dy, const float* w, const void* filter, const float* addend, float* dx, void* stream);

/* ---- bf16 STORAGE path ("bf16 MFMA conv" of BASELINE.json configs 3 / 5; replaces torch.nn.Conv2d under bf16 autocast, submodule_creators.py:251) ----
 * Activations (x, y, dy, dx) are bf16 NHWC tensors in HBM, the prepared filter is bf16 [tap][cout][cin]; LDS holds bf16, the matrix cores run
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation; master weights, bias, statistics (fp64 partials) and weight gradients stay fp32.
 * Shapes: NHWC, groups 1, input channels of the kernel a multiple of 64, output channels a multiple of 64, tensors below 2 GiB; op 0 forward (any
 * stride / padding / dilation), op 1 data gradient (stride <= 4), op 2 weight gradient (3x3, stride 1, pad 1).  Otherwise DCV_ERR_UNSUPPORTED: cast with
 * dcv_cast_* and use the fp32 entry points (math DCV_MATH_BF16 keeps bf16 operand rounding there).
 * `y_dtype` / `dx_dtype`: element type of the tensor written. */
#define DCV_DTYPE_F32 0
#define DCV_DTYPE_BF16 1
int dcv_conv2d_bf16_supported(const dcv_conv2d_desc* d, int op);
size_t dcv_conv2d_bf16_filter_bytes(const dcv_conv2d_desc* d, int op);                        /* op 0 / 1 */
/* ONE launch (per 24 pieces) rounding and re-laying out `count` fp32 OIHW master filters into their bf16 prepared forms */
int dcv_conv2d_bf16_prepare_filters(int count, const dcv_conv2d_desc* const* descs, const int* ops, const float* const* w, void* const* filters, void* stream);
int dcv_conv2d_bf16_fwd_stats_layout(const dcv_conv2d_desc* d, int* rows, int* rows_per_sample);   /* as dcv_conv2d_fwd_stats_layout, for the kernel below */
int dcv_conv2d_bf16_fwd(const dcv_conv2d_desc* d, const void* x_bf16, const void* filter, const float* bias, void* y, int y_dtype, double* stats, void* stream);
int dcv_conv2d_bf16_dgrad(const dcv_conv2d_desc* d, const void* dy_bf16, const void* filter, void* dx, int dx_dtype, void* stream);
/* the stride-1 data gradient (dx bf16) with a fused epilogue, either or both of:
 *   addend (nullable): a second gradient of dx's shape (bf16) added to the result before it is stored and summed — the residual link's gradient, which autograd's
 *     engine otherwise adds with a kernel of its own (reference submodule_creators.py:272-327, the `_from` links);
 *   y_prev / mean / rstd / moments (nullable as a group): what the normalisation backward of the BatchNorm-only layer below needs from dx (autograd's BatchNorm
 *     backward, reference nn.py:487-501): moments[0][c][row] = sum of dx, moments[1][c][row] = sum of dx * (y_prev - mean[c]) * rstd[c] over row blocks of 128
 *     pixels (moments_rows = ceil(n h w / 128)); y_prev: that layer's activated convolution output, bf16 [n][h][w][cin]; hand them to dcv_norm_chain_bwd
 *     (dcv_norm_chain_desc::moments).
 * DCV_ERR_UNSUPPORTED when the kernel serving the shape has no such epilogue: call dcv_conv2d_bf16_dgrad (and add / sum separately). */
int dcv_conv2d_bf16_dgrad_moments(const dcv_conv2d_desc* d, const void* dy_bf16, const void* filter, void* dx_bf16, const void* addend_bf16, const void* y_prev_bf16,
                                  const float* mean, const float* rstd, double* moments, int moments_rows, void* stream);
size_t dcv_conv2d_bf16_wgrad_workspace_bytes(const dcv_conv2d_desc* d);
int dcv_conv2d_bf16_wgrad(const dcv_conv2d_desc* d, const void* x_bf16, const void* dy_bf16, float* dw, void* ws, size_t ws_bytes, void* stream);
/* element type conversion (round to nearest even, NaN stays NaN); n elements, x 16-B (fp32) / 8-B (bf16) aligned */
int dcv_cast_f32_to_bf16(const float* x, void* y_bf16, size_t n, void* stream);
/* A first layer (cin <= 4, any kernel / stride / padding / dilation: the 3x3 layer of a CIFAR spec, the 7x7 / 2 stem of a ResNet) on the matrix cores: its
 * input is expanded ONCE into the bf16 matrix out[n ho wo][kpad] with out[m][(ci R + r) S + s] = x[n][ci][ho sh - ph + r dh][wo sw - pw + s dw] (zero outside the
 * image and in the padding columns k >= cin R S; kpad: a multiple of 64, at most DCV_IM2COL_MAX_K), after which the layer is a 1x1 convolution of kpad
 * channels whose filter is the OIHW filter flattened per output channel and zero-padded to kpad: dcv_conv2d_bf16_fwd / _wgrad serve it (the one-tap weight
 * gradient), no data gradient is needed.  x: fp32 in d->x_layout (the dataloader's NCHW batch is read in place).  replaces the first Conv2d of a wide spec
 * in bf16 math mode (submodule_creators.py:251). */
#define DCV_IM2COL_MAX_K 256
int dcv_im2col_bf16(const dcv_conv2d_desc* d, const float* x, void* out_bf16, int kpad, void* stream);
/* A STRIDED first layer without the im2col matrix (round 4: the 7x7 / 2 stem of resnet18_style at 512 images wrote and re-read 2.5 GB of it per step).  The fp32
 * input goes ONCE through space-to-depth into the bf16 NHWC tensor out[n][hs][ws][cp], out[n][i][j][(a sw + b) cin + c] = x[n][c][i sh + a - ph][j sw + b - pw]
 * (zeros outside the image and in the channels >= sh sw cin; cp a multiple of 8, at most 64; d: the ORIGINAL layer, x in d->x_layout), over which the layer is a
 * stride-1, pad-0 convolution of A x A taps, A = ceil(R / stride), hs = ho + A - 1, ws = wo + A - 1, with the filter
 * w2[co][(a sw + b) cin + c][ra][rb] = w[co][c][ra sh + a][rb sw + b] (zero beyond R, S).  With cp A = 64 that layer is one the bf16 entry points serve in place
 * ("folded rows": dcv_conv2d_bf16_supported(d2, 0 / 2) for cin = cp, r = s = A — forward and weight gradient; there is no data gradient), 217 MB instead of
 * 2.5 GB at 512 x 3 x 224 x 224.  replaces the first Conv2d of a wide spec in bf16 math mode (submodule_creators.py:251). */
int dcv_space_to_depth_bf16(const dcv_conv2d_desc* d, const float* x, void* out_bf16, int hs, int ws, int cp, void* stream);
int dcv_cast_bf16_to_f32(const void* x_bf16, float* y, size_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K4  elementwise activation (standalone; conv/linear fuse it into their epilogue)
 * replaces `act_fn()` modules of nn.py:553 (ReLU parameters.yml:9, LeakyReLU :80, Sigmoid :88)
 * ------------------------------------------------------------------------------------------------- */
int dcv_act_fwd(const float* x, float* y, size_t numel, int act, float slope, void* stream);
/* dx = dy * act'(.) evaluated from the activation OUTPUT y (all supported activations allow it).
 * If db != NULL, also reduces db[c] = sum over rows of dx for a [rows][channels] view (bias gradient);
 * needs ws of dcv_act_bwd_workspace_bytes(). */
size_t dcv_act_bwd_workspace_bytes(size_t rows, int channels);
int dcv_act_bwd(const float* dy, const float* y, float* dx, size_t rows, int channels, int act, float slope,
                float* db, void* ws, size_t ws_bytes, void* stream);

/* out[c] = sum over rows of x[rows][channels] (ordered two-stage reduction; ws of dcv_act_bwd_workspace_bytes) — the bias
 * gradient of a conv / linear layer without fused activation. */
int dcv_channel_sum(const float* x, float* out, size_t rows, int channels, void* ws, size_t ws_bytes, void* stream);
int dcv_channel_sum_bf16(const void* x_bf16, float* out, size_t rows, int channels, void* ws, size_t ws_bytes, void* stream);   /* bf16 rows, fp32 sums */

/* ---------------------------------------------------------------------------------------------------
 * K5  BatchNorm (training + inference), NHWC view [rows = N*H*W][C]
 * replaces torch.nn.BatchNorm2d built at nn.py:418,487-491,501
 * ------------------------------------------------------------------------------------------------- */
size_t dcv_batchnorm_workspace_bytes(size_t rows, int channels);
/* Training forward: batch statistics (biased var for normalisation, unbiased for running_var), running stats EMA
 * with `momentum` (if running_mean/var != NULL), num_batches_tracked += 1 (if != NULL).
 * save_mean / save_rstd [C] are outputs kept for the backward.  gamma/beta may be NULL (affine=False). */
int dcv_batchnorm_fwd_train(const float* x, float* y, size_t rows, int channels, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked,
                            float momentum, float eps, float* save_mean, float* save_rstd,
                            void* ws, size_t ws_bytes, void* stream);
/* Inference forward from running statistics. */
int dcv_batchnorm_fwd_eval(const float* x, float* y, size_t rows, int channels, const float* gamma, const float* beta,
                           const float* running_mean, const float* running_var, float eps,
                           void* ws, size_t ws_bytes, void* stream);
/* Backward of the training forward: dx, dgamma, dbeta (dgamma/dbeta may be NULL when affine=False). */
int dcv_batchnorm_bwd(const float* dy, const float* x, float* dx, size_t rows, int channels, const float* gamma,
                      const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                      void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K6  GroupNorm, NHWC [N][HW][C], `groups` groups of C/groups channels
 * replaces torch.nn.GroupNorm built at nn.py:494-495,501 (parameters.yml:10)
 * ------------------------------------------------------------------------------------------------- */
size_t dcv_groupnorm_workspace_bytes(int n, int channels);
int dcv_groupnorm_fwd(const float* x, float* y, int n, int hw, int channels, int groups, const float* gamma, const float* beta,
                      float eps, float* save_mean, float* save_rstd /* [N][groups] */, void* stream);
int dcv_groupnorm_bwd(const float* dy, const float* x, float* dx, int n, int hw, int channels, int groups, const float* gamma,
                      const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                      void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K6b  LayerNorm's per-element affine, NHWC [N][HW][C] activations
 * replaces torch.nn.LayerNorm of the normalisation factory (nn.py:444, normalized_shape = input_shape[1:] at nn.py:492-493: the map's [H][W]).  The
 * statistics are dcv_groupnorm_fwd / _bwd with groups = channels (weight [H][W], per_channel = 0) or groups = 1 (weight [C][H][W], per_channel = 1) and
 * no affine; these entry points apply y = xhat * weight + bias in place over xhat, and give dxhat = dy * weight (the dy of dcv_groupnorm_bwd), dweight
 * and dbias (fp64 sums in a fixed order).  `ws`: dcv_layernorm_affine_workspace_bytes(hw, channels).
 * ------------------------------------------------------------------------------------------------- */
size_t dcv_layernorm_affine_workspace_bytes(int hw, int channels);
int dcv_layernorm_affine_fwd(float* y, int n, int hw, int channels, int per_channel, const float* weight, const float* bias /* may be NULL */, void* stream);
int dcv_layernorm_affine_bwd(const float* dy, const float* x, const float* save_mean, const float* save_rstd, int n, int hw, int channels, int per_channel,
                             const float* weight, float* dxhat, float* dweight, float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K6c  LocalResponseNorm over the channels of a pixel, NHWC [rows][C]
 * replaces torch.nn.LocalResponseNorm of the normalisation factory (nn.py:445): y = x * (k + alpha / size * sum of x^2 over the channel window)^-beta,
 * the window [c - size / 2, c + (size - 1) / 2] clipped to the tensor (torch's zero padding).
 * ------------------------------------------------------------------------------------------------- */
int dcv_lrn_fwd(const float* x, float* y, size_t rows, int channels, int size, float alpha, float beta, float k, void* stream);
int dcv_lrn_bwd(const float* dy, const float* x, float* dx, size_t rows, int channels, int size, float alpha, float beta, float k, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K5 split in two phases for cross-GPU statistics (`use_sync_batch_norm`, ignite_training.py:203,386-388 -> torch.nn.SyncBatchNorm; collective C4):
 * the caller all-reduces `sums` ([2][channels] doubles) over the process group between the phases and passes the global row count.
 *   forward : dcv_batchnorm_stats (sum x, sum x^2 of the local rows)            -> all-reduce -> dcv_batchnorm_fwd_from_sums
 *   backward: dcv_batchnorm_bwd_stats (sum dy, sum dy*xhat: the LOCAL d_beta, d_gamma) -> all-reduce -> dcv_batchnorm_bwd_from_sums
 * workspace: dcv_batchnorm_workspace_bytes(rows, channels).
 * ------------------------------------------------------------------------------------------------- */
int dcv_batchnorm_stats(const float* x, size_t rows, int channels, double* sums, void* ws, size_t ws_bytes, void* stream);
int dcv_batchnorm_fwd_from_sums(const float* x, float* y, size_t rows, int channels, const double* sums, double total_rows, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream);
int dcv_batchnorm_bwd_stats(const float* dy, const float* x, size_t rows, int channels, const float* save_mean, const float* save_rstd, double* sums,
                            void* ws, size_t ws_bytes, void* stream);
int dcv_batchnorm_bwd_from_sums(const float* dy, const float* x, float* dx, size_t rows, int channels, const float* gamma, const float* save_mean,
                                const float* save_rstd, const double* sums, double total_rows, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K5+K6 fused: the normalisation chain of one reference layer, Conv -> act -> BatchNorm [-> GroupNorm] (op order of layer(), nn.py:553; norm stack
 * of normalization_techniques(), nn.py:448-516), two launches per direction.  x is the activated convolution output, groups == 0 means "no GroupNorm".
 *   forward : y = GN(BN(x)); batch statistics -> bn_save_mean / bn_save_rstd [C] and running stats; GroupNorm statistics -> gn_save_* [N][groups];
 *             the BatchNorm output is never materialised.
 *   backward: dz = d loss / d (pre-activation convolution output) given dout = d loss / d y: GroupNorm backward (on the recomputed BN output), BatchNorm
 *             backward and the activation derivative (from its output x) in one pass; parameter gradients and the convolution bias gradient
 *             (sum of dz over rows) are produced by the same launches.  dz may not alias dout or x.
 * replaces what autograd runs for the BatchNorm2d / GroupNorm / activation modules built at nn.py:487-501,553 during ignite_training.py:253.
 * ------------------------------------------------------------------------------------------------- */
size_t dcv_norm_chain_workspace_bytes(int n, int hw, int channels);
int dcv_bn_gn_fwd_train(const float* x, float* y, int n, int hw, int channels,
                        const float* bn_gamma, const float* bn_beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                        float momentum, float bn_eps, float* bn_save_mean, float* bn_save_rstd,
                        int groups, const float* gn_gamma, const float* gn_beta, float gn_eps, float* gn_save_mean, float* gn_save_rstd,
                        const double* stats /* nullable: [2][channels][stats_rows] partial sums of x, x^2 from the producer's epilogue */, int stats_rows,
                        void* ws, size_t ws_bytes, void* stream);
int dcv_bn_gn_act_bwd(const float* dout, const float* x, float* dz, int n, int hw, int channels,
                      const float* bn_gamma, const float* bn_beta, const float* bn_save_mean, const float* bn_save_rstd,
                      int groups, const float* gn_gamma, const float* gn_save_mean, const float* gn_save_rstd,
                      int act, float act_slope,
                      float* d_bn_gamma, float* d_bn_beta, float* d_gn_gamma, float* d_gn_beta, float* d_bias /* all nullable */,
                      void* ws, size_t ws_bytes, void* stream);

/* The same chain followed by a non-overlapping average pooling (pool x pool windows, stride pool, no padding; torch.nn.AvgPool2d at submodule_creators.py:168
 * right after the layer, as in the reference's stock architecture): y / dout are the POOLED tensors [N][H/pool][W/pool][C]; the chain's full-size output is
 * never written, and the pooling launches of both directions disappear.  groups > 0 required when pool > 0; width = W of x (hw = H * W); pool == 0 is the
 * plain chain. */
int dcv_bn_gn_pool_fwd_train(const float* x, float* y, int n, int hw, int channels,
                             const float* bn_gamma, const float* bn_beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                             float momentum, float bn_eps, float* bn_save_mean, float* bn_save_rstd,
                             int groups, const float* gn_gamma, const float* gn_beta, float gn_eps, float* gn_save_mean, float* gn_save_rstd,
                             const double* stats, int stats_rows, int width, int pool, void* ws, size_t ws_bytes, void* stream);
int dcv_bn_gn_pool_act_bwd(const float* dout, const float* x, float* dz, int n, int hw, int channels,
                           const float* bn_gamma, const float* bn_beta, const float* bn_save_mean, const float* bn_save_rstd,
                           int groups, const float* gn_gamma, const float* gn_save_mean, const float* gn_save_rstd,
                           int act, float act_slope,
                           float* d_bn_gamma, float* d_bn_beta, float* d_gn_gamma, float* d_gn_beta, float* d_bias /* all nullable */,
                           int width, int pool, void* ws, size_t ws_bytes, void* stream);

/* The same chain (with or without GroupNorm, with or without fused pooling) as STREAMING kernels around tiny finalize kernels (norm_chain.hip): per sample,
 * GroupNorm(BatchNorm(x)) is an affine function of x, out = S[n][c] * x + T[n][c], and S / T follow from per-(sample, channel) sums of x and x^2 — which the
 * convolution epilogue already emits (dcv_conv2d_fwd_stats_layout).  Forward = [sums from the producer, else one streaming pass] + finalize (batch statistics,
 * running statistics) + apply (its prologue turns the block's sample sums into GroupNorm statistics, S, T); backward = four moments per (sample, channel) of
 * (dout, u = (x - mean) * rstd) with the sample's algebra in the same block + finalize (every parameter gradient) + apply dz = act'(x) * (P * dout + Q * u + R).
 * Every apply kernel moves each byte once.
 *   dcv_norm_chain_supported   1 when these kernels cover the shape (channel vectors per pixel a power of two <= 256; with GroupNorm: a power of two of
 *                              channels per group and samples of at most 8192 channel vectors, one block each); otherwise use dcv_bn_gn_pool_fwd_train /
 *                              dcv_bn_gn_pool_act_bwd, which read and write the same saved statistics
 *   dcv_norm_chain_coef_floats size of `coef` (scratch of one call: the forward keeps S, T there, the backward P, Q, R)
 *   stats / stats_rows / stats_rows_per_sample: [2][channels][stats_rows] sums of x, x^2 left by the producer (nullable); with groups > 0 they are only used
 *                              when aligned to samples (stats_rows == n * stats_rows_per_sample)
 * The convolution bias gradient: dcv_norm_chain_desc::d_bias (summed by the backward's last launch), or dcv_conv2d_wgrad (dcv_conv2d_wgrad_fuses_bias) / dcv_channel_sum. */
typedef struct dcv_norm_chain_desc {
    int n, hw, channels;           /* x: [n][hw][channels] fp32 NHWC, the activated convolution output */
    int width, pool;               /* pool > 0: out / dout are the pooled tensors [n][hw / pool^2][channels]; width = W of x */
    const float* bn_gamma; const float* bn_beta;           /* nullable (affine = False) */
    float* running_mean; float* running_var; int64_t* num_batches_tracked;   /* nullable; updated by the forward */
    float momentum, bn_eps;
    float* bn_mean; float* bn_rstd;                        /* [channels]: written by the forward, read by the backward */
    int groups;                                            /* 0: no GroupNorm */
    const float* gn_gamma; const float* gn_beta; float gn_eps;
    float* gn_mean; float* gn_rstd;                        /* [n][groups]: written by the forward, read by the backward */
    float* coef;                                           /* dcv_norm_chain_coef_floats(d) floats */
    int act; float act_slope;                              /* activation which produced x (its derivative is taken from x) */
    int io_dtype;                                          /* element type of x / out / dout / dz: DCV_DTYPE_F32 (0, default) or DCV_DTYPE_BF16 (bf16 storage path:
                                                              16-byte aligned tensors when channels % 8 == 0 without GroupNorm / pooling) */
    float* d_bias;                                         /* nullable; dcv_norm_chain_bwd only: [channels] sums of dz over (n, hw) — the bias gradient of the
                                                              convolution which produced x, summed while dz is written (no second pass over dz) */
    float* d_bias_part;                                    /* nullable (instead of d_bias); dcv_norm_chain_bwd only: [dcv_norm_chain_bias_rows(d)][channels] per-block
                                                              sums of dz — the caller sums the rows (dcv_slab_reduce: ONE launch for every layer of a step) */
    const double* moments; int moments_rows;               /* nullable; dcv_norm_chain_bwd only, groups == 0: [2][channels][moments_rows] sums of dout and of dout * u
                                                              (u = (x - bn_mean) bn_rstd) over row blocks of (n, hw), left by the kernel which produced dout
                                                              (dcv_conv2d_bf16_dgrad_moments): the backward's pass over dout and x for these sums is skipped */
    int pool_mode;                                         /* 0: `pool` (non-overlapping average pooling, or none); 1 (round 4): the chain ends in MaxPool2d(pool_k,
                                                              pool_stride, pool_pad) with OVERLAPPING windows (the 3x3 / 2 / 1 pooling behind a ResNet stem): out / dout are
                                                              the pooled tensors [n][ho][wo][channels], ho = (h + 2 pad - k) / stride + 1; BatchNorm only, bf16 tensors,
                                                              channels % 8 == 0, k <= 2 stride, pool == 0; torch semantics (first maximum wins, NaN wins; the values compared
                                                              are the bf16-rounded affine outputs the unfused pair compares); neither the full-size output nor its gradient
                                                              is written.  replaces torch.nn.MaxPool2d behind a layer (nn_spec.py:158-169) */
    int pool_k, pool_stride, pool_pad;
    unsigned char* pool_arg;                               /* pool_mode 1: [n][ho][wo][channels] window position r k + s of each maximum: written by the forward, read by
                                                              the backward; 8-byte aligned */
    const void* addend;                                    /* nullable; dcv_norm_chain_fwd only, pool == 0, pool_mode == 0: out = chain(x) + addend, addend a tensor of out's
                                                              shape and element type — a residual link's sum behind the layer (submodule_creators.py:272-327) in the
                                                              layer's own pass over x; the backward is unchanged (the sum hands its gradient to both inputs) */
    void* pool_x;                                          /* nullable; dcv_norm_chain_fwd only, pool > 0 or pool_mode == 1: the window means of x itself (pool_mode 1: x at the
                                                              window's winner), a tensor of out's shape
                                                              and element type.  With it the data-gradient kernel of the layer above can leave this layer's backward moments
                                                              (dcv_conv2d_bf16_dgrad_moments with y_prev = pool_x: a pooled gradient multiplies the window mean of u, and the
                                                              mean of (x - mean) rstd is (mean of x - mean) rstd), so the backward's sums pass over x and dout is skipped for
                                                              layers followed by avg_pooling (submodule_creators.py:163-176) or a MaxPool2d as it is for unpooled ones */
} dcv_norm_chain_desc;
int dcv_norm_chain_supported(const dcv_norm_chain_desc* d);
size_t dcv_norm_chain_coef_floats(const dcv_norm_chain_desc* d);
size_t dcv_norm_chain_ws_bytes(const dcv_norm_chain_desc* d);
int dcv_norm_chain_bias_rows(const dcv_norm_chain_desc* d);         /* rows of d_bias_part */
int dcv_norm_chain_fwd(const dcv_norm_chain_desc* d, const void* x, void* out, const double* stats, int stats_rows, int stats_rows_per_sample,
                       void* ws, size_t ws_bytes, void* stream);
int dcv_norm_chain_bwd(const dcv_norm_chain_desc* d, const void* dout, const void* x, void* dz,
                       float* d_bn_gamma, float* d_bn_beta, float* d_gn_gamma, float* d_gn_beta /* all nullable */, void* ws, size_t ws_bytes, void* stream);


/* ---------------------------------------------------------------------------------------------------
 * K7 / K7'  2-D average / max pooling, NHWC
 * replaces torch.nn.AvgPool2d (submodule_creators.py:168 -> nn.py:416) and torch.nn.MaxPool2d (nn_spec.py:158-169)
 * avg: count_include_pad=True, ceil_mode=False (torch defaults).  max: first maximum wins on ties.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int n, h, w, c;
    int kh, kw, stride_h, stride_w, pad_h, pad_w;
    int ho, wo;
    int mode;                    /* DCV_POOL_AVG | DCV_POOL_MAX */
} dcv_pool2d_desc;
/* argmax: int32 [N,Ho,Wo,C] flat input offset (h*W + w) of the selected element; required for DCV_POOL_MAX, else NULL */
int dcv_pool2d_fwd(const dcv_pool2d_desc* d, const float* x, float* y, int32_t* argmax, void* stream);
int dcv_pool2d_bwd(const dcv_pool2d_desc* d, const float* dy, const int32_t* argmax, float* dx, void* stream);
/* the same on bf16 NHWC tensors (the bf16 storage path of the wide specs): fp32 arithmetic, one rounding of the result — what widening, pooling in fp32 and
 * narrowing again produced, without the two cast launches and with half the bytes */
int dcv_pool2d_fwd_bf16(const dcv_pool2d_desc* d, const void* x_bf16, void* y_bf16, int32_t* argmax, void* stream);
int dcv_pool2d_bwd_bf16(const dcv_pool2d_desc* d, const void* dy_bf16, const int32_t* argmax, void* dx_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K8  dense / residual link helpers: bilinear resize, channel concat (slice copy), n-ary add
 * replaces F.interpolate (nn.py:675) + torch.cat (submodule_creators.py:56,323)
 * ------------------------------------------------------------------------------------------------- */
int dcv_interp_bilinear_fwd(const float* x, float* y, int n, int h, int w, int c, int ho, int wo, int align_corners, void* stream);
int dcv_interp_bilinear_bwd(const float* dy, float* dx, int n, int h, int w, int c, int ho, int wo, int align_corners, void* stream);
/* dst[row][dst_off : dst_off+copy_c] = src[row][src_off : src_off+copy_c] for [rows][dst_c] / [rows][src_c] views */
int dcv_copy_channels(const float* src, int src_c, int src_off, float* dst, int dst_c, int dst_off, int copy_c, size_t rows, void* stream);
/* dst[row] = srcs[0][row] | srcs[1][row] | ... (channel concatenation of `count` <= DCV_MAX_CONCAT [rows][channels[k]] blocks) in ONE launch, and its
 * adjoint (a NULL dsts[k] skips that block): what torch.cat(dim=1) / its backward do for the dense links (submodule_creators.py:309-327) */
int dcv_concat_channels(int count, const float* const* srcs, const int* channels, float* dst, size_t rows, void* stream);
int dcv_split_channels(int count, float* const* dsts, const int* channels, const float* src, size_t rows, void* stream);
/* The same with parts of any spatial size: part k is [n][heights[k]][widths[k]][channels[k]]; parts whose size differs from the result's [n][ho][wo] are
 * resized bilinearly on the fly (F.interpolate(mode='bilinear', align_corners) of the dense / residual links with allow_scaling, submodule_creators.py:272-327,
 * followed by torch.cat) — and, for the split, receive the bilinear backward of their channel range of the wide gradient.  One launch per direction. */
int dcv_resize_concat_channels(int count, const float* const* srcs, const int* channels, const int* heights, const int* widths, float* dst,
                               int n, int ho, int wo, int align_corners, void* stream);
int dcv_resize_split_channels(int count, float* const* dsts, const int* channels, const int* heights, const int* widths, const float* src,
                              int n, int ho, int wo, int align_corners, void* stream);
/* out = a + b (elementwise; out may alias a) */
int dcv_add(const float* a, const float* b, float* out, size_t numel, void* stream);
int dcv_add_bf16(const void* a_bf16, const void* b_bf16, void* out_bf16, size_t numel, void* stream);   /* bf16 tensors (16-byte aligned), fp32 sum rounded once */
/* out = alpha * a */
int dcv_scale(const float* a, float alpha, float* out, size_t numel, void* stream);
/* dst[i] = index[i] >= 0 ? src[index[i]] : 0 for i < numel — the re-layout of a small tensor through a host-made index map.  Replaces the torch operators
 * (pad / reshape / permute on the parameter, and their autograd) with which a first layer's OIHW filter (`submodule_creators.py:251`: Conv2d(**params)) was
 * regrouped for the im2col and folded-rows kernels: as a library launch the re-layout and its gradient are part of a replayed launch list. */
int dcv_gather_f32(const float* src, const int* index, float* dst, size_t numel, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Input side: uint8 images -> normalised fp32 activations on the device, fusing torchvision's ToTensor (HWC u8 -> CHW f32 / 255)
 * and Normalize ((x - mean[c]) / std[c]) of the reference's preprocessing recipe (conf/base/parameters.yml:196-201) with the layout change:
 *   src: [N][H][W][C] uint8 (how image datasets store pixels; a quarter of the fp32 bytes over PCIe)
 *   dst: fp32, DCV_LAYOUT_NCHW (what `DeepcvModule.forward` is fed, base_module.py:113) or DCV_LAYOUT_NHWC
 * mean / inv_std: device arrays of C floats.
 * ------------------------------------------------------------------------------------------------- */
int dcv_images_u8_normalize(const unsigned char* src, float* dst, int n, int h, int w, int c, int dst_layout,
                            const float* mean, const float* inv_std, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * GPU augmentation: AugMix-style chains of the reference's 13 image operations (src/deepcv/meta/data/augmentation.py:65-132, mixed like
 * augment_and_mix, :165-203) applied per sample ON THE DEVICE and fused with ToTensor + Normalize:
 *     out = normalize( (1 - m) * x + m * sum_i w_i * chain_i(x) ),   chain_i = up to DCV_AUG_MAX_OPS ops applied in order
 * Every operation reproduces Pillow's arithmetic bit for bit on the uint8 images (the reference calls ImageOps / Image.rotate / Image.transform / ImageEnhance):
 *   POSTERIZE keeps the `ip` high bits; SOLARIZE inverts pixels >= `ip`; AUTOCONTRAST / EQUALIZE: per-band histogram tables;
 *   TRANSLATE_X / _Y, ROTATE, SHEAR_X / _Y: Image.transform(size, AFFINE, m, resample=BILINEAR) with zero fill, `m` = the six coefficients Pillow derives from
 *   the pixel offset / angle / shear factor (computed by the caller like Image.rotate does);
 *   COLOR / CONTRAST / BRIGHTNESS / SHARPNESS: ImageEnhance.*(img).enhance(fp) = Image.blend(degenerate, img, fp) in float32.
 * The random draws (which ops, parameters, mixing weights) are made on the host by the caller (`deepcv_amd.meta.data.augmentation`), reproducible from a seed;
 * this entry point is deterministic.  src: [N][H][W][C] uint8, C = 1 (grey levels) or 3 (RGB); dst: fp32 NCHW or NHWC.  scratch: the two uint8 working images
 * of every (sample, chain), dcv_images_u8_augment_scratch_bytes(n, h, w, c) bytes; on return buffer 0 of (sample i, chain j), at offset
 * ((i * DCV_AUG_MAX_CHAINS + j) * 2) * h * w * c, holds the image that chain ended with.
 * ------------------------------------------------------------------------------------------------- */
#define DCV_AUG_MAX_CHAINS 3
#define DCV_AUG_MAX_OPS 3
#define DCV_AUG_NONE 0
#define DCV_AUG_POSTERIZE 1
#define DCV_AUG_SOLARIZE 2
#define DCV_AUG_TRANSLATE_X 3
#define DCV_AUG_TRANSLATE_Y 4
#define DCV_AUG_BRIGHTNESS 5
#define DCV_AUG_AUTOCONTRAST 6
#define DCV_AUG_EQUALIZE 7
#define DCV_AUG_ROTATE 8
#define DCV_AUG_SHEAR_X 9
#define DCV_AUG_SHEAR_Y 10
#define DCV_AUG_COLOR 11
#define DCV_AUG_CONTRAST 12
#define DCV_AUG_SHARPNESS 13
typedef struct {
    int op[DCV_AUG_MAX_CHAINS][DCV_AUG_MAX_OPS];      /* DCV_AUG_* (DCV_AUG_NONE = identity) */
    int ip[DCV_AUG_MAX_CHAINS][DCV_AUG_MAX_OPS];      /* integer parameter: bits, threshold */
    float fp[DCV_AUG_MAX_CHAINS][DCV_AUG_MAX_OPS];    /* enhancement factor */
    float weight[DCV_AUG_MAX_CHAINS];                 /* w_i (unused chains: 0) */
    float mix;                                        /* m in [0, 1]; 0 = original image */
    int reserved;
    double m[DCV_AUG_MAX_CHAINS][DCV_AUG_MAX_OPS][6]; /* AFFINE coefficients of the geometric ops (offset 128) */
} dcv_augmix_recipe;                                   /* 560 bytes */
size_t dcv_images_u8_augment_scratch_bytes(int n, int h, int w, int c);
int dcv_images_u8_augment_normalize(const unsigned char* src, float* dst, int n, int h, int w, int c, int dst_layout,
                                    const dcv_augmix_recipe* recipes /* device, [n] */, const float* mean, const float* inv_std,
                                    void* scratch, size_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Indirect batch staging: up to DCV_MAX_COPY_JOBS (src, dst, bytes) copies whose descriptors are READ BY THE KERNEL AT RUN TIME from `jobs`, a block
 * of device-visible (pinned host or device) memory.  Captured as the first node of the training-step hipGraph it lets every replay fetch a NEW batch
 * (and the optimizer's scalar hyper-parameters) although the graph's kernel arguments are frozen: the host only rewrites the mailbox before
 * `hipGraphLaunch` — no hipMemcpyAsync calls (each ~20 us of host latency on the critical path of a 0.6 ms step).
 * replaces the H2D / D2D batch copies of process_function (ignite_training.py:235-242).  Sources may be pinned host memory (zero-copy read).
 * ------------------------------------------------------------------------------------------------- */
#define DCV_MAX_COPY_JOBS 8
typedef struct {
    const void* src[DCV_MAX_COPY_JOBS];
    void* dst[DCV_MAX_COPY_JOBS];
    unsigned long long bytes[DCV_MAX_COPY_JOBS];
    int count;
    int seq;                        /* sequence number of this mailbox content, echoed by dcv_publish_scalars */
} dcv_copy_jobs;
int dcv_indirect_copy(const dcv_copy_jobs* jobs, void* stream);
/* Early read-back: copies `n` <= DCV_MAX_COPY_JOBS device scalars (the loss terms) to host-visible memory `host_vals`, then (system-scope fence) stores the
 * mailbox's `seq` to `host_seq`.  Placed right after the loss kernels of the step graph it lets the host fetch the per-step loss of process_function
 * (ignite_training.py:255) by polling pinned memory while the backward pass is still running, and launch the next step's graph without a device sync. */
int dcv_publish_scalars(const dcv_copy_jobs* mailbox, const float* const* srcs, int n, float* host_vals, int* host_seq, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K9  layout change + Linear
 * replaces torch.nn.Flatten (parameters.yml:87) on NHWC activations and torch.nn.Linear (submodule_creators.py:268-269)
 * ------------------------------------------------------------------------------------------------- */
/* [N][A][B] -> [N][B][A]  (NHWC->NCHW flatten order: A=HW, B=C;  NCHW->NHWC: A=C, B=HW) */
int dcv_transpose_inner(const float* x, float* y, int n, int a, int b, void* stream);
/* whole backward of y = act(x W^T + b) in three launches: dx = dz W, dw = dz^T x, db = sum dz with dz = dy * act'(y) evaluated on the fly by every
 * consumer (no dz tensor, no separate activation-backward pass; the bias gradient rides in the weight-gradient slabs).  y may be NULL when act is
 * DCV_ACT_NONE; dx / dw / db are nullable. */
size_t dcv_linear_bwd_workspace_bytes(int n, int in_features, int out_features);
int dcv_linear_bwd(const float* dy, const float* y, int act, float act_slope, const float* x, const float* w, float* dx, float* dw, float* db,
                   int n, int in_features, int out_features, void* ws, size_t ws_bytes, void* stream);
/* as dcv_linear_bwd, with the ordered sum of the weight / bias gradient slabs left to dcv_slab_reduce (see dcv_conv2d_wgrad_deferred) */
int dcv_linear_bwd_deferred(const float* dy, const float* y, int act, float act_slope, const float* x, const float* w, float* dx, float* dw, float* db,
                            int n, int in_features, int out_features, void* ws, size_t ws_bytes,
                            dcv_slab_segment* segments /* [2] */, int* n_segments, void* stream);
size_t dcv_linear_wgrad_workspace_bytes(int n, int in_features, int out_features);
/* y[N][O] = act(x[N][K] . w[O][K]^T + bias) */
int dcv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int n, int in_features, int out_features,
                   int act, float slope, void* stream);
/* dx[N][K] = dy[N][O] . w[O][K] */
int dcv_linear_dgrad(const float* dy, const float* w, float* dx, int n, int in_features, int out_features, void* stream);
/* dw[O][K] = dy^T . x ; db[O] = sum_n dy (db may be NULL) */
int dcv_linear_wgrad(const float* x, const float* dy, float* dw, float* db, int n, int in_features, int out_features,
                     void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K10  softmax cross-entropy (mean reduction) on logits [N][C] with int64 targets
 * replaces torch.nn.CrossEntropyLoss() of classification/image.py:70
 * ------------------------------------------------------------------------------------------------- */
/* loss[0] = mean_n (logsumexp(z_n) - z_n[t_n]); lse[N] saved for the backward.  Also counts argmax==target into
 * correct[0] (int32) when correct != NULL (ignite Accuracy, classification/image.py:72). */
size_t dcv_cross_entropy_workspace_bytes(int n);
int dcv_cross_entropy_fwd(const float* logits, const int64_t* target, float* loss, float* lse, int32_t* correct,
                          int n, int classes, void* ws, size_t ws_bytes, void* stream);
/* dlogits[n][c] = (exp(z - lse_n) - [c == t_n]) * dloss[0] / N */
/* as dcv_cross_entropy_fwd; small batches of few classes (one launch computes the whole loss) can also write d loss / d logits for an upstream gradient of
 * exactly 1 into `dlogits_unit` [N][classes] (nullable) — *produced tells whether they did: a training step whose loss is the root of the backward
 * pass (ignite_training.py:253, loss.backward()) then needs no dcv_cross_entropy_bwd launch. */
int dcv_cross_entropy_fwd_grad(const float* logits, const int64_t* target, float* loss, float* lse, int32_t* correct, float* dlogits_unit, int* produced,
                               int n, int classes, void* ws, size_t ws_bytes, void* stream);
/* as dcv_cross_entropy_fwd_grad; the one-launch case also publishes the loss to pinned host memory the way dcv_publish_scalars does (the value into
 * host_vals[0 .. slots), then mailbox->seq into *host_seq), sparing a replayed training step that launch — *published tells whether it did; host_seq NULL:
 * nothing is published.  PACKED RECORD: when host_seq is the word right behind host_vals[0] and host_vals is 8-byte aligned, value and sequence number leave
 * as ONE 64-bit system-scope store {value, seq} and no fence is issued on the device (slots is then 1); the host polls the sequence word and reads the
 * value of the same write.  The same rule holds for the loss published by dcv_resident_backward's head (dcv_resident_bwd_head.pub_*). */
int dcv_cross_entropy_fwd_grad_publish(const float* logits, const int64_t* target, float* loss, float* lse, int32_t* correct, float* dlogits_unit, int* produced,
                                       int n, int classes, void* ws, size_t ws_bytes, const dcv_copy_jobs* mailbox, float* host_vals, int* host_seq, int slots,
                                       int* published, void* stream);
int dcv_cross_entropy_bwd(const float* logits, const int64_t* target, const float* lse, const float* dloss, float* dlogits,
                          int n, int classes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K11  fused multi-tensor optimizer steps (fp32 master params)
 * replaces torch.optim.AdamW (classification/image.py:71, opts parameters.yml:129-134) and torch.optim.SGD (:136)
 * Scalars that change from step to step live in DEVICE memory (`hyper`), so that a captured hipGraph replays with the
 * current learning rate: the host refreshes `hyper` with one small async H2D copy per step.
 * ------------------------------------------------------------------------------------------------- */
#define DCV_MAX_TENSORS 64
#define DCV_MAX_CONCAT 8
typedef struct {
    int count;                               /* number of tensors in this launch (<= DCV_MAX_TENSORS)     */
    float* param[DCV_MAX_TENSORS];
    const float* grad[DCV_MAX_TENSORS];
    float* state1[DCV_MAX_TENSORS];          /* AdamW: exp_avg      | SGD: momentum buffer (or NULL)      */
    float* state2[DCV_MAX_TENSORS];          /* AdamW: exp_avg_sq   | SGD: unused                         */
    float* state3[DCV_MAX_TENSORS];          /* AdamW: max_exp_avg_sq when amsgrad, else NULL             */
    int64_t numel[DCV_MAX_TENSORS];
} dcv_tensor_list;

/* hyper layout (float[16], device): see DCV_HYP_* ; step counter: int64 on device, incremented by *_prepare. */
#define DCV_HYP_LR 0
#define DCV_HYP_BETA1 1          /* SGD: momentum   */
#define DCV_HYP_BETA2 2          /* SGD: dampening  */
#define DCV_HYP_EPS 3
#define DCV_HYP_WEIGHT_DECAY 4
#define DCV_HYP_GRAD_SCALE 5     /* gradients are multiplied by this (1/world_size for data-parallel mean) */
#define DCV_HYP_BIAS_CORR1 6     /* written by dcv_adamw_prepare */
#define DCV_HYP_BIAS_CORR2 7     /* written by dcv_adamw_prepare */
#define DCV_HYP_FIRST_STEP 8     /* written by *_prepare: 1.0 on the first step (SGD momentum buffer init) */
#define DCV_HYP_COUNT 16
/* step[0] += 1, then derive the bias corrections 1 - beta^step in double precision */
int dcv_optim_prepare(float* hyper, int64_t* step, void* stream);
int dcv_adamw_step(const dcv_tensor_list* tl, const float* hyper, int amsgrad, void* stream);
int dcv_sgd_step(const dcv_tensor_list* tl, const float* hyper, int nesterov, void* stream);
/* AdamW step AND the re-layout of the prepared filters of the updated weights in ONE launch, for small parameter sets (a captured training step of the
 * stock classifier: 2 launches less).  A block owns a whole tensor (or a group of rows of a Linear weight): update, block barrier, then every prepared
 * layout of that tensor (dcv_conv2d_prepare_filters ops 0 - 4) gathered from the updated values.  `dcv_adamw_prep_build` (host only) packs the tensors of
 * `lists` and the filters (descs / ops / weights / filters as for dcv_conv2d_prepare_filters; every weight must be a `param` of the lists) into `out`;
 * DCV_ERR_UNSUPPORTED when the set does not fit (more than DCV_OPT_MAX_ITEMS blocks, DCV_OPT_MAX_PREPS layouts, a tensor above DCV_OPT_MAX_NUMEL elements
 * which cannot be split, amsgrad state).  Same arithmetic as dcv_adamw_step + dcv_conv2d_prepare_filters, bit for bit. */
#define DCV_OPT_MAX_ITEMS 40
#define DCV_OPT_MAX_PREPS 24
#define DCV_OPT_MAX_NUMEL 4096
typedef struct dcv_opt_item { float* param; const float* grad; float* state1; float* state2; int numel; short prep_first; short prep_count; } dcv_opt_item;
typedef struct dcv_adamw_prep_table { int items; int preps; dcv_opt_item item[DCV_OPT_MAX_ITEMS]; unsigned char prep[DCV_OPT_MAX_PREPS * 88]; } dcv_adamw_prep_table;
int dcv_adamw_prep_build(const dcv_tensor_list* lists, int n_lists, int n_filters, const dcv_conv2d_desc* const* descs, const int* ops,
                         const float* const* weights, void* const* filters, dcv_adamw_prep_table* out);
int dcv_adamw_step_prep(const dcv_adamw_prep_table* table, const float* hyper, void* stream);
/* ... with the step's pending ordered slab sums in front, in the SAME launch (round 4: `optimizer.step()` at ignite_training.py:254 behind `loss.backward()` at :253
 * as one trailing launch instead of dcv_slab_reduce + dcv_adamw_step_prep): a thread which finishes an entry of a segment stores the gradient, applies AdamW to the
 * parameter behind it and scatters the new value into that tensor's prepared layouts.  Same bits as the two launches.  DCV_ERR_UNSUPPORTED (nothing launched): more than
 * DCV_MAX_SLAB_SEGMENTS segments, a segment which does not write exactly one whole item's gradient, a prepared layout without an inverse map. */
int dcv_adamw_step_prep_slabs(const dcv_adamw_prep_table* table, int n_segments, const dcv_slab_segment* segments, const float* hyper, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * C1 helper  gradient bucket pack for the data-parallel all-reduce (RCCL over xGMI is driven from Python through
 * torch.distributed; the bucket is what gets reduced).  replaces DDP's bucket copy (ignite_training.py:380)
 * dst_flat[offset_i : offset_i + numel_i] = grad_i  (offsets = running sum of numel in list order); a tensor whose tl->param[i] is set is copied there
 * instead (bucket slices padded to 16-B boundaries).  Gradients which the backward kernels already wrote into their slice need no pack at all.
 * ------------------------------------------------------------------------------------------------- */
int dcv_pack_grads(const dcv_tensor_list* tl, float* dst_flat, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * C1  one-shot mean all-reduce of a small gradient bucket across the GPUs of one node, without a collective library (allreduce.hip).
 * replaces the bucket all-reduce of DistributedDataParallel (ignite_training.py:373-390) for models whose whole gradient is one bucket which only exists
 * when the backward pass has ended (the stock classifier: 68 KB): a latency problem.  Every rank owns an INBOX (device memory, one allocation, mapped into
 * the peer processes with hipIpc); a launch pushes its values as 8-byte {epoch, value} granules into every peer's inbox over xGMI, polls its own inbox for
 * the others' and adds the `world` values in rank order (the same bits on every rank), then scales by 1 / world, in place.  The epoch lives in the inbox
 * header and advances with every launch: a replayed launch list / hipGraph re-issues the identical call.  The inbox is FINE-GRAINED device memory (a running
 * kernel polls words which other devices store; DCV_ALLREDUCE_MEM=uncached|coarse for experiments).  A late rank is waited for, as DistributedDataParallel
 * would: the wait is bounded by time (DCV_ALLREDUCE_TIMEOUT_S, default 600 s); when it expires the whole launch fails together — error word in the header
 * (dcv_allreduce_status -> 1 + the silent rank) and in the pinned host word registered with dcv_allreduce_set_error_word (no synchronisation needed to look at
 * it), NaN in the bucket, and every later launch on this inbox yields NaN too.
 *   setup (once per process group):  bytes = dcv_allreduce_inbox_bytes(world, max_floats);  dcv_allreduce_alloc(bytes, &inbox);  dcv_allreduce_export(inbox, handle);
 *          exchange the handles (any host channel, e.g. torch.distributed.all_gather_object);  dcv_allreduce_import(handle_p, &inboxes[p]) for every p != rank,
 *          inboxes[rank] = inbox;   teardown: dcv_allreduce_unmap(inboxes[p]), dcv_allreduce_free(inbox)
 *   per step:  dcv_allreduce_oneshot(bucket, count, rank, world, inboxes, max_floats, stream)   — every rank the same count, the same number of calls
 * The only entry points of this library which allocate / map device memory (hipIpc exports whole allocations) and synchronise (alloc, status, set_error_word).
 * ------------------------------------------------------------------------------------------------- */
#define DCV_ALLREDUCE_MAX_RANKS 16
#define DCV_ALLREDUCE_HANDLE_BYTES 64
size_t dcv_allreduce_inbox_bytes(int world, size_t max_floats);
int dcv_allreduce_alloc(size_t bytes, void** inbox);
int dcv_allreduce_free(void* inbox);
int dcv_allreduce_export(void* inbox, unsigned char* handle);
int dcv_allreduce_import(const unsigned char* handle, void** peer_inbox);
int dcv_allreduce_unmap(void* peer_inbox);
int dcv_allreduce_oneshot(float* bucket, size_t count, int rank, int world, void* const* inboxes, size_t max_floats, void* stream);
int dcv_allreduce_status(const void* inbox, void* stream, int* err);
int dcv_allreduce_set_error_word(void* inbox, unsigned* host_word);   /* host_word: 4 bytes of pinned, device-visible host memory (0 now); NULL unregisters; synchronises */

/* ---------------------------------------------------------------------------------------------------
 * LDS-resident network (resident.hip): the training forward of a small-channel DeepcvModule as ONE persistent launch.
 * replaces the per-sub-module dispatch of DeepcvModule.forward (src/deepcv/meta/base_module.py:113-155) over `layer()` stacks
 * (src/deepcv/meta/nn.py:519-554: Conv2d -> act -> BatchNorm2d [-> GroupNorm]), avg_pooling (submodule_creators.py:163-176), dense links
 * (submodule_creators.py:272-332) and the Flatten + fully connected head (parameters.yml:87-88), for models whose per-sample activations fit the LDS.
 * One block of 1024 threads keeps TWO samples resident in LDS from the input image to the logits; grid = ceil(n / 2) <= 256 blocks, all co-resident
 * (<= 80 KB of LDS per block: two such grids fit the chip).  The only traffic between blocks is the BatchNorm batch statistics: per layer every block
 * publishes its sums of y and y^2 as 8-byte {epoch, value} granules and gathers everybody's (agent-scope stores / loads, no fence, bounded spin);
 * GroupNorm, pooling, the link and the head are per-sample and stay inside the block.  Every tensor the per-layer backward kernels read (y, the layer
 * outputs, the saved statistics) is written to HBM exactly as the per-layer forward writes it.
 *   program     ops in execution order; each op consumes the previous op's output (the first one the model input `x`)
 *   LAYER       conv (stride 1, padding k / 2) + bias + act [+ BatchNorm, batch statistics + running statistics] [+ GroupNorm] [+ avg pool `pool` x `pool`]
 *               `filt`: the filter prepared with dcv_conv2d_prepare_filters op 3 ([k][k][cin rounded up to 4][cout]); 16 -> 16 channels: the OIHW weight
 *   LINK        channel concatenation [current, bilinear resize of `ref` to the current size] (dense link with allow_scaling); `ref` must be the `out`
 *               of an earlier LAYER of this program; must be followed by FC or end the program
 *   FC          Flatten + Linear (+ act) on the NHWC activations: `filt` = the [O][H*W][C] weight (dcv_conv2d_prepare_filters op 2), `out` = logits
 *   ws          persistent, zero-initialised once by the caller (dcv_resident_ws_bytes): launch generation, error word, granules.  After a launch whose
 *               gather timed out (a block not resident: another grid holds the CUs) err != 0 and the logits are NaN: dcv_resident_status reads it.
 * ------------------------------------------------------------------------------------------------- */
#define DCV_RESIDENT_MAX_OPS 10
#define DCV_RES_LAYER 1
#define DCV_RES_LINK 2
#define DCV_RES_FC 3
typedef struct dcv_resident_op {
    int kind;
    int cin, cout, k, h, w;            /* LAYER: convolution geometry (h, w: input = output size); LINK: cin, h, w of the current tensor; FC: cin, h, w of its input, cout = out features */
    int act; float slope;              /* LAYER, FC */
    int bn, groups, pool;              /* LAYER: BatchNorm present, GroupNorm groups (0: none), average pooling window (0: none) */
    float momentum, bn_eps, gn_eps;
    int x_nchw;                        /* first op: the model input is NCHW (else NHWC) */
    int reserved;
    double inv_count, unbias, inv_group;   /* LAYER, filled by the caller: 1 / (n h w), n h w / (n h w - 1) (1 when n h w == 1), 1 / (h w cout / groups): the kernel
                                          multiplies by them instead of dividing in fp64 on a few lanes */
    int ref_c, ref_h, ref_w, align_corners;   /* LINK */
    const float* x;                    /* first op: the model input */
    const float* filt; const float* bias;
    const float* bn_gamma; const float* bn_beta; float* running_mean; float* running_var; int64_t* num_batches_tracked;
    const float* gn_gamma; const float* gn_beta;
    float* y;                          /* LAYER: [n][h][w][cout] activated convolution output (saved for the backward pass) */
    float* out;                        /* LAYER: layer output [n][h/pool][w/pool][cout]; LINK: the wide tensor; FC: logits [n][cout] */
    float* bn_mean; float* bn_rstd; float* gn_mean; float* gn_rstd;   /* saved statistics, as dcv_norm_chain_desc */
    const float* ref;                  /* LINK: [n][ref_h][ref_w][ref_c] */
    float* sample_sums;                /* LAYER, nullable: [n][2][cout] sums of y and y^2 per (sample, channel), as the layer's statistics epilogue computes them; handed to
                                          dcv_resident_backward (dcv_resident_bwd_op.sample_sums) they spare its moments phase two of its four reductions */
} dcv_resident_op;
/* `mailbox` (nullable) folds the batch staging of a replayed training step (dcv_indirect_copy) into the launch: the first op's input is read from
 * mailbox->src[x_job] — the address the host wrote for THIS replay — instead of op[0].x, and also stored to op[0].x (= mailbox->dst[x_job], the static tensor
 * the backward pass reads); the mailbox's other jobs (labels, optimizer scalars) are copied by one block each while the first layer's batch sums travel (that layer
 * must have BatchNorm).  Nothing of the launch's own outputs may alias them. */
typedef struct dcv_resident_program {
    int n; int count; dcv_resident_op op[DCV_RESIDENT_MAX_OPS];
    const dcv_copy_jobs* mailbox; int x_job; int reserved;
    const float* batch;                /* nullable; with `mailbox`: the batch's address when the host knows it at launch time (a re-issued launch: no read of
                                          mailbox->src[x_job] over PCIe on every block's critical path) */
} dcv_resident_program;
int dcv_resident_supported(const dcv_resident_program* p);           /* 1 when the resident kernel covers every op of the program */
size_t dcv_resident_ws_bytes(void);
int dcv_resident_forward(const dcv_resident_program* p, void* ws, size_t ws_bytes, void* stream);
/* host read of the error word of `ws` (synchronises the stream): 0, or 1 + index of the op whose gather timed out */
int dcv_resident_status(const void* ws, void* stream, int* err);

/* The serial chain of the BACKWARD pass of the same layer stack as a second persistent launch (same block / sample mapping; its blocks use ~110 KB of LDS:
 * one per CU): per layer, last to first, the normalisation-chain backward of dcv_norm_chain_bwd (four moments per (sample, channel), the sample's
 * GroupNorm algebra, the BatchNorm batch terms through ONE gather of the blocks' partial sums, dz = act'(y) (P dout + Q u + R)) and the data gradient
 * (the forward convolution kernels on the zero-padded dz tile with the transposed + flipped filter, op 4 of dcv_conv2d_prepare_filters), whose result is
 * the next layer's dout and never leaves LDS.  replaces autograd's walk over the `layer()` stack (src/deepcv/meta/ignite_training.py:253 `loss.backward()`).
 * Written to HBM: dz of every layer (what the weight-gradient kernels read, dcv_conv2d_wgrad_batch), d_bn_gamma / d_bn_beta (block 0, from the gathered
 * totals), per-block partial sums of the GroupNorm parameter gradients (`gn_part`, [blocks][2][cout]: d_gn_beta, d_gn_gamma = their sums over the
 * blocks, dcv_slab_reduce) and, if asked for, dx of the first layer.
 *   ops          layers in BACKWARD order; op 0 reads its output gradient `dout` from HBM ([n][h/pool][w/pool][cout]), the others from the previous op's
 *                data gradient; `addend` (nullable, [n][h][w][cin]) is added to an op's data gradient (the gradient of a dense link)
 *   ws           its own persistent, zero-initialised workspace of dcv_resident_ws_bytes() */
typedef struct dcv_resident_bwd_op {
    int cin, cout, k, h, w;            /* forward geometry of the layer */
    int act; float slope;
    int groups, pool;                  /* GroupNorm groups (0: none), average pooling window behind the layer (0: none) */
    int need_dx;                       /* compute the data gradient (0 for the first layer of the model) */
    double inv_count, inv_group;       /* 1 / (n h w), 1 / (h w cout / groups) */
    const float* y;                    /* saved activated convolution output */
    const float* dout; const float* addend;
    const float* filt;                 /* data-gradient filter (op 4); unused when !need_dx */
    const float* bn_gamma; const float* bn_beta; const float* bn_mean; const float* bn_rstd;
    const float* gn_gamma; const float* gn_mean; const float* gn_rstd;
    float* dz; float* dx;              /* dx: nullable (HBM copy of the last op's data gradient) */
    float* d_bn_gamma; float* d_bn_beta; float* gn_part;
    /* weight gradient inside the launch (nullable: the caller then runs dcv_conv2d_wgrad on dz): x = the layer's input ([n][h][w][cin] NHWC, or NCHW when
     * x_nchw), dw_part [2 * blocks][k * k][cout][cin] and db_part [2 * blocks][cout] (nullable) = per-(block, sample slot) partial sums, tap-major;
     * dw (OIHW) / db = their sums over the 2 * blocks chunks (dcv_slab_reduce with tr_a = k * k, tr_b = cout * cin).  Computed while the block waits for the next layer's all-reduce. */
    const float* x; float* dw_part; float* db_part; int x_nchw; int reserved;
    /* nullable: the forward launch's [n][2][cout] sums of y, y^2 per (sample, channel) (dcv_resident_op.sample_sums): sum u and sum u^2 of the GroupNorm algebra
     * (u = (y - mean) rstd) then come from them in fp64 instead of from two more reductions over the sample */
    const float* sample_sums;
} dcv_resident_bwd_op;
#define DCV_RESIDENT_BWD_MAX_OPS 8
/* Optional head of the backward pass in front of op 0, for a stack which ends LINK -> Flatten -> fully connected layer (dcv_resident_program): the data gradient
 * of the Linear (dwide = (dlogits * act'(logits)) W, W = the [O][H*W*C] weight of dcv_conv2d_prepare_filters op 2) and the link's split: the current tensor's
 * channels become op 0's dout (in LDS), the referenced tensor's channels go back through the bilinear resize (gather form) to `dref` in HBM, which the caller
 * names as `addend` of the op whose input the link referenced.  replaces the autograd nodes of submodule_creators.py:268-269 (Linear) and :272-332 (dense link).
 * The weight / bias gradient of the Linear: with fc_dw set (fc_x = the head's input [n][h][w][cc + cr], fc_dw [classes][(cc + cr) * h * w] in the reference's
 * C-major feature order, fc_db [classes] nullable) every block computes the final gradient rows of its share of the input positions for all samples while
 * the first all-reduce travels (no partial slabs; dcv_resident_head_wgrad_supported(n, k, classes): 8 positions per block, i.e. n >= k / 4, k % 8 == 0, classes <= 12);
 * with fc_dw NULL it stays with dcv_linear_bwd_deferred (dx = NULL). */
typedef struct dcv_resident_bwd_head {
    int enabled, classes, act; float slope;
    int h, w, cc, cr, ref_h, ref_w, align_corners, reserved;   /* wide tensor [n][h][w][cc + cr]; referenced tensor [n][ref_h][ref_w][cr] */
    const float* dlogits; const float* logits; const float* wp; float* dref;
    const float* fc_x; float* fc_dw; float* fc_db;
    /* the softmax cross-entropy of the step inside the launch (nullable ce_target: int64 class indices [n]): `logits` (= the head's output, which the
     * loss took as its input) are the loss's logits, dlogits is then NOT read: every block derives (softmax - onehot) / n for all samples itself (and the
     * loss: the same bits in every block), one block writes ce_loss [1], ce_lse [n] (nullable), ce_correct [1] (nullable) and — pub_seq non-NULL —
     * publishes the loss like dcv_cross_entropy_fwd_grad_publish.  n <= 512, classes <= 12.  replaces the dcv_cross_entropy_fwd_grad launch of a
     * training step whose loss is the root of the backward pass (ignite_training.py:253). */
    const long long* ce_target; float* ce_loss; float* ce_lse; int* ce_correct;
    const dcv_copy_jobs* pub_mailbox; float* pub_vals; int* pub_seq; int pub_slots; int reserved2;
} dcv_resident_bwd_head;
typedef struct dcv_resident_bwd_program { int n; int count; dcv_resident_bwd_op op[DCV_RESIDENT_BWD_MAX_OPS]; dcv_resident_bwd_head head; } dcv_resident_bwd_program;
int dcv_resident_bwd_supported(const dcv_resident_bwd_program* p);
int dcv_resident_head_wgrad_supported(int n, int k, int classes);
/* 1 when dcv_resident_backward computes the weight gradient of a (cin -> cout, k x k, stride 1, padding k / 2) layer on h x w maps itself (dw_part may be set for it) */
int dcv_resident_wgrad_supported(int cin, int cout, int k, int h, int w);
int dcv_resident_backward(const dcv_resident_bwd_program* p, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPCV_HIP_H */
