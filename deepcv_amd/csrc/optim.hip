// optim.hip — fused multi-tensor AdamW / SGD steps and the data-parallel gradient-bucket pack.
//
// Roofline: HBM.  Algorithmic bytes per parameter: AdamW 28 B (read p,g,m,v; write p,m,v), SGD-momentum 20 B.
// One launch updates up to DCV_MAX_TENSORS parameter tensors: the tensor table travels by value in the kernel arguments (no
// H2D copy), grid = one row of 1024-element blocks, tensor after tensor (`list_block`).  Step-dependent scalars (lr, momentum, bias corrections) are read
// from DEVICE memory so that a captured hipGraph replays with the scheduler's current values.
//
// Replaces torch.optim.AdamW (reference classification/image.py:71; options conf/base/parameters.yml:129-134), torch.optim.SGD
// (parameters.yml:136) as called at ignite_training.py:254, and DDP's gradient bucketing (ignite_training.py:380).
#include "common.h"
#include "optim_update.h"

namespace {

#define OPT_BLOCK 256
#define OPT_ELEMS_PER_BLOCK (OPT_BLOCK * 4)

__global__ void optim_prepare_kernel(float* __restrict__ hyper, int64_t* __restrict__ step) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t s = step[0] + 1;
    step[0] = s;
    hyper[DCV_HYP_BIAS_CORR1] = (float)(1.0 - pow((double)hyper[DCV_HYP_BETA1], (double)s));
    hyper[DCV_HYP_BIAS_CORR2] = (float)(1.0 - pow((double)hyper[DCV_HYP_BETA2], (double)s));
    hyper[DCV_HYP_FIRST_STEP] = (s == 1) ? 1.f : 0.f;
}

// Block -> (tensor, first element): the grid is ONE row of blocks, tensor after tensor (round 4, last session: grid = (blocks of the largest tensor, tensors) launched
// 92 000 blocks for the 40 tensors of cifar100_deep, of which 4 800 had work — dispatching the empty ones was a third of the kernel).  A scalar walk over the
// table's element counts (at most DCV_MAX_TENSORS steps).
template <int EPB = OPT_ELEMS_PER_BLOCK>
__device__ __forceinline__ bool list_block(const dcv_tensor_list& tl, int& t, int64_t& base) {
    int64_t b = blockIdx.x;
    for (int k = 0; k < tl.count; ++k) {
        const int64_t nb = (tl.numel[k] + EPB - 1) / EPB;
        if (b < nb) { t = k; base = b * EPB; return true; }
        b -= nb;
    }
    return false;
}

// (the update itself: optim_update.h, shared with the update + filter re-layout kernel of conv2d.hip)
template <bool AMSGRAD>
__global__ __launch_bounds__(OPT_BLOCK) void adamw_kernel(dcv_tensor_list tl, const float* __restrict__ hyper) {
    int t; int64_t base;
    if (!list_block(tl, t, base)) return;
    const int64_t n = tl.numel[t];
    const dcv_adamw_scalars sc = dcv_adamw_load_scalars(hyper);
    float* __restrict__ p = tl.param[t]; const float* __restrict__ g = tl.grad[t];
    float* __restrict__ m = tl.state1[t]; float* __restrict__ v = tl.state2[t]; float* __restrict__ vmax = tl.state3[t];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + (int64_t)k * OPT_BLOCK + threadIdx.x;
        if (i < n) dcv_adamw_update<AMSGRAD>(sc, p, g, m, v, vmax, i);
    }
}

// torch.optim.SGD semantics: g += wd*p ; buf = (first step) g : mu*buf + (1-dampening)*g ; g = nesterov ? g + mu*buf : buf ; p -= lr*g
// (one element's update, shared by the 16-byte and the scalar path of the kernel below: the same expression tree, the same bits)
template <bool NESTEROV>
__device__ __forceinline__ void sgd_one(float& pi, float gi, float& bi, bool has_buf, bool first, float lr, float mu, float damp, float wd, float gs) {
    gi = gi * gs + wd * pi;
    if (has_buf) {
        bi = first ? gi : mu * bi + (1.f - damp) * gi;
        gi = NESTEROV ? gi + mu * bi : bi;
    }
    pi = pi - lr * gi;
}

// A thread owns SGD_VECS groups of four CONSECUTIVE elements, a block 4096 elements: 16-byte loads / stores when the tensor's pointers allow it (round 4: the
// 4-byte accesses ran the 4.9 M parameters of cifar100_deep at 2.2 TB/s, 44 us per step; 16-byte accesses with ONE group per thread 41 us: 4 800 blocks of three
// loads per thread each — the last session of round 4 issues every load of the four groups before the first update, a quarter of the blocks).
#define SGD_VECS 4
#define SGD_ELEMS_PER_BLOCK (OPT_BLOCK * 4 * SGD_VECS)
template <bool NESTEROV>
__global__ __launch_bounds__(OPT_BLOCK) void sgd_kernel(dcv_tensor_list tl, const float* __restrict__ hyper) {
    int t; int64_t base;
    if (!list_block<SGD_ELEMS_PER_BLOCK>(tl, t, base)) return;
    const int64_t n = tl.numel[t];
    float* __restrict__ p = tl.param[t]; const float* __restrict__ g = tl.grad[t]; float* __restrict__ buf = tl.state1[t];
    const bool has_buf = buf != nullptr;
    const bool first = hyper[DCV_HYP_FIRST_STEP] != 0.f;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(buf)) & 15) == 0;
    const bool rd_buf = has_buf && !first;
    float4 pv[SGD_VECS], gv[SGD_VECS], bv[SGD_VECS];
    bool vec[SGD_VECS];
#pragma unroll
    for (int k = 0; k < SGD_VECS; ++k) {
        const int64_t i0 = base + 4 * ((int64_t)k * OPT_BLOCK + threadIdx.x);
        vec[k] = aligned && i0 + 3 < n;
        bv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec[k]) {
            pv[k] = *reinterpret_cast<const float4*>(p + i0);
            gv[k] = *reinterpret_cast<const float4*>(g + i0);
            if (rd_buf) bv[k] = *reinterpret_cast<const float4*>(buf + i0);
        }
    }
    const float lr = hyper[DCV_HYP_LR], mu = hyper[DCV_HYP_BETA1], damp = hyper[DCV_HYP_BETA2];
    const float wd = hyper[DCV_HYP_WEIGHT_DECAY], gs = hyper[DCV_HYP_GRAD_SCALE];
#pragma unroll
    for (int k = 0; k < SGD_VECS; ++k) {
        const int64_t i0 = base + 4 * ((int64_t)k * OPT_BLOCK + threadIdx.x);
        if (vec[k]) {
            sgd_one<NESTEROV>(pv[k].x, gv[k].x, bv[k].x, has_buf, first, lr, mu, damp, wd, gs);
            sgd_one<NESTEROV>(pv[k].y, gv[k].y, bv[k].y, has_buf, first, lr, mu, damp, wd, gs);
            sgd_one<NESTEROV>(pv[k].z, gv[k].z, bv[k].z, has_buf, first, lr, mu, damp, wd, gs);
            sgd_one<NESTEROV>(pv[k].w, gv[k].w, bv[k].w, has_buf, first, lr, mu, damp, wd, gs);
            if (has_buf) *reinterpret_cast<float4*>(buf + i0) = bv[k];
            *reinterpret_cast<float4*>(p + i0) = pv[k];
        } else {
            for (int64_t i = i0; i < i0 + 4 && i < n; ++i) {
                float pi = p[i], bi = rd_buf ? buf[i] : 0.f;
                sgd_one<NESTEROV>(pi, g[i], bi, has_buf, first, lr, mu, damp, wd, gs);
                if (has_buf) buf[i] = bi;
                p[i] = pi;
            }
        }
    }
}

__global__ __launch_bounds__(OPT_BLOCK) void pack_kernel(dcv_tensor_list tl, float* __restrict__ dst) {
    int t; int64_t base;
    if (!list_block(tl, t, base)) return;
    const int64_t n = tl.numel[t];
    float* __restrict__ out = tl.param[t];                        // explicit destination (bucket slices with alignment padding) ...
    if (out == nullptr) {                                         // ... or the tensors back to back in `dst`
        int64_t off = 0;
        for (int k = 0; k < t; ++k) off += tl.numel[k];
        out = dst + off;
    }
    const float* __restrict__ g = tl.grad[t];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + (int64_t)k * OPT_BLOCK + threadIdx.x;
        if (i < n) out[i] = g[i];
    }
}

inline int list_grid(const dcv_tensor_list* tl, dim3* grid, bool need_param, bool need_state1, bool need_state2, bool need_state3, int epb = OPT_ELEMS_PER_BLOCK) {
    if (!tl || tl->count <= 0 || tl->count > DCV_MAX_TENSORS) return DCV_ERR_INVALID_ARG;
    int64_t mx = 0;
    for (int i = 0; i < tl->count; ++i) {
        if (tl->numel[i] <= 0 || !tl->grad[i]) return DCV_ERR_INVALID_ARG;
        if (need_param && !tl->param[i]) return DCV_ERR_INVALID_ARG;
        if (need_state1 && !tl->state1[i]) return DCV_ERR_INVALID_ARG;
        if (need_state2 && !tl->state2[i]) return DCV_ERR_INVALID_ARG;
        if (need_state3 && !tl->state3[i]) return DCV_ERR_INVALID_ARG;
        mx += (tl->numel[i] + epb - 1) / epb;
    }
    if (mx > 0x7fffffffLL) return DCV_ERR_UNSUPPORTED;
    *grid = dim3((unsigned)mx, 1, 1);
    return DCV_OK;
}

}  // namespace

extern "C" {

int dcv_optim_prepare(float* hyper, int64_t* step, void* stream) {
    DCV_CHECK_ARG(hyper && step);
    hipLaunchKernelGGL(optim_prepare_kernel, dim3(1), dim3(64), 0, dcv_stream(stream), hyper, step);
    DCV_RETURN_LAUNCH();
}

int dcv_adamw_step(const dcv_tensor_list* tl, const float* hyper, int amsgrad, void* stream) {
    DCV_CHECK_ARG(hyper);
    dim3 grid;
    const int rc = list_grid(tl, &grid, true, true, true, amsgrad != 0);
    if (rc != DCV_OK) return rc;
    if (amsgrad) hipLaunchKernelGGL((adamw_kernel<true>), grid, dim3(OPT_BLOCK), 0, dcv_stream(stream), *tl, hyper);
    else hipLaunchKernelGGL((adamw_kernel<false>), grid, dim3(OPT_BLOCK), 0, dcv_stream(stream), *tl, hyper);
    DCV_RETURN_LAUNCH();
}

int dcv_sgd_step(const dcv_tensor_list* tl, const float* hyper, int nesterov, void* stream) {
    DCV_CHECK_ARG(hyper);
    dim3 grid;
    const int rc = list_grid(tl, &grid, true, false, false, false, SGD_ELEMS_PER_BLOCK);
    if (rc != DCV_OK) return rc;
    if (nesterov) hipLaunchKernelGGL((sgd_kernel<true>), grid, dim3(OPT_BLOCK), 0, dcv_stream(stream), *tl, hyper);
    else hipLaunchKernelGGL((sgd_kernel<false>), grid, dim3(OPT_BLOCK), 0, dcv_stream(stream), *tl, hyper);
    DCV_RETURN_LAUNCH();
}

int dcv_pack_grads(const dcv_tensor_list* tl, float* dst_flat, void* stream) {
    DCV_CHECK_ARG(dst_flat);
    dim3 grid;
    const int rc = list_grid(tl, &grid, false, false, false, false);
    if (rc != DCV_OK) return rc;
    hipLaunchKernelGGL(pack_kernel, grid, dim3(OPT_BLOCK), 0, dcv_stream(stream), *tl, dst_flat);
    DCV_RETURN_LAUNCH();
}

}  // extern "C"
