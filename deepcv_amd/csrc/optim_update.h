// optim_update.h — the per-element AdamW update, shared by the multi-tensor kernel (optim.hip) and the update + filter re-layout kernel (conv2d.hip): one
// expression tree, so both produce the same bits.
// torch.optim.AdamW single-tensor semantics:  p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
// denom = sqrt(v)/sqrt(bc2) + eps ; p -= (lr/bc1) * m / denom      (amsgrad: denom from max(v_max, v))
#ifndef DEEPCV_OPTIM_UPDATE_H
#define DEEPCV_OPTIM_UPDATE_H
#include <hip/hip_runtime.h>
#include "../../include/deepcv_hip.h"

struct dcv_adamw_scalars { float lr, b1, b2, eps, wd, gs, step_size, bc2_sqrt; };

static __device__ __forceinline__ dcv_adamw_scalars dcv_adamw_load_scalars(const float* __restrict__ hyper) {
    dcv_adamw_scalars s;
    s.lr = hyper[DCV_HYP_LR]; s.b1 = hyper[DCV_HYP_BETA1]; s.b2 = hyper[DCV_HYP_BETA2]; s.eps = hyper[DCV_HYP_EPS];
    s.wd = hyper[DCV_HYP_WEIGHT_DECAY]; s.gs = hyper[DCV_HYP_GRAD_SCALE];
    s.step_size = __fdiv_rn(s.lr, hyper[DCV_HYP_BIAS_CORR1]);
    s.bc2_sqrt = __fsqrt_rn(hyper[DCV_HYP_BIAS_CORR2]);
    return s;
}

// (the gradient as a VALUE and the element's parameter / moments as values requested earlier: the optimizer's tail launch — conv2d.hip, adamw_prep_slabs_kernel —
// updates an element in the thread which has just finished its slab sum, with the three loads already in flight behind the sum's)
template <bool AMSGRAD>
static __device__ __forceinline__ float dcv_adamw_update_loaded(const dcv_adamw_scalars& s, float* __restrict__ p, float g_raw, float* __restrict__ m,
                                                                float* __restrict__ v, float* __restrict__ vmax, long long i, float p0, float m0, float v0) {
    // every operation spelled out (no contraction left to the compiler): the kernels which share this function must agree to the bit
    const float gi = __fmul_rn(g_raw, s.gs);
    float pi = __fmul_rn(p0, __fsub_rn(1.f, __fmul_rn(s.lr, s.wd)));
    const float mi = __fmaf_rn(__fsub_rn(1.f, s.b1), __fsub_rn(gi, m0), m0);      // torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vi = __fmaf_rn(__fmul_rn(__fsub_rn(1.f, s.b2), gi), gi, __fmul_rn(s.b2, v0));
    m[i] = mi; v[i] = vi;
    float vden = vi;
    if (AMSGRAD) { vden = fmaxf(vmax[i], vi); vmax[i] = vden; }
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vden), s.bc2_sqrt), s.eps);
    pi = __fmaf_rn(-s.step_size, __fdiv_rn(mi, denom), pi);
    p[i] = pi;
    return pi;
}
template <bool AMSGRAD>
static __device__ __forceinline__ float dcv_adamw_update_value(const dcv_adamw_scalars& s, float* __restrict__ p, float g_raw, float* __restrict__ m,
                                                               float* __restrict__ v, float* __restrict__ vmax, long long i) {
    return dcv_adamw_update_loaded<AMSGRAD>(s, p, g_raw, m, v, vmax, i, p[i], m[i], v[i]);
}

template <bool AMSGRAD>
static __device__ __forceinline__ void dcv_adamw_update(const dcv_adamw_scalars& s, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ vmax, long long i) {
    dcv_adamw_update_value<AMSGRAD>(s, p, g[i], m, v, vmax, i);
}
#endif
