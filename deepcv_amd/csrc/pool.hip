// pool.hip — 2-D average / max pooling forward + backward on NHWC fp32 activations.
//
// Roofline: HBM.  Algorithmic bytes for 2x2/2: fwd 1.25 * 4 B * N*C*H*W, bwd 1.25 * 4 B * N*C*H*W
// (+ 4 B per output element of argmax for max pooling).  One thread per (pixel, 4-channel vector): 16-B coalesced accesses.
// The backward is a gather (each input pixel visits the windows that cover it), so it is deterministic without atomics.
//
// Replaces torch.nn.AvgPool2d (reference submodule_creators.py:168 -> nn.py:416) and torch.nn.MaxPool2d (reachable through nn_spec.py:158-169).
#include "common.h"
#include <float.h>

namespace {

// element types: float, or bf16 (the bf16 STORAGE path of the wide specs: activations and their gradients are bf16 tensors; arithmetic stays fp32, so the result
// is exactly the fp32 kernel's rounded once — what the widen / pool / narrow sequence it replaces produced, without the two cast launches and half the bytes)
typedef unsigned short pl_u16;
__device__ __forceinline__ float pl_ld(const float* p) { return *p; }
__device__ __forceinline__ float pl_ld(const pl_u16* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void pl_ld4(const float* p, float (&v)[4]) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
__device__ __forceinline__ void pl_ld4(const pl_u16* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void pl_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void pl_st(pl_u16* p, float v) { const __bf16 h = (__bf16)v; *p = __builtin_bit_cast(pl_u16, h); }
// four consecutive channels with ONE store (round 3: the kernels wrote them as four 2-byte / 4-byte stores and read the max indices one by one; the ResNet stem's
// max-pool backward took 191 us for 77 MB)
__device__ __forceinline__ void pl_st4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void pl_st4(pl_u16* p, const float (&v)[4]) {
    typedef __bf16 pl_bf4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<pl_bf4*>(p) = (pl_bf4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

// eight consecutive bf16 channels per thread (round 4: 16-byte accesses and half the index arithmetic per byte — the 3x3 / 2 max pooling behind the ResNet stem moves
// 0.8 GB tensors: 424 + 741 us of the bs 512 step with four channels per thread)
__device__ __forceinline__ void pl_ld8(const pl_u16* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u); v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
}
__device__ __forceinline__ void pl_ld8(const float* p, float (&v)[8]) { for (int k = 0; k < 8; ++k) v[k] = p[k]; }
__device__ __forceinline__ void pl_st8(pl_u16* p, const float (&v)[8]) {
    typedef __bf16 pl_bf8 __attribute__((ext_vector_type(8)));
    *reinterpret_cast<pl_bf8*>(p) = (pl_bf8){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
}
__device__ __forceinline__ void pl_st8(float* p, const float (&v)[8]) { for (int k = 0; k < 8; ++k) p[k] = v[k]; }

// IDX: unsigned (tensors below 2^32 elements: a 64-bit division is ~100 VALU instructions, and the index decode was most of these kernels' time) or size_t
template <int VEC, int MODE, typename T, typename IDX>
__global__ __launch_bounds__(256) void pool_fwd_kernel(dcv_pool2d_desc d, const T* __restrict__ x, T* __restrict__ y, int32_t* __restrict__ argmax) {
    const IDX CV = (IDX)(d.c / VEC);
    const IDX total = (IDX)d.n * (IDX)d.ho * (IDX)d.wo * CV;
    const float inv = 1.f / (float)(d.kh * d.kw);                        // count_include_pad=True
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * VEC; IDX p = i / CV;
        const int wo = (int)(p % (IDX)d.wo); p /= (IDX)d.wo; const int ho = (int)(p % (IDX)d.ho); const int n = (int)(p / (IDX)d.ho);
        const T* xs = x + (size_t)n * d.h * d.w * d.c;
        float acc[VEC]; int32_t arg[VEC];
        const int h0 = ho * d.stride_h - d.pad_h, w0 = wo * d.stride_w - d.pad_w;
        const int32_t first = (h0 < 0 ? 0 : h0) * d.w + (w0 < 0 ? 0 : w0);   // torch initialises maxindex to the window's first in-range element
#pragma unroll
        for (int k = 0; k < VEC; ++k) { acc[k] = (MODE == DCV_POOL_MAX) ? -INFINITY : 0.f; arg[k] = first; }
        for (int r = 0; r < d.kh; ++r) {
            const int h = h0 + r;
            if (h < 0 || h >= d.h) continue;
            for (int s = 0; s < d.kw; ++s) {
                const int w = w0 + s;
                if (w < 0 || w >= d.w) continue;
                float v[VEC];
                if constexpr (VEC == 8) pl_ld8(xs + ((size_t)h * d.w + w) * d.c + c, v);
                else if constexpr (VEC == 4) pl_ld4(xs + ((size_t)h * d.w + w) * d.c + c, v);
                else v[0] = pl_ld(xs + ((size_t)h * d.w + w) * d.c + c);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    if (MODE == DCV_POOL_MAX) {
                        // torch semantics: strict '>' (first maximum wins), a NaN always replaces the running value
                        if (v[k] > acc[k] || v[k] != v[k]) { acc[k] = v[k]; arg[k] = h * d.w + w; }
                    } else acc[k] += v[k];
                }
            }
        }
        const size_t o = (((size_t)n * d.ho + ho) * d.wo + wo) * d.c + c;
        if constexpr (VEC == 8) {
            if (MODE != DCV_POOL_MAX) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] *= inv;
            }
            pl_st8(y + o, acc);
            if (MODE == DCV_POOL_MAX) { *reinterpret_cast<int4*>(argmax + o) = make_int4(arg[0], arg[1], arg[2], arg[3]); *reinterpret_cast<int4*>(argmax + o + 4) = make_int4(arg[4], arg[5], arg[6], arg[7]); }
        } else if constexpr (VEC == 4) {
            if (MODE != DCV_POOL_MAX) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] *= inv;
            }
            pl_st4(y + o, acc);
            if (MODE == DCV_POOL_MAX) *reinterpret_cast<int4*>(argmax + o) = make_int4(arg[0], arg[1], arg[2], arg[3]);
        } else {
            pl_st(y + o, (MODE == DCV_POOL_MAX) ? acc[0] : acc[0] * inv);
            if (MODE == DCV_POOL_MAX) argmax[o] = arg[0];
        }
    }
}

__device__ __forceinline__ void window_range(int i, int pad, int k, int stride, int out_size, int& lo, int& hi) {
    // windows o with o*stride - pad <= i < o*stride - pad + k
    const int t = i + pad;
    hi = t / stride; if (hi > out_size - 1) hi = out_size - 1;
    const int u = t - k + 1;
    lo = u <= 0 ? 0 : (u + stride - 1) / stride;
}

template <int VEC, int MODE, typename T, typename IDX>
__global__ __launch_bounds__(256) void pool_bwd_kernel(dcv_pool2d_desc d, const T* __restrict__ dy, const int32_t* __restrict__ argmax, T* __restrict__ dx) {
    const IDX CV = (IDX)(d.c / VEC);
    const IDX total = (IDX)d.n * (IDX)d.h * (IDX)d.w * CV;
    const float inv = 1.f / (float)(d.kh * d.kw);
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * VEC; IDX p = i / CV;
        const int w = (int)(p % (IDX)d.w); p /= (IDX)d.w; const int h = (int)(p % (IDX)d.h); const int n = (int)(p / (IDX)d.h);
        int holo, hohi, wolo, wohi;
        window_range(h, d.pad_h, d.kh, d.stride_h, d.ho, holo, hohi);
        window_range(w, d.pad_w, d.kw, d.stride_w, d.wo, wolo, wohi);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        const int me = h * d.w + w;
        for (int ho = holo; ho <= hohi; ++ho)
            for (int wo = wolo; wo <= wohi; ++wo) {
                const size_t o = (((size_t)n * d.ho + ho) * d.wo + wo) * d.c + c;
                if constexpr (VEC == 8) {
                    float g[8]; pl_ld8(dy + o, g);
                    if (MODE == DCV_POOL_MAX) {
                        const int4 am = *reinterpret_cast<const int4*>(argmax + o), an = *reinterpret_cast<const int4*>(argmax + o + 4);
                        acc[0] += am.x == me ? g[0] : 0.f; acc[1] += am.y == me ? g[1] : 0.f; acc[2] += am.z == me ? g[2] : 0.f; acc[3] += am.w == me ? g[3] : 0.f;
                        acc[4] += an.x == me ? g[4] : 0.f; acc[5] += an.y == me ? g[5] : 0.f; acc[6] += an.z == me ? g[6] : 0.f; acc[7] += an.w == me ? g[7] : 0.f;
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[k] += g[k] * inv;
                    }
                } else if constexpr (VEC == 4) {
                    float g[4]; pl_ld4(dy + o, g);
                    if (MODE == DCV_POOL_MAX) {
                        const int4 am = *reinterpret_cast<const int4*>(argmax + o);
                        acc[0] += am.x == me ? g[0] : 0.f; acc[1] += am.y == me ? g[1] : 0.f; acc[2] += am.z == me ? g[2] : 0.f; acc[3] += am.w == me ? g[3] : 0.f;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] += g[k] * inv;
                    }
                } else {
                    if (MODE == DCV_POOL_MAX) { if (argmax[o] == me) acc[0] += pl_ld(dy + o); }
                    else acc[0] += pl_ld(dy + o) * inv;
                }
            }
        const size_t xi = (((size_t)n * d.h + h) * d.w + w) * d.c + c;
        if constexpr (VEC == 8) pl_st8(dx + xi, acc); else if constexpr (VEC == 4) pl_st4(dx + xi, acc); else pl_st(dx + xi, acc[0]);
    }
}

// Backward of a NON-OVERLAPPING pooling (kernel == stride, no padding, sizes divisible: every average pooling of the BASELINE specs): a thread owns one OUTPUT
// element vector — one index decode, one read of dy (and of the max indices) — and writes the kh x kw input pixels of its window.  The gather kernel above decodes
// every INPUT pixel (three 32-bit divisions + two window ranges with divisions by the stride: ~200 instructions for an 8-byte store; 23 us for 47 MB = 2 TB/s at
// cifar100_deep bs 512, PMC).
template <int MODE, typename T>
__global__ __launch_bounds__(256) void pool_bwd_tile_kernel(dcv_pool2d_desc d, const T* __restrict__ dy, const int32_t* __restrict__ argmax, T* __restrict__ dx) {
    const unsigned CV = (unsigned)(d.c / 4);
    const unsigned total = (unsigned)d.n * (unsigned)d.ho * (unsigned)d.wo * CV;
    const float inv = 1.f / (float)(d.kh * d.kw);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * 4; unsigned p = i / CV;
        const int wo = (int)(p % (unsigned)d.wo); p /= (unsigned)d.wo; const int ho = (int)(p % (unsigned)d.ho); const int n = (int)(p / (unsigned)d.ho);
        float g[4]; pl_ld4(dy + (size_t)i * 4, g);
        int4 am = make_int4(0, 0, 0, 0);
        if (MODE == DCV_POOL_MAX) am = *reinterpret_cast<const int4*>(argmax + (size_t)i * 4);
        T* xs = dx + (size_t)n * d.h * d.w * d.c + c;
        for (int r = 0; r < d.kh; ++r)
            for (int s = 0; s < d.kw; ++s) {
                const int h = ho * d.kh + r, w = wo * d.kw + s, me = h * d.w + w;
                float o[4];
                if (MODE == DCV_POOL_MAX) { o[0] = am.x == me ? g[0] : 0.f; o[1] = am.y == me ? g[1] : 0.f; o[2] = am.z == me ? g[2] : 0.f; o[3] = am.w == me ? g[3] : 0.f; }
                else { o[0] = g[0] * inv; o[1] = g[1] * inv; o[2] = g[2] * inv; o[3] = g[3] * inv; }
                pl_st4(xs + (size_t)me * d.c, o);
            }
    }
}

inline bool pool_desc_ok(const dcv_pool2d_desc* d) {
    if (!d || d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0) return false;
    if (d->pad_h * 2 > d->kh || d->pad_w * 2 > d->kw) return false;     // same restriction as torch
    const int ho = (d->h + 2 * d->pad_h - d->kh) / d->stride_h + 1, wo = (d->w + 2 * d->pad_w - d->kw) / d->stride_w + 1;
    return ho == d->ho && wo == d->wo && ho > 0 && wo > 0 && (d->mode == DCV_POOL_AVG || d->mode == DCV_POOL_MAX);
}

template <typename T>
int pool_fwd_launch(const dcv_pool2d_desc* d, const T* x, T* y, int32_t* argmax, hipStream_t st) {
    // (four channels per thread need the vectors' alignment: 4 elements of T, 16 B for the max indices)
    int vec = (d->c % 4 == 0 && reinterpret_cast<uintptr_t>(x) % (4 * sizeof(T)) == 0 && reinterpret_cast<uintptr_t>(y) % (4 * sizeof(T)) == 0 && reinterpret_cast<uintptr_t>(argmax) % 16 == 0) ? 4 : 1;
    if (vec == 4 && sizeof(T) == 2 && d->c % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0) vec = 8;
    const size_t items = (size_t)d->n * d->ho * d->wo * (d->c / vec);
    const unsigned grid = dcv_stream_grid(items, 256);
    const bool small = (size_t)d->n * d->h * d->w * d->c < ((size_t)1 << 31);      // every element index fits 32 bits
#define DCV_POOL_FWD(V, M) do { if (small) hipLaunchKernelGGL((pool_fwd_kernel<V, M, T, unsigned>), dim3(grid), dim3(256), 0, st, *d, x, y, argmax); \
                                else hipLaunchKernelGGL((pool_fwd_kernel<V, M, T, size_t>), dim3(grid), dim3(256), 0, st, *d, x, y, argmax); } while (0)
    if (d->mode == DCV_POOL_AVG) { if (vec == 8) DCV_POOL_FWD(8, DCV_POOL_AVG); else if (vec == 4) DCV_POOL_FWD(4, DCV_POOL_AVG); else DCV_POOL_FWD(1, DCV_POOL_AVG); }
    else { if (vec == 8) DCV_POOL_FWD(8, DCV_POOL_MAX); else if (vec == 4) DCV_POOL_FWD(4, DCV_POOL_MAX); else DCV_POOL_FWD(1, DCV_POOL_MAX); }
#undef DCV_POOL_FWD
    DCV_RETURN_LAUNCH();
}

template <typename T>
int pool_bwd_launch(const dcv_pool2d_desc* d, const T* dy, const int32_t* argmax, T* dx, hipStream_t st) {
    int vec = (d->c % 4 == 0 && reinterpret_cast<uintptr_t>(dy) % (4 * sizeof(T)) == 0 && reinterpret_cast<uintptr_t>(dx) % (4 * sizeof(T)) == 0 && reinterpret_cast<uintptr_t>(argmax) % 16 == 0) ? 4 : 1;
    const bool tile4 = vec == 4 && d->kh == d->stride_h && d->kw == d->stride_w && d->pad_h == 0 && d->pad_w == 0 && d->h == d->ho * d->kh && d->w == d->wo * d->kw;
    if (vec == 4 && !tile4 && sizeof(T) == 2 && d->c % 8 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0 && reinterpret_cast<uintptr_t>(dx) % 16 == 0) vec = 8;
    const unsigned grid = dcv_stream_grid((size_t)d->n * d->h * d->w * (d->c / vec), 256);
    const bool small = (size_t)d->n * d->h * d->w * d->c < ((size_t)1 << 31);
    if (vec == 4 && small && d->kh == d->stride_h && d->kw == d->stride_w && d->pad_h == 0 && d->pad_w == 0 && d->h == d->ho * d->kh && d->w == d->wo * d->kw) {
        const unsigned tgrid = dcv_stream_grid((size_t)d->n * d->ho * d->wo * (d->c / 4), 256);
        if (d->mode == DCV_POOL_AVG) hipLaunchKernelGGL((pool_bwd_tile_kernel<DCV_POOL_AVG, T>), dim3(tgrid), dim3(256), 0, st, *d, dy, argmax, dx);
        else hipLaunchKernelGGL((pool_bwd_tile_kernel<DCV_POOL_MAX, T>), dim3(tgrid), dim3(256), 0, st, *d, dy, argmax, dx);
        DCV_RETURN_LAUNCH();
    }
#define DCV_POOL_BWD(V, M) do { if (small) hipLaunchKernelGGL((pool_bwd_kernel<V, M, T, unsigned>), dim3(grid), dim3(256), 0, st, *d, dy, argmax, dx); \
                                else hipLaunchKernelGGL((pool_bwd_kernel<V, M, T, size_t>), dim3(grid), dim3(256), 0, st, *d, dy, argmax, dx); } while (0)
    if (d->mode == DCV_POOL_AVG) { if (vec == 8) DCV_POOL_BWD(8, DCV_POOL_AVG); else if (vec == 4) DCV_POOL_BWD(4, DCV_POOL_AVG); else DCV_POOL_BWD(1, DCV_POOL_AVG); }
    else { if (vec == 8) DCV_POOL_BWD(8, DCV_POOL_MAX); else if (vec == 4) DCV_POOL_BWD(4, DCV_POOL_MAX); else DCV_POOL_BWD(1, DCV_POOL_MAX); }
#undef DCV_POOL_BWD
    DCV_RETURN_LAUNCH();
}

}  // namespace

extern "C" {

int dcv_pool2d_fwd(const dcv_pool2d_desc* d, const float* x, float* y, int32_t* argmax, void* stream) {
    DCV_CHECK_ARG(pool_desc_ok(d) && x && y && (d->mode == DCV_POOL_AVG || argmax));
    return pool_fwd_launch<float>(d, x, y, argmax, dcv_stream(stream));
}

int dcv_pool2d_bwd(const dcv_pool2d_desc* d, const float* dy, const int32_t* argmax, float* dx, void* stream) {
    DCV_CHECK_ARG(pool_desc_ok(d) && dy && dx && (d->mode == DCV_POOL_AVG || argmax));
    return pool_bwd_launch<float>(d, dy, argmax, dx, dcv_stream(stream));
}

int dcv_pool2d_fwd_bf16(const dcv_pool2d_desc* d, const void* x, void* y, int32_t* argmax, void* stream) {
    DCV_CHECK_ARG(pool_desc_ok(d) && x && y && (d->mode == DCV_POOL_AVG || argmax) && (reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0);
    return pool_fwd_launch<pl_u16>(d, reinterpret_cast<const pl_u16*>(x), reinterpret_cast<pl_u16*>(y), argmax, dcv_stream(stream));
}

int dcv_pool2d_bwd_bf16(const dcv_pool2d_desc* d, const void* dy, const int32_t* argmax, void* dx, void* stream) {
    DCV_CHECK_ARG(pool_desc_ok(d) && dy && dx && (d->mode == DCV_POOL_AVG || argmax) && (reinterpret_cast<uintptr_t>(dy) & 7) == 0 && (reinterpret_cast<uintptr_t>(dx) & 7) == 0);
    return pool_bwd_launch<pl_u16>(d, reinterpret_cast<const pl_u16*>(dy), argmax, reinterpret_cast<pl_u16*>(dx), dcv_stream(stream));
}

}  // extern "C"
