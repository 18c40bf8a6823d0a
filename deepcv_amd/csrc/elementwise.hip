// elementwise.hip — activation fwd/bwd (+ fused bias-gradient reduction), channel-slice copy (concat / split), add, scale,
// inner transpose (NHWC <-> NCHW), bilinear resize fwd/bwd.  All HBM-bound, 16-B vectorised where the shape allows.
//
// Replaces: `act_fn()` modules (reference src/deepcv/meta/nn.py:553), torch.cat (submodule_creators.py:56,323),
// F.interpolate (nn.py:675), torch.nn.Flatten on channels-last data (conf/base/parameters.yml:87).
#include "colreduce.h"
#include "slab_sum.h"
#include <string.h>

namespace {

template <int VEC>
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t nv, int act, float slope) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        if constexpr (VEC == 4) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            float4 o; o.x = dcv_act_apply(v.x, act, slope); o.y = dcv_act_apply(v.y, act, slope);
            o.z = dcv_act_apply(v.z, act, slope); o.w = dcv_act_apply(v.w, act, slope);
            reinterpret_cast<float4*>(y)[i] = o;
        } else y[i] = dcv_act_apply(x[i], act, slope);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                                       size_t nv, int act, float slope) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        if constexpr (VEC == 4) {
            const float4 d = reinterpret_cast<const float4*>(dy)[i]; const float4 v = reinterpret_cast<const float4*>(y)[i];
            float4 o; o.x = d.x * dcv_act_grad_from_out(v.x, act, slope); o.y = d.y * dcv_act_grad_from_out(v.y, act, slope);
            o.z = d.z * dcv_act_grad_from_out(v.z, act, slope); o.w = d.w * dcv_act_grad_from_out(v.w, act, slope);
            reinterpret_cast<float4*>(dx)[i] = o;
        } else dx[i] = dy[i] * dcv_act_grad_from_out(y[i], act, slope);
    }
}

// dx = dy * act'(y) written in the same pass that accumulates the per-channel sum of dx (bias gradient)
struct ActBwdLoader {
    const float* dy; const float* y; float* dx; int act; float slope;
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int, float (&q)[1][VEC]) const {
        if constexpr (VEC == 4) {
            const float4 d = *reinterpret_cast<const float4*>(dy + idx); const float4 v = *reinterpret_cast<const float4*>(y + idx);
            float4 o; o.x = d.x * dcv_act_grad_from_out(v.x, act, slope); o.y = d.y * dcv_act_grad_from_out(v.y, act, slope);
            o.z = d.z * dcv_act_grad_from_out(v.z, act, slope); o.w = d.w * dcv_act_grad_from_out(v.w, act, slope);
            *reinterpret_cast<float4*>(dx + idx) = o;
            q[0][0] = o.x; q[0][1] = o.y; q[0][2] = o.z; q[0][3] = o.w;
        } else {
            const float o = dy[idx] * dcv_act_grad_from_out(y[idx], act, slope);
            dx[idx] = o; q[0][0] = o;
        }
    }
};

template <int VEC>
__global__ __launch_bounds__(DCV_CR_THREADS) void act_bwd_bias_kernel(ActBwdLoader ld, size_t rows, int C, double* __restrict__ partial) {
    __shared__ float scratch[DCV_CR_THREADS * VEC];
    const size_t rpb = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * rpb;
    size_t r1 = r0 + rpb; if (r1 > rows) r1 = rows;
    dcv_colreduce_block<1, VEC>(ld, r0 < rows ? r0 : rows, r1, C, scratch, partial + (size_t)blockIdx.x * C);
}

// one wave per channel: lanes stride over the per-block partials, then a fixed-order butterfly -> deterministic and ~nb/64 serial loads per lane
__global__ __launch_bounds__(64) void sum_partials_kernel(const double* __restrict__ partial, int nb, int C, float* __restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int b = lane; b < nb; b += DCV_WAVE) s += partial[(size_t)b * C + c];
    s = dcv_wave_sum(s);
    if (lane == 0) out[c] = (float)s;
}

struct PlainLoader {
    const float* x;
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int, float (&q)[1][VEC]) const {
        if constexpr (VEC == 4) { const float4 v = *reinterpret_cast<const float4*>(x + idx); q[0][0] = v.x; q[0][1] = v.y; q[0][2] = v.z; q[0][3] = v.w; }
        else q[0][0] = x[idx];
    }
};

struct Bf16Loader {                // bf16 rows (the bf16 storage path's dz), widened on load
    const unsigned short* x;
    template <int VEC>
    __device__ __forceinline__ void load(size_t idx, int, float (&q)[1][VEC]) const {
        if constexpr (VEC == 4) {
            const uint2 v = *reinterpret_cast<const uint2*>(x + idx);
            q[0][0] = __uint_as_float(v.x << 16); q[0][1] = __uint_as_float(v.x & 0xffff0000u); q[0][2] = __uint_as_float(v.y << 16); q[0][3] = __uint_as_float(v.y & 0xffff0000u);
        } else q[0][0] = __uint_as_float((unsigned)x[idx] << 16);
    }
};

template <int VEC, typename Loader = PlainLoader>
__global__ __launch_bounds__(DCV_CR_THREADS) void channel_sum_kernel(Loader ld, size_t rows, int C, double* __restrict__ partial) {
    __shared__ float scratch[DCV_CR_THREADS * VEC];
    const size_t rpb = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * rpb;
    size_t r1 = r0 + rpb; if (r1 > rows) r1 = rows;
    dcv_colreduce_block<1, VEC>(ld, r0 < rows ? r0 : rows, r1, C, scratch, partial + (size_t)blockIdx.x * C);
}

template <int VEC>
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, int src_c, int src_off, float* __restrict__ dst, int dst_c, int dst_off,
                                                             int copy_c, size_t rows) {
    const int cv = copy_c / VEC;
    const size_t total = rows * (size_t)cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cv; const int c = (int)(i % cv) * VEC;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + r * dst_c + dst_off + c) = *reinterpret_cast<const float4*>(src + r * src_c + src_off + c);
        else dst[r * dst_c + dst_off + c] = src[r * src_c + src_off + c];
    }
}

// concat (to_wide = 1: parts -> wide rows) or split (to_wide = 0: wide rows -> parts) of up to DCV_MAX_CONCAT channel blocks in ONE launch; blockIdx.y = part
struct ConcatTable { float* part[DCV_MAX_CONCAT]; int c[DCV_MAX_CONCAT]; int off[DCV_MAX_CONCAT]; int count; };

template <int VEC>
__global__ __launch_bounds__(256) void concat_split_kernel(ConcatTable t, float* __restrict__ wide, int wide_c, size_t rows, int to_wide) {
    const int k = blockIdx.y;
    float* __restrict__ part = t.part[k];
    const int pc = t.c[k], off = t.off[k];
    const int cv = pc / VEC;
    const size_t total = rows * (size_t)cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cv; const int c = (int)(i % cv) * VEC;
        float* pw = wide + r * wide_c + off + c; float* pp = part + r * pc + c;
        if constexpr (VEC == 4) { if (to_wide) *reinterpret_cast<float4*>(pw) = *reinterpret_cast<const float4*>(pp); else *reinterpret_cast<float4*>(pp) = *reinterpret_cast<const float4*>(pw); }
        else { if (to_wide) *pw = *pp; else *pp = *pw; }
    }
}

// copies described by a mailbox read at run time (see dcv_indirect_copy); 16-B vectors when all three of src, dst, bytes allow it
__global__ __launch_bounds__(256) void indirect_copy_kernel(const dcv_copy_jobs* __restrict__ jobs) {
    const int count = jobs->count < DCV_MAX_COPY_JOBS ? jobs->count : DCV_MAX_COPY_JOBS;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gstride = (size_t)gridDim.x * blockDim.x;
    for (int j = 0; j < count; ++j) {
        const char* src = reinterpret_cast<const char*>(jobs->src[j]);
        char* dst = reinterpret_cast<char*>(jobs->dst[j]);
        const size_t bytes = (size_t)jobs->bytes[j];
        if (!src || !dst || bytes == 0) continue;
        const bool vec = (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0);
        const size_t n16 = vec ? bytes / 16 : 0;
        for (size_t i = gtid; i < n16; i += gstride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (size_t i = n16 * 16 + gtid; i < bytes; i += gstride) dst[i] = src[i];
    }
}

// uint8 HWC pixels -> (x / 255 - mean[c]) * inv_std[c] in fp32; one thread per pixel, C <= 4 handled from registers (one 4-byte load for C = 4,
// three byte loads for C = 3), NCHW destination written as C coalesced planes
template <int DST_NCHW>
__global__ __launch_bounds__(256) void images_u8_normalize_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t pixels, int hw, int C,
                                                                   const float* __restrict__ mean, const float* __restrict__ inv_std) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (size_t)gridDim.x * blockDim.x) {
        const unsigned char* sp = src + p * C;
        const size_t n = p / hw, q = p - n * hw;
        for (int c = 0; c < C; ++c) {
            const float v = ((float)sp[c] * (1.f / 255.f) - mean[c]) * inv_std[c];
            if (DST_NCHW) dst[(n * C + c) * hw + q] = v; else dst[p * C + c] = v;
        }
    }
}

struct ScalarPtrs { const float* p[DCV_MAX_COPY_JOBS]; };
__global__ __launch_bounds__(64) void publish_scalars_kernel(const dcv_copy_jobs* __restrict__ mailbox, ScalarPtrs srcs, int n, float* host_vals, int* host_seq) {
    const int t = threadIdx.x;
    if (t < n) host_vals[t] = *srcs.p[t];
    __threadfence_system();                                  // the values are in host memory before the sequence number is
    __syncthreads();
    if (t == 0) { *reinterpret_cast<volatile int*>(host_seq) = mailbox->seq; __threadfence_system(); }
}

template <int VEC>
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t nv) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        if constexpr (VEC == 4) {
            const float4 u = reinterpret_cast<const float4*>(a)[i]; const float4 v = reinterpret_cast<const float4*>(b)[i];
            float4 o; o.x = u.x + v.x; o.y = u.y + v.y; o.z = u.z + v.z; o.w = u.w + v.w;
            reinterpret_cast<float4*>(out)[i] = o;
        } else out[i] = a[i] + b[i];
    }
}

// bf16 tensors (the wide specs' residual links on the bf16 storage path): eight values per thread, fp32 sums rounded once
__global__ __launch_bounds__(256) void add_bf16_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out, size_t n8, const unsigned short* ta,
                                                       const unsigned short* tb, unsigned short* tout, int tail) {
    auto add2 = [](unsigned u, unsigned v) {
        const float lo = __uint_as_float(u << 16) + __uint_as_float(v << 16), hi = __uint_as_float(u & 0xffff0000u) + __uint_as_float(v & 0xffff0000u);
        const __bf16 l = (__bf16)lo, h = (__bf16)hi;
        return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 u = a[i], v = b[i];
        out[i] = make_uint4(add2(u.x, v.x), add2(u.y, v.y), add2(u.z, v.z), add2(u.w, v.w));
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) {
        const float s = __uint_as_float((unsigned)ta[threadIdx.x] << 16) + __uint_as_float((unsigned)tb[threadIdx.x] << 16);
        const __bf16 h = (__bf16)s;
        tout[threadIdx.x] = __builtin_bit_cast(unsigned short, h);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ a, float alpha, float* __restrict__ out, size_t nv) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        if constexpr (VEC == 4) {
            const float4 u = reinterpret_cast<const float4*>(a)[i];
            float4 o; o.x = u.x * alpha; o.y = u.y * alpha; o.z = u.z * alpha; o.w = u.w * alpha;
            reinterpret_cast<float4*>(out)[i] = o;
        } else out[i] = a[i] * alpha;
    }
}

// dst[i] = src[index[i]], 0 where index[i] < 0: a small tensor re-laid by an index map made once on the host (a first layer's filter regrouped for the im2col /
// folded-rows kernels and its gradient on the way back, deepcv_amd/ops.py `_RegroupFn`: as a library launch the re-layout is part of a replayed launch list)
__global__ __launch_bounds__(256) void gather_f32_kernel(const float* __restrict__ src, const int* __restrict__ index, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const int k = index[i]; dst[i] = k >= 0 ? src[k] : 0.f; }
}

// [N][A][B] -> [N][B][A] through a 32x33 LDS tile (coalesced on both sides; +1 padding avoids bank conflicts)
__global__ __launch_bounds__(256) void transpose_inner_kernel(const float* __restrict__ x, float* __restrict__ y, int A, int B) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const float* xs = x + (size_t)n * A * B;
    float* ys = y + (size_t)n * A * B;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int a = a0 + ty + k, b = b0 + tx;
        if (a < A && b < B) tile[ty + k][tx] = xs[(size_t)a * B + b];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int b = b0 + ty + k, a = a0 + tx;
        if (a < A && b < B) ys[(size_t)b * A + a] = tile[tx][ty + k];
    }
}

// source coordinate of torch's bilinear upsample kernel (area_pixel_compute_source_index)
__device__ __forceinline__ float src_index(float scale, int dst, bool align_corners) {
    if (align_corners) return scale * dst;
    const float s = scale * (dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

__device__ __forceinline__ void bilinear_taps(float scale, int dst, int in_size, bool align_corners, int& i0, int& i1, float& l1) {
    const float s = src_index(scale, dst, align_corners);
    i0 = (int)s; if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = s - (float)i0;
}

template <int VEC>
__global__ __launch_bounds__(256) void interp_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo,
                                                          float sh, float sw, int align) {
    const int CV = C / VEC;
    const size_t total = (size_t)N * Ho * Wo * CV;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * VEC; size_t p = i / CV;
        const int wo = (int)(p % Wo); p /= Wo; const int ho = (int)(p % Ho); const int n = (int)(p / Ho);
        int h0, h1, w0, w1; float lh, lw;
        bilinear_taps(sh, ho, H, align, h0, h1, lh); bilinear_taps(sw, wo, W, align, w0, w1, lw);
        const float* xs = x + (size_t)n * H * W * C;
        const float c00 = (1.f - lh) * (1.f - lw), c01 = (1.f - lh) * lw, c10 = lh * (1.f - lw), c11 = lh * lw;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float v = c00 * xs[((size_t)h0 * W + w0) * C + c + k] + c01 * xs[((size_t)h0 * W + w1) * C + c + k]
                          + c10 * xs[((size_t)h1 * W + w0) * C + c + k] + c11 * xs[((size_t)h1 * W + w1) * C + c + k];
            y[(((size_t)n * Ho + ho) * Wo + wo) * C + c + k] = v;
        }
    }
}

// Deterministic gather form of the backward: each INPUT pixel sums the output pixels whose 2x2 footprint contains it.
__device__ __forceinline__ void cand_range(float scale, int i, int out_size, int& lo, int& hi) {
    // outputs whose source coordinate s satisfies i-1 <= s < i+1 ; widened by one on both sides, exact test done by the caller
    const float inv = scale > 0.f ? 1.f / scale : 0.f;
    float a = ((float)i - 1.5f) * inv - 1.5f, b = ((float)i + 1.5f) * inv + 1.5f;
    lo = (int)floorf(a); hi = (int)ceilf(b);
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}

template <int VEC>
__global__ __launch_bounds__(256) void interp_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo,
                                                          float sh, float sw, int align) {
    const int CV = C / VEC;
    const size_t total = (size_t)N * H * W * CV;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * VEC; size_t p = i / CV;
        const int w = (int)(p % W); p /= W; const int h = (int)(p % H); const int n = (int)(p / H);
        int holo, hohi, wolo, wohi;
        cand_range(sh, h, Ho, holo, hohi); cand_range(sw, w, Wo, wolo, wohi);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        const float* dys = dy + (size_t)n * Ho * Wo * C;
        for (int ho = holo; ho <= hohi; ++ho) {
            int h0, h1; float lh; bilinear_taps(sh, ho, H, align, h0, h1, lh);
            float wh = 0.f;
            if (h0 == h) wh += 1.f - lh;
            if (h1 == h) wh += lh;
            if (wh == 0.f) continue;
            for (int wo = wolo; wo <= wohi; ++wo) {
                int w0, w1; float lw; bilinear_taps(sw, wo, W, align, w0, w1, lw);
                float ww = 0.f;
                if (w0 == w) ww += 1.f - lw;
                if (w1 == w) ww += lw;
                if (ww == 0.f) continue;
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += wh * ww * dys[((size_t)ho * Wo + wo) * C + c + k];
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) dx[(((size_t)n * H + h) * W + w) * C + c + k] = acc[k];
    }
}

inline float interp_scale(int in_size, int out_size, int align) {
    if (align) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    return (float)in_size / (float)out_size;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Channel concatenation whose parts may have another spatial size than the result (dense links with `allow_scaling`, reference submodule_creators.py:272-327:
// F.interpolate of the referenced features, then torch.cat): blockIdx.y = part; a part of the result's size is copied, any other one is resampled
// bilinearly on the fly (to_wide = 1, the body of interp_fwd_kernel writing into its channel range of the wide rows) or receives the gather form of the
// bilinear backward read straight from its channel range of the wide gradient (to_wide = 0, the body of interp_bwd_kernel) — one launch per direction
// instead of one per resized part plus the concat / split.
struct ResizeConcatTable { float* part[DCV_MAX_CONCAT]; int c[DCV_MAX_CONCAT]; int off[DCV_MAX_CONCAT]; int h[DCV_MAX_CONCAT]; int w[DCV_MAX_CONCAT];
                           float sh[DCV_MAX_CONCAT]; float sw[DCV_MAX_CONCAT]; int count; };

template <int VEC>
__global__ __launch_bounds__(256) void resize_concat_kernel(ResizeConcatTable t, float* __restrict__ wide, int wide_c, int N, int Ho, int Wo, int align, int to_wide) {
    const int k = blockIdx.y;
    float* __restrict__ part = t.part[k];
    const int C = t.c[k], off = t.off[k], H = t.h[k], W = t.w[k];
    const int CV = C / VEC;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gstride = (size_t)gridDim.x * blockDim.x;
    if (H == Ho && W == Wo) {
        const size_t total = (size_t)N * Ho * Wo * CV;
        for (size_t i = gtid; i < total; i += gstride) {
            const size_t r = i / CV; const int c = (int)(i % CV) * VEC;
            float* pw = wide + r * wide_c + off + c; float* pp = part + r * C + c;
            if constexpr (VEC == 4) { if (to_wide) *reinterpret_cast<float4*>(pw) = *reinterpret_cast<const float4*>(pp); else *reinterpret_cast<float4*>(pp) = *reinterpret_cast<const float4*>(pw); }
            else { if (to_wide) *pw = *pp; else *pp = *pw; }
        }
        return;
    }
    const float sh = t.sh[k], sw = t.sw[k];
    if (to_wide) {
        const size_t total = (size_t)N * Ho * Wo * CV;
        for (size_t i = gtid; i < total; i += gstride) {
            const int c = (int)(i % CV) * VEC; size_t p = i / CV;
            const int wo = (int)(p % Wo); p /= Wo; const int ho = (int)(p % Ho); const int n = (int)(p / Ho);
            int h0, h1, w0, w1; float lh, lw;
            bilinear_taps(sh, ho, H, align, h0, h1, lh); bilinear_taps(sw, wo, W, align, w0, w1, lw);
            const float* xs = part + (size_t)n * H * W * C;
            const float c00 = (1.f - lh) * (1.f - lw), c01 = (1.f - lh) * lw, c10 = lh * (1.f - lw), c11 = lh * lw;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float v = c00 * xs[((size_t)h0 * W + w0) * C + c + q] + c01 * xs[((size_t)h0 * W + w1) * C + c + q]
                              + c10 * xs[((size_t)h1 * W + w0) * C + c + q] + c11 * xs[((size_t)h1 * W + w1) * C + c + q];
                wide[(((size_t)n * Ho + ho) * Wo + wo) * wide_c + off + c + q] = v;
            }
        }
        return;
    }
    const size_t total = (size_t)N * H * W * CV;
    for (size_t i = gtid; i < total; i += gstride) {
        const int c = (int)(i % CV) * VEC; size_t p = i / CV;
        const int w = (int)(p % W); p /= W; const int h = (int)(p % H); const int n = (int)(p / H);
        int holo, hohi, wolo, wohi;
        cand_range(sh, h, Ho, holo, hohi); cand_range(sw, w, Wo, wolo, wohi);
        float acc[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
        const float* dys = wide + (size_t)n * Ho * Wo * wide_c + off;
        for (int ho = holo; ho <= hohi; ++ho) {
            int h0, h1; float lh; bilinear_taps(sh, ho, H, align, h0, h1, lh);
            float wh = 0.f;
            if (h0 == h) wh += 1.f - lh;
            if (h1 == h) wh += lh;
            if (wh == 0.f) continue;
            for (int wo = wolo; wo <= wohi; ++wo) {
                int w0, w1; float lw; bilinear_taps(sw, wo, W, align, w0, w1, lw);
                float ww = 0.f;
                if (w0 == w) ww += 1.f - lw;
                if (w1 == w) ww += lw;
                if (ww == 0.f) continue;
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] += wh * ww * dys[((size_t)ho * Wo + wo) * wide_c + c + q];
            }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) part[(((size_t)n * H + h) * W + w) * C + c + q] = acc[q];
    }
}


// ------------------------------------------------------------------------------------------------ ordered sums of partial slabs
// dst[i] = sum over k < chunks of src[k * stride + i] for up to DCV_MAX_SLAB_SEGMENTS independent segments in one launch (second stage of the weight
// gradients).  Block = 8 entries x 128 slab lanes: a thread adds chunks / 128 slabs (a short load chain: this kernel is pure latency), the 8 slab lanes
// of a wave are merged by a fixed xor butterfly, the 16 waves in order through LDS -> the result only depends on (segment shape, entry), not on the launch.
// Long segments (>= SLAB_WIDE_COUNT entries: the linear layer's 12.8 k, a 16->16 3x3 filter's 2.3 k) use blocks of 32 entries x 32 slab lanes instead:
// 128-B runs per slab row instead of 32-B ones (8.5 of this kernel's 9.5 MB on the stock step).
__global__ __launch_bounds__(1024) void slab_reduce_kernel(SlabTable t) {
    __shared__ float red[16][SLAB_WIDE];
    int si = 0;
    while (si + 1 < t.count && (int)blockIdx.x >= t.block_start[si + 1]) ++si;          // wave-uniform: scalar loop
    const dcv_slab_segment& g = t.seg[si];
    float v; size_t di;
    if (dcv_slab_block_sum(g, t.wshift[si], (int)blockIdx.x - t.block_start[si], red, v, di)) g.dst[di] = v;      // (the body: slab_sum.h, shared with the optimizer's tail launch)
}

}  // namespace

int dcv_launch_slab_reduce(int n, const dcv_slab_segment* segments, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += DCV_MAX_SLAB_SEGMENTS) {
        SlabTable t{};
        int blocks = 0;
        for (int i = i0; i < n && i < i0 + DCV_MAX_SLAB_SEGMENTS; ++i) {
            const dcv_slab_segment& g = segments[i];
            if (!g.src || !g.dst || g.chunks < 1 || g.stride < g.count) return DCV_ERR_INVALID_ARG;
            if (g.tr_a < 0 || g.tr_b < 0 || ((g.tr_a > 0) != (g.tr_b > 0)) || (g.tr_a > 0 && g.count % ((unsigned long long)g.tr_a * g.tr_b) != 0)) return DCV_ERR_INVALID_ARG;
            if (g.count == 0) continue;
            const int ws = g.count >= SLAB_WIDE_COUNT ? 5 : 3;
            t.block_start[t.count] = blocks;
            t.wshift[t.count] = ws;
            t.seg[t.count++] = g;
            blocks += (int)dcv_ceil_div((size_t)g.count, (size_t)1 << ws);
        }
        t.block_start[t.count] = blocks;
        if (blocks > 0) hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, t);
    }
    DCV_RETURN_LAUNCH();
}

unsigned long long dcv_launch_counter_[DCV_KERNEL_FAMILIES] = {};

extern "C" {

int dcv_slab_reduce(int n, const dcv_slab_segment* segments, void* stream) {
    DCV_CHECK_ARG(n >= 0 && (n == 0 || segments));
    return dcv_launch_slab_reduce(n, segments, dcv_stream(stream));
}


const char* dcv_strerror(int code) {
    switch (code) {
        case DCV_OK: return "ok";
        case DCV_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive dimension or inconsistent shapes)";
        case DCV_ERR_UNSUPPORTED: return "unsupported configuration for this build of libdeepcv_hip";
        case DCV_ERR_WORKSPACE: return "workspace too small (query the matching *_workspace_bytes function)";
        case DCV_ERR_LAUNCH: return "HIP kernel launch failed (hipGetLastError != hipSuccess)";
        default: return "unknown deepcv_hip error code";
    }
}

int dcv_version(void) { return 100; }

int dcv_launch_counts(unsigned long long* counts, int n) {
    if (counts) for (int i = 0; i < n && i < DCV_KERNEL_FAMILIES; ++i) counts[i] = __atomic_load_n(&dcv_launch_counter_[i], __ATOMIC_RELAXED);
    return DCV_KERNEL_FAMILIES;
}

int dcv_device_info(int device, int* cu_count, int* wave_size, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DCV_ERR_INVALID_ARG;
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1); arch[arch_len - 1] = 0; }
    return DCV_OK;
}

int dcv_act_fwd(const float* x, float* y, size_t numel, int act, float slope, void* stream) {
    DCV_CHECK_ARG(x && y && numel > 0 && act >= DCV_ACT_NONE && act <= DCV_ACT_TANH);
    hipStream_t st = dcv_stream(stream);
    if (numel % 4 == 0 && aligned16(x) && aligned16(y)) hipLaunchKernelGGL((act_fwd_kernel<4>), dim3(dcv_stream_grid(numel / 4, 256)), dim3(256), 0, st, x, y, numel / 4, act, slope);
    else hipLaunchKernelGGL((act_fwd_kernel<1>), dim3(dcv_stream_grid(numel, 256)), dim3(256), 0, st, x, y, numel, act, slope);
    DCV_RETURN_LAUNCH();
}

size_t dcv_act_bwd_workspace_bytes(size_t rows, int channels) {
    if (rows == 0 || channels <= 0) return 0;
    return dcv_align_up((size_t)dcv_colreduce_blocks(rows, channels) * channels * sizeof(double), 256);
}

int dcv_act_bwd(const float* dy, const float* y, float* dx, size_t rows, int C, int act, float slope, float* db, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(dy && y && dx && rows > 0 && C > 0 && act >= DCV_ACT_NONE && act <= DCV_ACT_TANH);
    hipStream_t st = dcv_stream(stream);
    const size_t numel = rows * (size_t)C;
    if (!db) {
        if (numel % 4 == 0 && aligned16(dy) && aligned16(y) && aligned16(dx))
            hipLaunchKernelGGL((act_bwd_kernel<4>), dim3(dcv_stream_grid(numel / 4, 256)), dim3(256), 0, st, dy, y, dx, numel / 4, act, slope);
        else hipLaunchKernelGGL((act_bwd_kernel<1>), dim3(dcv_stream_grid(numel, 256)), dim3(256), 0, st, dy, y, dx, numel, act, slope);
        DCV_RETURN_LAUNCH();
    }
    if (!ws || ws_bytes < dcv_act_bwd_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    double* partial = reinterpret_cast<double*>(ws);
    const unsigned nb = dcv_colreduce_blocks(rows, C);
    ActBwdLoader ld{dy, y, dx, act, slope};
    if (C % 4 == 0) hipLaunchKernelGGL((act_bwd_bias_kernel<4>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    else hipLaunchKernelGGL((act_bwd_bias_kernel<1>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)C), dim3(64), 0, st, partial, (int)nb, C, db);
    DCV_RETURN_LAUNCH();
}

int dcv_channel_sum(const float* x, float* out, size_t rows, int C, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(x && out && rows > 0 && C > 0);
    if (!ws || ws_bytes < dcv_act_bwd_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    double* partial = reinterpret_cast<double*>(ws);
    const unsigned nb = dcv_colreduce_blocks(rows, C);
    PlainLoader ld{x};
    if (C % 4 == 0 && aligned16(x)) hipLaunchKernelGGL((channel_sum_kernel<4>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    else hipLaunchKernelGGL((channel_sum_kernel<1>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)C), dim3(64), 0, st, partial, (int)nb, C, out);
    DCV_RETURN_LAUNCH();
}

int dcv_channel_sum_bf16(const void* x, float* out, size_t rows, int C, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(x && out && rows > 0 && C > 0);
    if (!ws || ws_bytes < dcv_act_bwd_workspace_bytes(rows, C)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    double* partial = reinterpret_cast<double*>(ws);
    const unsigned nb = dcv_colreduce_blocks(rows, C);
    Bf16Loader ld{reinterpret_cast<const unsigned short*>(x)};
    if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0) hipLaunchKernelGGL((channel_sum_kernel<4, Bf16Loader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    else hipLaunchKernelGGL((channel_sum_kernel<1, Bf16Loader>), dim3(nb), dim3(DCV_CR_THREADS), 0, st, ld, rows, C, partial);
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)C), dim3(64), 0, st, partial, (int)nb, C, out);
    DCV_RETURN_LAUNCH();
}

int dcv_copy_channels(const float* src, int src_c, int src_off, float* dst, int dst_c, int dst_off, int copy_c, size_t rows, void* stream) {
    DCV_CHECK_ARG(src && dst && rows > 0 && copy_c > 0 && src_off >= 0 && dst_off >= 0 && src_off + copy_c <= src_c && dst_off + copy_c <= dst_c);
    hipStream_t st = dcv_stream(stream);
    const bool v4 = (copy_c % 4 == 0) && (src_c % 4 == 0) && (dst_c % 4 == 0) && (src_off % 4 == 0) && (dst_off % 4 == 0) && aligned16(src) && aligned16(dst);
    if (v4) hipLaunchKernelGGL((copy_channels_kernel<4>), dim3(dcv_stream_grid(rows * (size_t)(copy_c / 4), 256)), dim3(256), 0, st, src, src_c, src_off, dst, dst_c, dst_off, copy_c, rows);
    else hipLaunchKernelGGL((copy_channels_kernel<1>), dim3(dcv_stream_grid(rows * (size_t)copy_c, 256)), dim3(256), 0, st, src, src_c, src_off, dst, dst_c, dst_off, copy_c, rows);
    DCV_RETURN_LAUNCH();
}

static int concat_split(int count, float* const* parts, const int* cs, float* wide, size_t rows, int to_wide, void* stream) {
    DCV_CHECK_ARG(count > 0 && count <= DCV_MAX_CONCAT && parts && cs && wide && rows > 0);
    ConcatTable t{}; t.count = count;
    int off = 0, maxc = 0; bool v4 = aligned16(wide);
    for (int k = 0; k < count; ++k) {
        DCV_CHECK_ARG(cs[k] > 0 && (parts[k] != nullptr || !to_wide));
        t.part[k] = parts[k]; t.c[k] = cs[k]; t.off[k] = off;
        v4 = v4 && (cs[k] % 4 == 0) && (off % 4 == 0) && (parts[k] == nullptr || aligned16(parts[k]));
        if (cs[k] > maxc) maxc = cs[k];
        off += cs[k];
    }
    v4 = v4 && (off % 4 == 0);
    // split: a NULL part means "gradient not needed": launch only the others (compact the table)
    int n = 0;
    for (int k = 0; k < count; ++k) if (t.part[k]) { t.part[n] = t.part[k]; t.c[n] = t.c[k]; t.off[n] = t.off[k]; ++n; }
    if (n == 0) return DCV_OK;
    hipStream_t st = dcv_stream(stream);
    if (v4) hipLaunchKernelGGL((concat_split_kernel<4>), dim3(dcv_stream_grid(rows * (size_t)(maxc / 4), 256), (unsigned)n), dim3(256), 0, st, t, wide, off, rows, to_wide);
    else hipLaunchKernelGGL((concat_split_kernel<1>), dim3(dcv_stream_grid(rows * (size_t)maxc, 256), (unsigned)n), dim3(256), 0, st, t, wide, off, rows, to_wide);
    DCV_RETURN_LAUNCH();
}

static int resize_concat(int count, float* const* parts, const int* cs, const int* hs, const int* ws, float* wide, int n, int ho, int wo, int align, int to_wide, void* stream) {
    DCV_CHECK_ARG(count > 0 && count <= DCV_MAX_CONCAT && parts && cs && hs && ws && wide && n > 0 && ho > 0 && wo > 0);
    ResizeConcatTable t{}; t.count = count;
    int off = 0; bool v4 = aligned16(wide);
    size_t maxwork = 0;
    for (int k = 0; k < count; ++k) {
        DCV_CHECK_ARG(cs[k] > 0 && hs[k] > 0 && ws[k] > 0 && (parts[k] != nullptr || !to_wide));
        t.part[k] = parts[k]; t.c[k] = cs[k]; t.off[k] = off; t.h[k] = hs[k]; t.w[k] = ws[k];
        t.sh[k] = interp_scale(hs[k], ho, align); t.sw[k] = interp_scale(ws[k], wo, align);
        v4 = v4 && (cs[k] % 4 == 0) && (off % 4 == 0) && (parts[k] == nullptr || aligned16(parts[k]));
        const bool same = hs[k] == ho && ws[k] == wo;
        const size_t work = (size_t)n * ((to_wide || same) ? (size_t)ho * wo : (size_t)hs[k] * ws[k]) * cs[k];
        if (work > maxwork) maxwork = work;
        off += cs[k];
    }
    v4 = v4 && (off % 4 == 0);
    int m = 0;                                                            // split: a NULL part means "gradient not needed"
    for (int k = 0; k < count; ++k) if (t.part[k]) {
        t.part[m] = t.part[k]; t.c[m] = t.c[k]; t.off[m] = t.off[k]; t.h[m] = t.h[k]; t.w[m] = t.w[k]; t.sh[m] = t.sh[k]; t.sw[m] = t.sw[k]; ++m; }
    if (m == 0) return DCV_OK;
    hipStream_t st = dcv_stream(stream);
    if (v4) hipLaunchKernelGGL((resize_concat_kernel<4>), dim3(dcv_stream_grid(maxwork / 4, 256), (unsigned)m), dim3(256), 0, st, t, wide, off, n, ho, wo, align, to_wide);
    else hipLaunchKernelGGL((resize_concat_kernel<1>), dim3(dcv_stream_grid(maxwork, 256), (unsigned)m), dim3(256), 0, st, t, wide, off, n, ho, wo, align, to_wide);
    DCV_RETURN_LAUNCH();
}

int dcv_resize_concat_channels(int count, const float* const* srcs, const int* channels, const int* heights, const int* widths, float* dst,
                               int n, int ho, int wo, int align_corners, void* stream) {
    return resize_concat(count, const_cast<float* const*>(reinterpret_cast<const float* const*>(srcs)), channels, heights, widths, dst, n, ho, wo, align_corners ? 1 : 0, 1, stream);
}

int dcv_resize_split_channels(int count, float* const* dsts, const int* channels, const int* heights, const int* widths, const float* src,
                              int n, int ho, int wo, int align_corners, void* stream) {
    return resize_concat(count, dsts, channels, heights, widths, const_cast<float*>(src), n, ho, wo, align_corners ? 1 : 0, 0, stream);
}

int dcv_concat_channels(int count, const float* const* srcs, const int* channels, float* dst, size_t rows, void* stream) {
    return concat_split(count, const_cast<float* const*>(reinterpret_cast<const float* const*>(srcs)), channels, dst, rows, 1, stream);
}

int dcv_split_channels(int count, float* const* dsts, const int* channels, const float* src, size_t rows, void* stream) {
    return concat_split(count, dsts, channels, const_cast<float*>(src), rows, 0, stream);
}

int dcv_indirect_copy(const dcv_copy_jobs* jobs, void* stream) {
    DCV_CHECK_ARG(jobs != nullptr);
    // the grid cannot depend on sizes only known at run time: 4 blocks per CU stream any batch at HBM speed and cost nothing for tiny ones
    hipLaunchKernelGGL(indirect_copy_kernel, dim3(DCV_NUM_CU * 4), dim3(256), 0, dcv_stream(stream), jobs);
    DCV_RETURN_LAUNCH();
}

int dcv_images_u8_normalize(const unsigned char* src, float* dst, int n, int h, int w, int c, int dst_layout, const float* mean, const float* inv_std, void* stream) {
    DCV_CHECK_ARG(src && dst && mean && inv_std && n > 0 && h > 0 && w > 0 && c > 0 && (dst_layout == DCV_LAYOUT_NCHW || dst_layout == DCV_LAYOUT_NHWC));
    const size_t pixels = (size_t)n * h * w;
    hipStream_t st = dcv_stream(stream);
    if (dst_layout == DCV_LAYOUT_NCHW) hipLaunchKernelGGL((images_u8_normalize_kernel<1>), dim3(dcv_stream_grid(pixels, 256)), dim3(256), 0, st, src, dst, pixels, h * w, c, mean, inv_std);
    else hipLaunchKernelGGL((images_u8_normalize_kernel<0>), dim3(dcv_stream_grid(pixels, 256)), dim3(256), 0, st, src, dst, pixels, h * w, c, mean, inv_std);
    DCV_RETURN_LAUNCH();
}

int dcv_publish_scalars(const dcv_copy_jobs* mailbox, const float* const* srcs, int n, float* host_vals, int* host_seq, void* stream) {
    DCV_CHECK_ARG(mailbox && srcs && n > 0 && n <= DCV_MAX_COPY_JOBS && host_vals && host_seq);
    ScalarPtrs sp{};
    for (int k = 0; k < n; ++k) { DCV_CHECK_ARG(srcs[k] != nullptr); sp.p[k] = srcs[k]; }
    hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(64), 0, dcv_stream(stream), mailbox, sp, n, host_vals, host_seq);
    DCV_RETURN_LAUNCH();
}

int dcv_add(const float* a, const float* b, float* out, size_t numel, void* stream) {
    DCV_CHECK_ARG(a && b && out && numel > 0);
    hipStream_t st = dcv_stream(stream);
    if (numel % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(out)) hipLaunchKernelGGL((add_kernel<4>), dim3(dcv_stream_grid(numel / 4, 256)), dim3(256), 0, st, a, b, out, numel / 4);
    else hipLaunchKernelGGL((add_kernel<1>), dim3(dcv_stream_grid(numel, 256)), dim3(256), 0, st, a, b, out, numel);
    DCV_RETURN_LAUNCH();
}

int dcv_add_bf16(const void* a, const void* b, void* out, size_t numel, void* stream) {
    DCV_CHECK_ARG(a && b && out && numel > 0 && aligned16(a) && aligned16(b) && aligned16(out));
    const size_t n8 = numel / 8; const int tail = (int)(numel - 8 * n8);
    const unsigned short* ta = reinterpret_cast<const unsigned short*>(a) + 8 * n8; const unsigned short* tb = reinterpret_cast<const unsigned short*>(b) + 8 * n8;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(dcv_stream_grid(n8 ? n8 : 1, 256)), dim3(256), 0, dcv_stream(stream), reinterpret_cast<const uint4*>(a), reinterpret_cast<const uint4*>(b),
                       reinterpret_cast<uint4*>(out), n8, ta, tb, reinterpret_cast<unsigned short*>(out) + 8 * n8, tail);
    DCV_RETURN_LAUNCH();
}

int dcv_scale(const float* a, float alpha, float* out, size_t numel, void* stream) {
    DCV_CHECK_ARG(a && out && numel > 0);
    hipStream_t st = dcv_stream(stream);
    if (numel % 4 == 0 && aligned16(a) && aligned16(out)) hipLaunchKernelGGL((scale_kernel<4>), dim3(dcv_stream_grid(numel / 4, 256)), dim3(256), 0, st, a, alpha, out, numel / 4);
    else hipLaunchKernelGGL((scale_kernel<1>), dim3(dcv_stream_grid(numel, 256)), dim3(256), 0, st, a, alpha, out, numel);
    DCV_RETURN_LAUNCH();
}

int dcv_gather_f32(const float* src, const int* index, float* dst, size_t numel, void* stream) {
    DCV_CHECK_ARG(src && index && dst && numel > 0);
    hipLaunchKernelGGL(gather_f32_kernel, dim3(dcv_stream_grid(numel, 256)), dim3(256), 0, dcv_stream(stream), src, index, dst, numel);
    DCV_RETURN_LAUNCH();
}

int dcv_transpose_inner(const float* x, float* y, int n, int a, int b, void* stream) {
    DCV_CHECK_ARG(x && y && n > 0 && a > 0 && b > 0 && n <= 65535);
    hipStream_t st = dcv_stream(stream);
    dim3 grid((unsigned)dcv_ceil_div(b, 32), (unsigned)dcv_ceil_div(a, 32), (unsigned)n);
    DCV_CHECK_ARG(grid.y <= 65535);
    hipLaunchKernelGGL(transpose_inner_kernel, grid, dim3(256), 0, st, x, y, a, b);
    DCV_RETURN_LAUNCH();
}

int dcv_interp_bilinear_fwd(const float* x, float* y, int n, int h, int w, int c, int ho, int wo, int align_corners, void* stream) {
    DCV_CHECK_ARG(x && y && n > 0 && h > 0 && w > 0 && c > 0 && ho > 0 && wo > 0);
    hipStream_t st = dcv_stream(stream);
    const float sh = interp_scale(h, ho, align_corners), sw = interp_scale(w, wo, align_corners);
    if (c % 4 == 0) hipLaunchKernelGGL((interp_fwd_kernel<4>), dim3(dcv_stream_grid((size_t)n * ho * wo * (c / 4), 256)), dim3(256), 0, st, x, y, n, h, w, c, ho, wo, sh, sw, align_corners);
    else hipLaunchKernelGGL((interp_fwd_kernel<1>), dim3(dcv_stream_grid((size_t)n * ho * wo * c, 256)), dim3(256), 0, st, x, y, n, h, w, c, ho, wo, sh, sw, align_corners);
    DCV_RETURN_LAUNCH();
}

int dcv_interp_bilinear_bwd(const float* dy, float* dx, int n, int h, int w, int c, int ho, int wo, int align_corners, void* stream) {
    DCV_CHECK_ARG(dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && ho > 0 && wo > 0);
    hipStream_t st = dcv_stream(stream);
    const float sh = interp_scale(h, ho, align_corners), sw = interp_scale(w, wo, align_corners);
    if (c % 4 == 0) hipLaunchKernelGGL((interp_bwd_kernel<4>), dim3(dcv_stream_grid((size_t)n * h * w * (c / 4), 256)), dim3(256), 0, st, dy, dx, n, h, w, c, ho, wo, sh, sw, align_corners);
    else hipLaunchKernelGGL((interp_bwd_kernel<1>), dim3(dcv_stream_grid((size_t)n * h * w * c, 256)), dim3(256), 0, st, dy, dx, n, h, w, c, ho, wo, sh, sw, align_corners);
    DCV_RETURN_LAUNCH();
}

}  // extern "C"
