// augment.hip — GPU augmentation: AugMix-style chains of the reference's 13 image operations (src/deepcv/meta/data/augmentation.py:65-132,
// mixed like `augment_and_mix`, :165-203), applied per sample on the device and fused with ToTensor + Normalize.
//
// The reference calls Pillow on the CPU; every operation here reproduces Pillow's arithmetic BIT FOR BIT on the uint8 images (checked against
// oracle/augment_ref.py, which is pinned against Pillow itself): integer tables (posterize, solarize, equalize), double-precision table / resampling
// arithmetic without fused multiply-adds (autocontrast; rotate / shear / translate = Image.transform(AFFINE, BILINEAR)), float32 blends (ImageEnhance.Color /
// Contrast / Brightness / Sharpness), the 3x3 SMOOTH filter.  Several operations need a statistic of the WHOLE intermediate image (histogram, mean grey level)
// or a neighbourhood of it, so a chain cannot be one pass per output pixel: one block runs one chain of one sample, stage by stage, on two uint8 image buffers
// (scratch, L2-resident) with block barriers in between; histograms use integer LDS atomics (order-independent, exact).  A second kernel mixes the chains with
// the original image and normalises.
//
// Roofline: HBM / latency; per batch of 512 CIFAR images the chains touch ~10 MB of L2-resident scratch.  Not on the training step's critical path: it runs on
// the prefetch stream (deepcv_amd/meta/data/datasets.py) while the previous batch trains.
#include "common.h"

// Pillow's C code is compiled without fused multiply-adds; hipcc contracts a * b + c by default, which moves results across a truncation boundary now and
// then.  No contraction anywhere in this file — and the arithmetic goes through the helpers below rather than HIP's __dmul_rn / __dadd_rn, which are plain
// operators defined in a header (outside this pragma's reach: their products and sums stay contractable after inlining).
#pragma clang fp contract(off)

namespace {

#define AUG_THREADS 256
__device__ __forceinline__ double aug_dmul(double a, double b) { return a * b; }
__device__ __forceinline__ double aug_dadd(double a, double b) { return a + b; }
__device__ __forceinline__ double aug_ddiv(double a, double b) { return a / b; }
__device__ __forceinline__ float aug_fmul(float a, float b) { return a * b; }
__device__ __forceinline__ float aug_fadd(float a, float b) { return a + b; }

__device__ __forceinline__ int aug_gray(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }      // Pillow Convert.c rgb2l

// Image.blend(deg, img, a) of Blend.c: float32  deg + a * (img - deg), truncated; clipped when a is outside [0, 1]
__device__ __forceinline__ int aug_blend(int d, int i, float a, bool inside) {
    const float t = aug_fadd((float)d, aug_fmul(a, (float)(i - d)));
    if (inside) return (int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

// per-channel histograms of an image into hist[C][256] (LDS)
__device__ __forceinline__ void aug_histogram(const unsigned char* img, int size, int C, unsigned* hist) {
    for (int i = threadIdx.x; i < C * 256; i += AUG_THREADS) hist[i] = 0;
    __syncthreads();
    if (C == 1) { for (int i = threadIdx.x; i < size; i += AUG_THREADS) atomicAdd(&hist[img[i]], 1u); }
    else { for (int i = threadIdx.x; i < size; i += AUG_THREADS) atomicAdd(&hist[(i % C) * 256 + img[i]], 1u); }
    __syncthreads();
}

__device__ __forceinline__ void aug_apply_lut(unsigned char* img, int size, int C, const unsigned char* lut) {
    if (C == 1) { for (int i = threadIdx.x; i < size; i += AUG_THREADS) img[i] = lut[img[i]]; }
    else { for (int i = threadIdx.x; i < size; i += AUG_THREADS) img[i] = lut[(i % C) * 256 + img[i]]; }
}

// Image.transform(size, AFFINE, m, resample=BILINEAR), zero fill: Geometry.c affine_transform + bilinear_filter8 / 32RGB (double arithmetic, truncation)
__device__ __forceinline__ void aug_affine(const unsigned char* in, unsigned char* out, int H, int W, int C, const double* m) {
    const double a0 = m[0], a1 = m[1], a2 = m[2], a3 = m[3], a4 = m[4], a5 = m[5];
    for (int p = threadIdx.x; p < H * W; p += AUG_THREADS) {
        const int y = p / W, x = p - y * W;
        const double xin = (double)x + 0.5, yin = (double)y + 0.5;
        double xo = aug_dadd(aug_dadd(aug_dmul(a0, xin), aug_dmul(a1, yin)), a2);
        double yo = aug_dadd(aug_dadd(aug_dmul(a3, xin), aug_dmul(a4, yin)), a5);
        const bool ok = !(xo < 0.0 || xo >= (double)W || yo < 0.0 || yo >= (double)H);
        unsigned char* op = out + (size_t)p * C;
        if (!ok) { for (int c = 0; c < C; ++c) op[c] = 0; continue; }
        xo = aug_dadd(xo, -0.5); yo = aug_dadd(yo, -0.5);
        const double fx = floor(xo), fy = floor(yo);
        const int xi = (int)fx, yi = (int)fy;
        const double dx = aug_dadd(xo, -fx), dy = aug_dadd(yo, -fy);
        const int x0 = xi < 0 ? 0 : (xi >= W ? W - 1 : xi), x1 = xi + 1 < 0 ? 0 : (xi + 1 >= W ? W - 1 : xi + 1);
        const int y0 = yi < 0 ? 0 : (yi >= H ? H - 1 : yi);
        const bool y1ok = yi + 1 >= 0 && yi + 1 < H;
        const unsigned char* r0 = in + (size_t)y0 * W * C;
        const unsigned char* r1 = in + (size_t)(y1ok ? yi + 1 : y0) * W * C;
        for (int c = 0; c < C; ++c) {
            const double p00 = (double)r0[x0 * C + c], p01 = (double)r0[x1 * C + c];
            const double v1 = aug_dadd(p00, aug_dmul(aug_dadd(p01, -p00), dx));
            double v2 = v1;
            if (y1ok) { const double p10 = (double)r1[x0 * C + c], p11 = (double)r1[x1 * C + c]; v2 = aug_dadd(p10, aug_dmul(aug_dadd(p11, -p10), dx)); }
            const double v = aug_dadd(v1, aug_dmul(aug_dadd(v2, -v1), dy));
            op[c] = (unsigned char)(long long)v;
        }
    }
}

// ImageEnhance.Sharpness: blend(filter(SMOOTH), img, a); SMOOTH = (1 1 1; 1 5 1; 1 1 1) / 13 in float32 + 0.5, truncated, border pixels copied (Filter.c)
__device__ __forceinline__ void aug_sharpness(const unsigned char* in, unsigned char* out, int H, int W, int C, float a) {
    const bool inside = a >= 0.f && a <= 1.f;
    const float k1 = (float)(1.0 / 13.0), k5 = (float)(5.0 / 13.0);
    for (int i = threadIdx.x; i < H * W * C; i += AUG_THREADS) {
        const int c = i % C, p = i / C;
        const int y = p / W, x = p - y * W;
        const int v = in[i];
        int s = v;
        if (H >= 3 && W >= 3 && y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
            const unsigned char* up = in + ((size_t)(y - 1) * W + x) * C + c;
            const unsigned char* md = in + ((size_t)y * W + x) * C + c;
            const unsigned char* dn = in + ((size_t)(y + 1) * W + x) * C + c;
            float ss = 0.5f;
            ss = aug_fadd(ss, aug_fadd(aug_fadd(aug_fmul((float)dn[-C], k1), aug_fmul((float)dn[0], k1)), aug_fmul((float)dn[C], k1)));
            ss = aug_fadd(ss, aug_fadd(aug_fadd(aug_fmul((float)md[-C], k1), aug_fmul((float)md[0], k5)), aug_fmul((float)md[C], k1)));
            ss = aug_fadd(ss, aug_fadd(aug_fadd(aug_fmul((float)up[-C], k1), aug_fmul((float)up[0], k1)), aug_fmul((float)up[C], k1)));
            s = ss <= 0.f ? 0 : (ss >= 255.f ? 255 : (int)ss);
        }
        out[i] = (unsigned char)aug_blend(s, v, a, inside);
    }
}

// One block = one chain of one sample, stage by stage on the two image buffers of (sample, chain) in `scratch`; the chain's final image ends in buffer 0.
__global__ __launch_bounds__(AUG_THREADS) void augmix_chain_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ scratch, int H, int W, int C,
                                                                    const dcv_augmix_recipe* __restrict__ recipes) {
    __shared__ unsigned hist[3 * 256];
    __shared__ unsigned char lut[3 * 256];
    __shared__ unsigned gsum;
    const int chain = blockIdx.x, n = blockIdx.y;
    const dcv_augmix_recipe& rp = recipes[n];
    if (rp.weight[chain] == 0.f) return;
    const int size = H * W * C;
    unsigned char* bufA = scratch + ((size_t)(n * DCV_AUG_MAX_CHAINS + chain) * 2) * size;
    unsigned char* cur = bufA;
    unsigned char* oth = bufA + size;
    const unsigned char* img = src + (size_t)n * size;
    for (int i = threadIdx.x; i < size; i += AUG_THREADS) cur[i] = img[i];
    __syncthreads();
    for (int k = 0; k < DCV_AUG_MAX_OPS; ++k) {
        const int op = rp.op[chain][k], ip = rp.ip[chain][k];
        const float fp = rp.fp[chain][k];
        const bool inside = fp >= 0.f && fp <= 1.f;
        switch (op) {
            case DCV_AUG_POSTERIZE: { const int mask = ~((1 << (8 - ip)) - 1) & 0xFF; for (int i = threadIdx.x; i < size; i += AUG_THREADS) cur[i] = cur[i] & mask; } break;
            case DCV_AUG_SOLARIZE: for (int i = threadIdx.x; i < size; i += AUG_THREADS) { const int v = cur[i]; cur[i] = (unsigned char)(v < ip ? v : 255 - v); } break;
            case DCV_AUG_BRIGHTNESS: for (int i = threadIdx.x; i < size; i += AUG_THREADS) cur[i] = (unsigned char)aug_blend(0, cur[i], fp, inside); break;
            case DCV_AUG_AUTOCONTRAST: {
                // ImageOps.autocontrast (cutoff 0), band by band: lut[i] = clip(int(i * scale + offset)), double arithmetic as Python evaluates it
                aug_histogram(cur, size, C, hist);
                for (int i = threadIdx.x; i < C * 256; i += AUG_THREADS) {
                    const unsigned* h = hist + (i / 256) * 256;
                    int lo = 0, hi = 255;
                    while (lo < 255 && h[lo] == 0) ++lo;
                    while (hi > 0 && h[hi] == 0) --hi;
                    int v = i & 255;
                    if (hi > lo) {
                        const double scale = aug_ddiv(255.0, (double)(hi - lo)), offset = aug_dmul(-(double)lo, scale);
                        const double t = aug_dadd(aug_dmul((double)v, scale), offset);
                        v = (int)t; v = v < 0 ? 0 : (v > 255 ? 255 : v);
                    }
                    lut[i] = (unsigned char)v;
                }
                __syncthreads();
                aug_apply_lut(cur, size, C, lut);
            } break;
            case DCV_AUG_EQUALIZE: {
                // ImageOps.equalize, band by band (integer arithmetic; Image.point clips the table to 255)
                aug_histogram(cur, size, C, hist);
                if ((int)threadIdx.x < C) {
                    const unsigned* h = hist + threadIdx.x * 256;
                    unsigned char* l = lut + threadIdx.x * 256;
                    unsigned total = 0, last = 0; int nonzero = 0;
                    for (int i = 0; i < 256; ++i) if (h[i]) { total += h[i]; last = h[i]; ++nonzero; }
                    const unsigned step = (total - last) / 255u;
                    if (nonzero <= 1 || step == 0) { for (int i = 0; i < 256; ++i) l[i] = (unsigned char)i; }
                    else { unsigned nn = step / 2; for (int i = 0; i < 256; ++i) { const unsigned q = nn / step; l[i] = (unsigned char)(q > 255u ? 255u : q); nn += h[i]; } }
                }
                __syncthreads();
                aug_apply_lut(cur, size, C, lut);
            } break;
            case DCV_AUG_COLOR:
                // ImageEnhance.Color: blend(grey image, img, a); a grey-level image is its own degenerate
                if (C == 3) for (int p = threadIdx.x; p < H * W; p += AUG_THREADS) {
                    unsigned char* q = cur + (size_t)p * 3;
                    const int g = aug_gray(q[0], q[1], q[2]);
                    q[0] = (unsigned char)aug_blend(g, q[0], fp, inside); q[1] = (unsigned char)aug_blend(g, q[1], fp, inside); q[2] = (unsigned char)aug_blend(g, q[2], fp, inside);
                }
                break;
            case DCV_AUG_CONTRAST: {
                // ImageEnhance.Contrast: blend(constant image of the mean grey level int(sum / count + 0.5), img, a)
                if (threadIdx.x == 0) gsum = 0;
                __syncthreads();
                unsigned part = 0;
                if (C == 3) { for (int p = threadIdx.x; p < H * W; p += AUG_THREADS) { const unsigned char* q = cur + (size_t)p * 3; part += (unsigned)aug_gray(q[0], q[1], q[2]); } }
                else { for (int p = threadIdx.x; p < H * W; p += AUG_THREADS) part += cur[p]; }
                atomicAdd(&gsum, part);
                __syncthreads();
                const int mean = (int)aug_dadd(aug_ddiv((double)gsum, (double)(H * W)), 0.5);
                for (int i = threadIdx.x; i < size; i += AUG_THREADS) cur[i] = (unsigned char)aug_blend(mean, cur[i], fp, inside);
            } break;
            case DCV_AUG_SHARPNESS: { aug_sharpness(cur, oth, H, W, C, fp); unsigned char* t = cur; cur = oth; oth = t; } break;
            case DCV_AUG_TRANSLATE_X: case DCV_AUG_TRANSLATE_Y: case DCV_AUG_ROTATE: case DCV_AUG_SHEAR_X: case DCV_AUG_SHEAR_Y: {
                aug_affine(cur, oth, H, W, C, rp.m[chain][k]); unsigned char* t = cur; cur = oth; oth = t;
            } break;
            default: break;
        }
        __syncthreads();
    }
    if (cur != bufA) for (int i = threadIdx.x; i < size; i += AUG_THREADS) bufA[i] = cur[i];
}

// out = normalize((1 - m) * x + m * sum_i w_i * chain_i): one thread per pixel
template <int DST_NCHW>
__global__ __launch_bounds__(256) void augmix_mix_kernel(const unsigned char* __restrict__ src, const unsigned char* __restrict__ scratch, float* __restrict__ dst,
                                                          int N, int H, int W, int C, const dcv_augmix_recipe* __restrict__ recipes,
                                                          const float* __restrict__ mean, const float* __restrict__ inv_std) {
    const int hw = H * W;
    const size_t pixels = (size_t)N * hw;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < pixels; pix += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(pix / hw), q = (int)(pix - (size_t)n * hw);
        const dcv_augmix_recipe& rp = recipes[n];
        float mixv[3] = {0.f, 0.f, 0.f};
        for (int ch = 0; ch < DCV_AUG_MAX_CHAINS; ++ch) {
            const float wgt = rp.weight[ch];
            if (wgt == 0.f) continue;
            const unsigned char* pp = scratch + ((size_t)(n * DCV_AUG_MAX_CHAINS + ch) * 2) * hw * C + (size_t)q * C;
            for (int c = 0; c < C; ++c) mixv[c] = aug_fadd(mixv[c], aug_fmul(wgt, (float)pp[c]));
        }
        const unsigned char* po = src + pix * C;
        for (int c = 0; c < C; ++c) {
            const float v = aug_fadd(aug_fmul(1.f - rp.mix, (float)po[c]), aug_fmul(rp.mix, mixv[c]));
            const float o = (v * (1.f / 255.f) - mean[c]) * inv_std[c];
            if (DST_NCHW) dst[((size_t)n * C + c) * hw + q] = o; else dst[pix * C + c] = o;
        }
    }
}

}  // namespace

extern "C" {

size_t dcv_images_u8_augment_scratch_bytes(int n, int h, int w, int c) {
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return 0;
    return dcv_align_up((size_t)n * DCV_AUG_MAX_CHAINS * 2 * h * w * c, 256);
}

int dcv_images_u8_augment_normalize(const unsigned char* src, float* dst, int n, int h, int w, int c, int dst_layout, const dcv_augmix_recipe* recipes,
                                    const float* mean, const float* inv_std, void* scratch, size_t scratch_bytes, void* stream) {
    DCV_CHECK_ARG(src && dst && recipes && mean && inv_std && scratch && n > 0 && n <= 65535 && h > 0 && w > 0 && (c == 1 || c == 3)
                  && (dst_layout == DCV_LAYOUT_NCHW || dst_layout == DCV_LAYOUT_NHWC));
    if (scratch_bytes < dcv_images_u8_augment_scratch_bytes(n, h, w, c)) return DCV_ERR_WORKSPACE;
    hipStream_t st = dcv_stream(stream);
    unsigned char* sc = reinterpret_cast<unsigned char*>(scratch);
    hipLaunchKernelGGL(augmix_chain_kernel, dim3(DCV_AUG_MAX_CHAINS, (unsigned)n), dim3(AUG_THREADS), 0, st, src, sc, h, w, c, recipes);
    const size_t pixels = (size_t)n * h * w;
    if (dst_layout == DCV_LAYOUT_NCHW) hipLaunchKernelGGL((augmix_mix_kernel<1>), dim3(dcv_stream_grid(pixels, 256)), dim3(256), 0, st, src, sc, dst, n, h, w, c, recipes, mean, inv_std);
    else hipLaunchKernelGGL((augmix_mix_kernel<0>), dim3(dcv_stream_grid(pixels, 256)), dim3(256), 0, st, src, sc, dst, n, h, w, c, recipes, mean, inv_std);
    DCV_RETURN_LAUNCH();
}

}  // extern "C"
