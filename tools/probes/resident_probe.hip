// resident_probe.hip — is an LDS-resident network worth building?  One persistent launch, 256 blocks x 1024 threads, each block keeps TWO samples of the
// stock model's 32x32x4 activations in LDS and runs L layers  conv5x5(4->4) + ReLU -> BatchNorm (batch statistics: all-gather of the blocks' sums through
// 8-byte {tag, value} granules) -> GroupNorm(4 groups) -> next layer's padded input tile, writing y and out to HBM like the training forward does.
// Measures the per-layer time against the per-layer launches of the step (conv 15 us + norm chain 6.7 us) and checks the result against a CPU reference.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/resident_probe.hip -o gpurun_out/resident_probe && gpurun_out/resident_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NB = 256, NT = 1024, SPB = 2, H = 32, W = 32, C = 4, K = 5, PAD = 2, XH = H + 2 * PAD, XW = W + 2 * PAD;
constexpr int MAXL = 16;
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(4))) const float cfloat;     // constant address space: wave-uniform loads become s_load (scalar cache)
#define AS_CONST(p) ((cfloat*)(unsigned long long)(p))

struct Args {
    const float* x;          // [N][H][W][4]
    const float* wt;         // [L][K][K][4][4]  (ky, kx, ci, co)
    const float* bias;       // [L][4]
    const float* gamma; const float* beta; const float* ggamma; const float* gbeta;   // [L][4] each
    float* y;                // [L][N][H][W][4]
    float* out;              // [L][N][H][W][4]
    u64* gran;               // [MAXL][8][NB]
    unsigned* gen;           // launch generation
    unsigned* err;
    int L, N, do_sync, do_store;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(NT) void resident_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* xin = reinterpret_cast<float4*>(smem);                       // [SPB][XH][XW]
    float4* yt = xin + SPB * XH * XW;                                    // [SPB][H*W]
    float* scr = reinterpret_cast<float*>(yt + SPB * H * W);             // scratch: 1024 floats
    __shared__ float s_samp[SPB][8];                                     // per-sample sums of y, y^2
    __shared__ double s_part[8][16];
    __shared__ float s_ab[8];                                            // BatchNorm a[4], b[4]
    __shared__ float s_ST[SPB][8];
    __shared__ int s_fail;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int s = t >> 9, p = t & 511, col = p & 31, rp = p >> 5;        // sample, column, row pair
    const int n = blockIdx.x * SPB + s;
    const unsigned gen = __hip_atomic_load(a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0) s_fail = 0;
    // zero the tiles' halos once, stage the two samples
    for (int i = t; i < SPB * XH * XW; i += NT) xin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    {
        const float4* xs = reinterpret_cast<const float4*>(a.x) + (size_t)n * H * W;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = rp * 2 + k;
            xin[(s * XH + r + PAD) * XW + col + PAD] = xs[r * W + col];
        }
    }
    __syncthreads();
    for (int l = 0; l < a.L; ++l) {
        // ---- convolution: two vertically adjacent output pixels per thread, input rows rotate through registers, filter row from the scalar cache
        float acc[2][4];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[k][c] = 0.f;
        const float4* xs = xin + (s * XH + rp * 2) * XW + col;
        cfloat* wl = AS_CONST(a.wt + (size_t)l * K * K * 16);
        float4 old[K], nw[K];
#pragma unroll
        for (int sx = 0; sx < K; ++sx) old[sx] = xs[sx];
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll
            for (int sx = 0; sx < K; ++sx) nw[sx] = xs[(ky + 1) * XW + sx];
            cfloat* wr = wl + ky * K * 16;
#pragma unroll
            for (int sx = 0; sx < K; ++sx) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[0][c] = fmaf(old[sx].x, wr[sx * 16 + c], acc[0][c]);
                    acc[0][c] = fmaf(old[sx].y, wr[sx * 16 + 4 + c], acc[0][c]);
                    acc[0][c] = fmaf(old[sx].z, wr[sx * 16 + 8 + c], acc[0][c]);
                    acc[0][c] = fmaf(old[sx].w, wr[sx * 16 + 12 + c], acc[0][c]);
                    acc[1][c] = fmaf(nw[sx].x, wr[sx * 16 + c], acc[1][c]);
                    acc[1][c] = fmaf(nw[sx].y, wr[sx * 16 + 4 + c], acc[1][c]);
                    acc[1][c] = fmaf(nw[sx].z, wr[sx * 16 + 8 + c], acc[1][c]);
                    acc[1][c] = fmaf(nw[sx].w, wr[sx * 16 + 12 + c], acc[1][c]);
                }
            }
#pragma unroll
            for (int sx = 0; sx < K; ++sx) old[sx] = nw[sx];
        }
        float sum[4], sq[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { sum[c] = 0.f; sq[c] = 0.f; }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { o[c] = acc[k][c] + AS_CONST(a.bias)[l * 4 + c]; o[c] = o[c] > 0.f ? o[c] : 0.f; sum[c] += o[c]; sq[c] += o[c] * o[c]; }
            const int r = rp * 2 + k;
            const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            yt[s * H * W + r * W + col] = ov;
            if (a.do_store) reinterpret_cast<float4*>(a.y)[((size_t)l * a.N + n) * H * W + r * W + col] = ov;
        }
        // ---- per-sample sums (8 waves per sample), block totals -> granules
#pragma unroll
        for (int c = 0; c < 4; ++c) { sum[c] = wave_sum(sum[c]); sq[c] = wave_sum(sq[c]); }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { scr[wave * 8 + c] = sum[c]; scr[wave * 8 + 4 + c] = sq[c]; }
        }
        __syncthreads();
        if (t < 16) {
            const int ss = t >> 3, q = t & 7;
            float v = 0.f;
            for (int w8 = 0; w8 < 8; ++w8) v += scr[(ss * 8 + w8) * 8 + q];
            s_samp[ss][q] = v;
        }
        __syncthreads();
        const unsigned epoch = gen * MAXL + l + 1;
        if (a.do_sync) {
            if (t < 8) {
                const float tot = s_samp[0][t] + s_samp[1][t];
                __hip_atomic_store((gu64*)(a.gran + ((size_t)l * 8 + t) * NB + blockIdx.x), ((u64)epoch << 32) | __float_as_uint(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // every thread sweeps two granules (8 quantities x 256 producers)
            float gv[2];
            bool ok = false;
            for (unsigned spins = 0; spins < (1u << 20); ++spins) {
                ok = true;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u64 g = __hip_atomic_load((gu64*)(a.gran + (size_t)l * 8 * NB + t + k * NT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gv[k] = __uint_as_float((unsigned)g);
                    ok &= (unsigned)(g >> 32) == epoch;
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!__all(ok) && lane == 0) s_fail = 1;
            float* arr = reinterpret_cast<float*>(xin);                  // the input tile is dead until the apply step: [8][256] gathered sums
            arr[t] = gv[0]; arr[t + NT] = gv[1];
            __syncthreads();
            if (s_fail) { if (t == 0) atomicAdd(a.err, 1u); return; }
            if (t < 128) {
                const int q = t >> 4, j = t & 15;
                double v = 0.0;
                for (int i = 0; i < 16; ++i) v += (double)arr[q * 256 + j * 16 + i];
                s_part[q][j] = v;
            }
            __syncthreads();
        } else {
            if (t < 128) { const int q = t >> 4, j = t & 15; s_part[q][j] = j == 0 ? (double)(s_samp[0][q] + s_samp[1][q]) * NB : 0.0; }
            __syncthreads();
        }
        if (t < 4) {
            double s1 = 0.0, s2 = 0.0;
            for (int j = 0; j < 16; ++j) { s1 += s_part[t][j]; s2 += s_part[4 + t][j]; }
            const double cnt = (double)a.N * H * W, mean = s1 / cnt;
            double var = s2 / cnt - mean * mean; if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + 1e-5));
            const float sc = a.gamma[l * 4 + t] * rstd;
            s_ab[t] = sc; s_ab[4 + t] = a.beta[l * 4 + t] - (float)mean * sc;
        }
        __syncthreads();
        if (t < 8) {   // GroupNorm with one channel per group: statistics of v = a y + b of (sample, channel)
            const int ss = t >> 2, c = t & 3;
            const double hw = (double)(H * W), ad = (double)s_ab[c], bd = (double)s_ab[4 + c], s1 = (double)s_samp[ss][c], s2 = (double)s_samp[ss][4 + c];
            const double sv = ad * s1 + hw * bd, svv = ad * ad * s2 + 2.0 * ad * bd * s1 + hw * bd * bd;
            const double mean = sv / hw; double var = svv / hw - mean * mean; if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + 1e-5));
            const float gg = a.ggamma[l * 4 + c], gb = a.gbeta[l * 4 + c];
            s_ST[ss][c] = s_ab[c] * rstd * gg; s_ST[ss][4 + c] = (s_ab[4 + c] - (float)mean) * rstd * gg + gb;
        }
        __syncthreads();
        // ---- apply: out = S y + T into the next layer's padded tile (interior) and to HBM
        {
            const float S0 = s_ST[s][0], S1 = s_ST[s][1], S2 = s_ST[s][2], S3 = s_ST[s][3], T0 = s_ST[s][4], T1 = s_ST[s][5], T2 = s_ST[s][6], T3 = s_ST[s][7];
            // (the gather array aliased the tile: rewrite the halo rows / columns it covered)
            if (a.do_sync) for (int i = t; i < 2048 / 4; i += NT) xin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = rp * 2 + k;
                const float4 v = yt[s * H * W + r * W + col];
                const float4 o = make_float4(v.x * S0 + T0, v.y * S1 + T1, v.z * S2 + T2, v.w * S3 + T3);
                xin[(s * XH + r + PAD) * XW + col + PAD] = o;
                if (a.do_store) reinterpret_cast<float4*>(a.out)[((size_t)l * a.N + n) * H * W + r * W + col] = o;
            }
        }
        __syncthreads();
    }
    // the last block to finish does not matter: every block read `gen` before its first publication, and the next launch is stream-ordered after this one
    if (blockIdx.x == 0 && t == 0) __hip_atomic_store(a.gen, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static void cpu_ref(const std::vector<float>& x, const std::vector<float>& wt, const std::vector<float>& bias, const std::vector<float>& g, const std::vector<float>& b,
                    const std::vector<float>& gg, const std::vector<float>& gb, int N, int L, std::vector<float>& out) {
    std::vector<float> cur = x, y((size_t)N * H * W * 4);
    for (int l = 0; l < L; ++l) {
        double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
        for (int n = 0; n < N; ++n)
            for (int r = 0; r < H; ++r)
                for (int c0 = 0; c0 < W; ++c0)
                    for (int co = 0; co < 4; ++co) {
                        double acc = bias[l * 4 + co];
                        for (int ky = 0; ky < K; ++ky)
                            for (int kx = 0; kx < K; ++kx) {
                                const int ri = r + ky - PAD, ci = c0 + kx - PAD;
                                if (ri < 0 || ri >= H || ci < 0 || ci >= W) continue;
                                for (int c = 0; c < 4; ++c) acc += (double)cur[(((size_t)n * H + ri) * W + ci) * 4 + c] * wt[(((size_t)l * K + ky) * K + kx) * 16 + c * 4 + co];
                            }
                        const float v = acc > 0 ? (float)acc : 0.f;
                        y[(((size_t)n * H + r) * W + c0) * 4 + co] = v; s1[co] += v; s2[co] += (double)v * v;
                    }
        for (int n = 0; n < N; ++n)
            for (int co = 0; co < 4; ++co) {
                const double cnt = (double)N * H * W, mean = s1[co] / cnt, var = s2[co] / cnt - mean * mean, rstd = 1.0 / sqrt(var + 1e-5);
                double m = 0, q = 0;
                for (int i = 0; i < H * W; ++i) { const double v = (y[((size_t)n * H * W + i) * 4 + co] - mean) * rstd * g[l * 4 + co] + b[l * 4 + co]; m += v; q += v * v; }
                m /= H * W; q = q / (H * W) - m * m; const double gr = 1.0 / sqrt(q + 1e-5);
                for (int i = 0; i < H * W; ++i) {
                    const double v = (y[((size_t)n * H * W + i) * 4 + co] - mean) * rstd * g[l * 4 + co] + b[l * 4 + co];
                    cur[((size_t)n * H * W + i) * 4 + co] = (float)((v - m) * gr * gg[l * 4 + co] + gb[l * 4 + co]);
                }
            }
    }
    out = cur;
}

int main(int argc, char** argv) {
    const int N = NB * SPB;
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<float> hx((size_t)N * H * W * 4), hw((size_t)MAXL * K * K * 16), hb(MAXL * 4), hg(MAXL * 4), hbe(MAXL * 4), hgg(MAXL * 4), hgb(MAXL * 4);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.15f;
    for (int i = 0; i < MAXL * 4; ++i) { hb[i] = rnd() * 0.1f; hg[i] = 1.f + 0.2f * rnd(); hbe[i] = 0.1f * rnd(); hgg[i] = 1.f + 0.2f * rnd(); hgb[i] = 0.1f * rnd(); }
    float *dx, *dw, *db, *dg, *dbe, *dgg, *dgb, *dy, *dout; u64* gran; unsigned *gen, *err;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, 64 * 4)); CK(hipMalloc(&dg, 64 * 4)); CK(hipMalloc(&dbe, 64 * 4));
    CK(hipMalloc(&dgg, 64 * 4)); CK(hipMalloc(&dgb, 64 * 4)); CK(hipMalloc(&dy, (size_t)MAXL * hx.size() * 4)); CK(hipMalloc(&dout, (size_t)MAXL * hx.size() * 4));
    CK(hipMalloc(&gran, (size_t)MAXL * 8 * NB * 8)); CK(hipMalloc(&gen, 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(gran, 0, (size_t)MAXL * 8 * NB * 8)); CK(hipMemset(gen, 0, 4)); CK(hipMemset(err, 0, 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, hg.data(), 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbe, hbe.data(), 64 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgg, hgg.data(), 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgb, hgb.data(), 64 * 4, hipMemcpyHostToDevice));
    const size_t smem = (size_t)SPB * XH * XW * 16 + (size_t)SPB * H * W * 16 + 1024 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, resident_kernel, NT, smem));
    printf("dynamic LDS %zu B, occupancy API: %d blocks per CU\n", smem, occ);
    Args a{dx, dw, db, dg, dbe, dgg, dgb, dy, dout, gran, gen, err, 3, N, 1, 1};
    // correctness: 3 layers against the CPU reference
    hipLaunchKernelGGL(resident_kernel, dim3(NB), dim3(NT), smem, st, a);
    CK(hipStreamSynchronize(st));
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<float> got((size_t)N * H * W * 4), ref;
    CK(hipMemcpy(got.data(), dout + (size_t)2 * hx.size(), got.size() * 4, hipMemcpyDeviceToHost));
    {
        // the reference only needs the batch statistics of all samples: run it on the whole batch (a few seconds)
        cpu_ref(hx, hw, hb, hg, hbe, hgg, hgb, N, 3, ref);
        double md = 0; for (size_t i = 0; i < got.size(); ++i) md = fmax(md, fabs((double)got[i] - ref[i]));
        printf("3 layers: timeouts %u, max |gpu - cpu| = %.3g\n", herr, md);
    }
    // timing: L layers per launch, graph of 20 launches
    for (int mode = 0; mode < 3; ++mode) {
        for (int L : {1, 2, 4, 8, 16}) {
            Args b = a; b.L = L; b.do_sync = mode != 1; b.do_store = mode != 2;
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(resident_kernel, dim3(NB), dim3(NT), smem, st, b);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d (%s) L=%2d: %7.2f us per launch, %6.2f us per layer\n", mode, mode == 0 ? "sync+store" : mode == 1 ? "no sync" : "no HBM stores", L, ms * 1e3 / 100, ms * 1e3 / 100 / L);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("timeouts after timing: %u\n", herr);
    return 0;
}
