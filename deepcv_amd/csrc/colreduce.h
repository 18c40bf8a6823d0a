// colreduce.h — per-channel ("column") reductions of an NHWC activation viewed as [rows][C], the shared building block of the
// BatchNorm / GroupNorm / bias-gradient kernels.  HBM-bound: every element is read once with coalesced (float4 when C % 4 == 0)
// loads; partial sums are kept per thread in fp32 over a short run of rows, merged across the block through LDS in a FIXED
// order and promoted to fp64 there, so results are bit-reproducible and free of E[x^2]-E[x]^2 cancellation at N*H*W ~ 5e5.
#pragma once
#include "common.h"

#define DCV_CR_THREADS 256

// Functor contract:  f.template load<VEC>(size_t elem_index, int c, float (&q)[NQ][VEC])
//   elem_index = row * C + c is the flat offset of the first of VEC consecutive channels.
//
// Reduces rows [row_begin, row_end) for all channels; on return (after an internal __syncthreads) `out[q * C + c]` (LDS or global,
// double) holds the block's sum for quantity q / channel c.  `scratch` must hold NQ * DCV_CR_THREADS * VEC floats of LDS.
// Must be called by all DCV_CR_THREADS threads of the block.
// Second half of dcv_colreduce_block for CV <= DCV_CR_THREADS: merges the per-thread partial sums acc[q][v] of thread (rsub = t / CV, cv = t % CV) over
// the row sub-streams in a fixed order; `out[q * C + c]` (LDS or global, double) receives the block's sums.  Must be called by all threads.
template <int NQ, int VEC>
__device__ __forceinline__ void dcv_colreduce_merge(const float (&acc)[NQ][VEC], int C, float* scratch, double* out) {
    const int t = threadIdx.x;
    const int CV = C / VEC;
    const int rows_per_iter = DCV_CR_THREADS / CV;
        __syncthreads();                          // scratch may still be in use by a previous call
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int v = 0; v < VEC; ++v) scratch[(q * VEC + v) * DCV_CR_THREADS + t] = acc[q][v];
        __syncthreads();
        const int NO = NQ * C;
        if (NO <= DCV_CR_THREADS / 2) {
            // few outputs (small C), many row sub-streams: ordered two-level merge.  Thread (part, o) sums a contiguous run of sub-streams of output o,
            // then thread o sums the `P` run totals in order -> a serial chain of rows_per_iter / P + P adds instead of rows_per_iter.
            __shared__ double cr_part[DCV_CR_THREADS];
            const int P = DCV_CR_THREADS / NO;
            const int part = t / NO, o = t - part * NO;
            if (part < P) {
                const int q = o / C, c = o - q * C;
                const int cvi = c / VEC, v = c - cvi * VEC;
                const int chunk = (rows_per_iter + P - 1) / P;
                const int rs0 = part * chunk;
                int rs1 = rs0 + chunk; if (rs1 > rows_per_iter) rs1 = rows_per_iter;
                const float* sp = scratch + (q * VEC + v) * DCV_CR_THREADS + cvi;
                double s = 0.0;
                for (int rs = rs0; rs < rs1; ++rs) s += (double)sp[rs * CV];
                cr_part[part * NO + o] = s;
            }
            __syncthreads();
            if (t < NO) {
                double s = 0.0;
                for (int pp = 0; pp < P; ++pp) s += cr_part[pp * NO + t];
                out[t] = s;                                   // t = q * C + c
            }
        } else {
            // ordered merge over the row sub-streams: thread (q, c) sums rsub = 0..rows_per_iter-1
            for (int idx = t; idx < NO; idx += DCV_CR_THREADS) {
                const int q = idx / C, c = idx % C;
                const int cvi = c / VEC, v = c % VEC;
                double s = 0.0;
                for (int rs = 0; rs < rows_per_iter; ++rs) s += (double)scratch[(q * VEC + v) * DCV_CR_THREADS + rs * CV + cvi];
                out[q * C + c] = s;
            }
        }
        __syncthreads();
}

template <int NQ, int VEC, typename F>
__device__ __forceinline__ void dcv_colreduce_block(const F& f, size_t row_begin, size_t row_end, int C, float* scratch, double* out) {
    const int t = threadIdx.x;
    const int CV = C / VEC;                       // channel-vectors per row (C % VEC == 0 guaranteed by the caller)
    if (CV <= DCV_CR_THREADS) {
        const int rows_per_iter = DCV_CR_THREADS / CV;
        const int active = rows_per_iter * CV;
        const int cv = t % CV, rsub = t / CV;
        float acc[NQ][VEC];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[q][v] = 0.f;
        if (t < active) {
            for (size_t r = row_begin + rsub; r < row_end; r += rows_per_iter) {
                float qv[NQ][VEC];
                f.template load<VEC>(r * (size_t)C + (size_t)cv * VEC, cv * VEC, qv);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[q][v] += qv[q][v];
            }
        }
        dcv_colreduce_merge<NQ, VEC>(acc, C, scratch, out);
    } else {
        // wide rows: each thread owns channel-vectors cv = t, t + 256, ... and walks every row of the range
        for (int cv = t; cv < CV; cv += DCV_CR_THREADS) {
            double acc[NQ][VEC];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[q][v] = 0.0;
            for (size_t r = row_begin; r < row_end; ++r) {
                float qv[NQ][VEC];
                f.template load<VEC>(r * (size_t)C + (size_t)cv * VEC, cv * VEC, qv);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[q][v] += (double)qv[q][v];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int v = 0; v < VEC; ++v) out[q * C + cv * VEC + v] = acc[q][v];
        }
        __syncthreads();
    }
}

// Row partition used by every two-stage reduction: block b of nb handles rows [b*rpb, min(rows,(b+1)*rpb)).
static inline unsigned dcv_colreduce_blocks(size_t rows, int C) {
    // 4096 elements per block keep per-thread fp32 runs short (<= 16 adds) while giving >= 2 blocks per CU at the stock sizes
    size_t elems_per_block = 4096;
    size_t rows_per_block = elems_per_block / (size_t)C;
    if (rows_per_block < 1) rows_per_block = 1;
    size_t nb = dcv_ceil_div(rows, rows_per_block);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}
