// slab_sum.h — the per-block body of the ordered slab sum, shared by slab_reduce_kernel (elementwise.hip) and the optimizer's tail launch (conv2d.hip:
// adamw_prep_slabs_kernel), so that both produce the same bits: the result only depends on (segment shape, entry), not on the launch.
// Block = 1024 threads: (1 << ws) entries x (1024 >> ws) slab lanes; a thread adds chunks / lanes slabs, the slab lanes of a wave are merged by a fixed xor butterfly,
// the 16 waves in order through LDS.
#ifndef DEEPCV_SLAB_SUM_H
#define DEEPCV_SLAB_SUM_H
#include <hip/hip_runtime.h>
#include "../../include/deepcv_hip.h"
#define SLAB_W 8
#define SLAB_WIDE 32
#define SLAB_WIDE_COUNT 1024
struct SlabTable {
    int count;
    int wshift[DCV_MAX_SLAB_SEGMENTS];      // log2 of the segment's entries per block (3 or 5)
    int block_start[DCV_MAX_SLAB_SEGMENTS + 1];
    dcv_slab_segment seg[DCV_MAX_SLAB_SEGMENTS];
};
// `local_block`: the block's index inside its segment.  Returns true on the threads which hold a finished entry: `value` = the sum, `di` = its index in dst.
static __device__ __forceinline__ bool dcv_slab_block_sum(const dcv_slab_segment& g, int ws, int local_block, float (*red)[SLAB_WIDE], float& value, size_t& di) {
    const int W = 1 << ws, lanes = 1024 >> ws;
    const int wl = threadIdx.x & (W - 1), cl = threadIdx.x >> ws, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t wi = ((size_t)local_block << ws) + wl;
    const float* __restrict__ src = g.src;
    float s = 0.f;
    if (wi < g.count) {
#pragma unroll 4
        for (int c = cl; c < g.chunks; c += lanes) s += src[(size_t)c * g.stride + wi];
    }
    for (int off = W; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
    if (lane < W) red[wave][lane] = s;
    __syncthreads();
    if ((int)threadIdx.x < W && wi < g.count) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
        di = wi;
        if (g.tr_a > 0) {                                                                // [o][tr_a][tr_b] -> [o][tr_b][tr_a]
            const size_t ab = (size_t)g.tr_a * g.tr_b, o = wi / ab, rem = wi - o * ab;
            const size_t pp = rem / g.tr_b, c = rem - pp * g.tr_b;
            di = o * ab + c * g.tr_a + pp;
        }
        value = v;
        return true;
    }
    return false;
}
#endif
