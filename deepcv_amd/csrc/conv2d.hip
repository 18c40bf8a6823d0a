// conv2d.hip — public C-ABI of the convolution family; validates the descriptor and dispatches to a back-end:
//   * MFMA implicit GEMM (conv2d_mfma.hip) when the layer is wide enough to feed the matrix cores,
//   * direct convolution (conv2d_direct.hip) otherwise (the stock small-channel layers, and every shape as fallback).
// Replaces torch.nn.Conv2d (reference src/deepcv/meta/submodule_creators.py:251) forward and autograd backward.
#include "conv2d_internal.h"
#include "optim_update.h"
#include "slab_sum.h"

namespace {

inline bool desc_ok(const dcv_conv2d_desc* d) {
    if (!d) return false;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cout <= 0 || d->r <= 0 || d->s <= 0) return false;
    if (d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h <= 0 || d->dil_w <= 0) return false;
    const int ho = (d->h + 2 * d->pad_h - d->dil_h * (d->r - 1) - 1) / d->stride_h + 1;
    const int wo = (d->w + 2 * d->pad_w - d->dil_w * (d->s - 1) - 1) / d->stride_w + 1;
    if (ho != d->ho || wo != d->wo || ho <= 0 || wo <= 0) return false;
    if (d->act < DCV_ACT_NONE || d->act > DCV_ACT_TANH) return false;
    if (d->x_layout != DCV_LAYOUT_NHWC && d->x_layout != DCV_LAYOUT_NCHW) return false;
    if (d->math != DCV_MATH_FP32 && d->math != DCV_MATH_BF16) return false;
    return true;
}

// ONE launch re-laying out up to DCV_PREP_MAX filters (blockIdx.y = filter)
__global__ __launch_bounds__(256) void prep_multi_kernel(dcv_prep_list l) {
    const dcv_prep_entry& e = l.e[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < e.total; i += gridDim.x * blockDim.x) e.dst[i] = dcv_prep_element(e, i);
}

// AdamW update of a whole tensor (or of a group of rows of a Linear weight) by ONE block, then — behind a block barrier — every prepared layout of that
// tensor gathered from the updated values: dcv_adamw_step + dcv_conv2d_prepare_filters of a small parameter set in one launch.
static_assert(sizeof(dcv_prep_entry) == 88, "dcv_adamw_prep_table reserves 88 bytes per prepared layout");
__global__ __launch_bounds__(1024) void adamw_prep_kernel(dcv_adamw_prep_table t, const float* __restrict__ hyper) {
    const dcv_opt_item& it = t.item[blockIdx.x];
    const dcv_adamw_scalars sc = dcv_adamw_load_scalars(hyper);
    // the first two prepared layouts' descriptors are requested with the update's operands (scalar loads: they sit in registers across the barrier) — read
    // behind the barrier they were one more memory round trip of a kernel which is nothing but three of them
    const dcv_prep_entry* preps = reinterpret_cast<const dcv_prep_entry*>(t.prep);
    dcv_prep_entry e0{}, e1{};
    if (it.prep_count > 0) e0 = preps[it.prep_first];
    if (it.prep_count > 1) e1 = preps[it.prep_first + 1];
    for (int i = threadIdx.x; i < it.numel; i += 1024) dcv_adamw_update<false>(sc, it.param, it.grad, it.state1, it.state2, nullptr, i);
    if (it.prep_count == 0) return;                                                  // block-uniform
    __syncthreads();                                                                 // the block's stores are visible to its own gathers below
    for (int i = threadIdx.x; i < e0.total; i += 1024) e0.dst[i] = dcv_prep_element(e0, i);
    if (it.prep_count > 1) for (int i = threadIdx.x; i < e1.total; i += 1024) e1.dst[i] = dcv_prep_element(e1, i);
    for (int q = 2; q < it.prep_count; ++q) {
        const dcv_prep_entry& e = preps[it.prep_first + q];
        for (int i = threadIdx.x; i < e.total; i += 1024) e.dst[i] = dcv_prep_element(e, i);
    }
}

// Round 4: the ordered slab sums of the step, AdamW and the filter re-layout in ONE launch (the stock step's two trailing launches -> one).  Blocks [0, slab blocks):
// the body of slab_reduce_kernel (slab_sum.h: same summation tree, same bits); a thread which finishes an entry stores the gradient, updates the parameter the entry
// belongs to (`item_of`: the optimizer item whose gradient tensor the segment writes) and SCATTERS the new value into every prepared layout of that tensor
// (dcv_prep_scatter_index) — no second kernel, no barrier between the sums and the update, no gather from whole tensors.  Blocks behind them: the items no slab
// segment writes (BatchNorm parameters, the fully connected layer's weight: their gradients are final when the backward launch ends), a block per item as in
// adamw_prep_kernel but scattering.  Same arithmetic as dcv_slab_reduce + dcv_adamw_step + dcv_conv2d_prepare_filters, bit for bit.
struct TailTable { SlabTable slabs; int slab_blocks; short item_of[DCV_MAX_SLAB_SEGMENTS]; short direct_count; short direct[DCV_OPT_MAX_ITEMS]; };
__device__ __forceinline__ void tail_scatter(const dcv_adamw_prep_table& t, const dcv_opt_item& it, int off, float pnew) {
    const dcv_prep_entry* preps = reinterpret_cast<const dcv_prep_entry*>(t.prep);
    for (int q = 0; q < it.prep_count; ++q) {
        const dcv_prep_entry& e = preps[it.prep_first + q];
        const int idx = dcv_prep_scatter_index(e, off);
        if (idx >= 0) e.dst[idx] = pnew;
    }
}
__device__ __forceinline__ void tail_update(const dcv_adamw_prep_table& t, const dcv_adamw_scalars& sc, const dcv_opt_item& it, int off, float g) {
    tail_scatter(t, it, off, dcv_adamw_update_value<false>(sc, it.param, g, it.state1, it.state2, nullptr, off));
}
__global__ __launch_bounds__(1024) void adamw_prep_slabs_kernel(dcv_adamw_prep_table t, TailTable tt, const float* __restrict__ hyper) {
    __shared__ float red[16][SLAB_WIDE];
    const dcv_adamw_scalars sc = dcv_adamw_load_scalars(hyper);
    if ((int)blockIdx.x < tt.slab_blocks) {
        int si = 0;
        while (si + 1 < tt.slabs.count && (int)blockIdx.x >= tt.slabs.block_start[si + 1]) ++si;
        const dcv_slab_segment& g = tt.slabs.seg[si];
        // the entry this thread will finish (threads < W of the block: slab_sum.h) and its parameter / moments, requested NOW: the update's loads travel with the
        // sum's instead of behind them (one memory round trip of a kernel which is two)
        const int ws = tt.slabs.wshift[si], lb = (int)blockIdx.x - tt.slabs.block_start[si], ii = tt.item_of[si];
        float p0 = 0.f, m0 = 0.f, v0 = 0.f; int off = -1;
        if (ii >= 0 && (int)threadIdx.x < (1 << ws)) {
            const size_t wi = ((size_t)lb << ws) + threadIdx.x;
            if (wi < g.count) {
                size_t di = wi;
                if (g.tr_a > 0) { const size_t ab = (size_t)g.tr_a * g.tr_b, o = wi / ab, rem = wi - o * ab, pp = rem / g.tr_b, c = rem - pp * g.tr_b; di = o * ab + c * g.tr_a + pp; }
                const dcv_opt_item& it = t.item[ii];
                off = (int)((g.dst + di) - it.grad);
                p0 = it.param[off]; m0 = it.state1[off]; v0 = it.state2[off];
            }
        }
        float v; size_t di;
        if (dcv_slab_block_sum(g, ws, lb, red, v, di)) {
            g.dst[di] = v;
            if (ii >= 0) {
                const dcv_opt_item& it = t.item[ii];
                tail_scatter(t, it, off, dcv_adamw_update_loaded<false>(sc, it.param, v, it.state1, it.state2, nullptr, off, p0, m0, v0));
            }
        }
        return;
    }
    const dcv_opt_item& it = t.item[tt.direct[(int)blockIdx.x - tt.slab_blocks]];
    for (int i = threadIdx.x; i < it.numel; i += 1024) tail_update(t, sc, it, i, it.grad[i]);
}

// op 2: the weight [O][C * H * W] of a Linear which follows a Flatten of [N][C][H][W] activations, re-laid as [O][H * W][C] so that the Linear reads the
// NHWC activations in place (no NHWC -> NCHW transposition of the batch in either direction).  Descriptor: h, w, cin of the activations, cout = O,
// r = h, s = w, no padding (the Linear seen as a full-image convolution); same element mapping as the MFMA kernels' K-major filter rows.
inline bool linear_desc(const dcv_conv2d_desc* d) {
    return d->r == d->h && d->s == d->w && d->pad_h == 0 && d->pad_w == 0 && d->stride_h == 1 && d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1;
}

inline int prep_entries(const dcv_conv2d_desc* d, int op, dcv_prep_entry* out) {
    if (op == 2) {
        if (!linear_desc(d)) return 0;
        dcv_prep_entry e{};
        e.kind = 1; e.kcout = d->cout; e.kcin = d->cin; e.R = d->r; e.S = d->s; e.w_cin = d->cin; e.dgrad = 0;
        e.dst_off = 0; e.total = d->cout * d->cin * d->r * d->s;
        out[0] = e;
        return 1;
    }
    if (op == 3) {
        // the LDS-resident network's forward filter (resident.hip): [R][S][cin rounded up to 4][cout], the direct kernels' layout with ONE tile of all
        // output channels (wave-uniform scalar loads of 4 x cout weights per tap)
        dcv_prep_entry e{};
        e.kind = 0; e.kcout = d->cout; e.kcin = d->cin; e.R = d->r; e.S = d->s; e.w_cin = d->cin; e.dgrad = 0;
        e.tco = d->cout; e.tiles = 1; e.cin_pad = (d->cin + 3) & ~3;
        e.dst_off = 0; e.total = d->r * d->s * e.cin_pad * e.tco;
        out[0] = e;
        return 1;
    }
    if (op == 4) {
        // the LDS-resident network's data-gradient filter (resident.hip).  16 output channels of the layer on rows of 16 pixels: the matrix-core B fragments
        // (kind 2); otherwise the direct kernels' transposed + flipped layout with ONE tile [R][S][cout rounded up to 4][cin rounded up to 4]
        dcv_prep_entry e{};
        e.R = d->r; e.S = d->s; e.w_cin = d->cin; e.dgrad = 1; e.dst_off = 0;
        if (d->cout == 16 && d->w == 16 && d->r == 3 && d->s == 3 && d->cin <= 16 && d->cin != 4) {      // (4 <- 16: the VALU variant reads the direct layout)
            e.kind = 2; e.kcout = d->cin; e.kcin = 16;
            e.total = 16 * 4 * 4 * 9;
        } else {
            e.kind = 0; e.kcout = d->cin; e.kcin = d->cout;
            e.tco = (d->cin + 3) & ~3; e.tiles = 1; e.cin_pad = (d->cout + 3) & ~3;
            e.total = d->r * d->s * e.cin_pad * e.tco;
        }
        out[0] = e;
        return 1;
    }
    if (op == 1 && d->x_layout != DCV_LAYOUT_NHWC) return 0;
    return dcv_mfma_conv2d_supported(d, op) ? dcv_mfma_conv2d_prep_entries(d, op, out) : dcv_direct_conv2d_prep_entries(d, op, out);
}

inline size_t db_ws(const dcv_conv2d_desc* d) { return dcv_act_bwd_workspace_bytes((size_t)d->n * d->ho * d->wo, d->cout); }

}  // namespace

extern "C" {

size_t dcv_conv2d_fwd_workspace_bytes(const dcv_conv2d_desc* d) {
    if (!desc_ok(d)) return 0;
    return dcv_mfma_conv2d_supported(d, 0) ? dcv_mfma_conv2d_ws(d, 0) : dcv_direct_conv2d_fwd_ws(d);
}

size_t dcv_conv2d_dgrad_workspace_bytes(const dcv_conv2d_desc* d) {
    if (!desc_ok(d)) return 0;
    return dcv_mfma_conv2d_supported(d, 1) ? dcv_mfma_conv2d_ws(d, 1) : dcv_direct_conv2d_dgrad_ws(d);
}

size_t dcv_conv2d_wgrad_workspace_bytes(const dcv_conv2d_desc* d) {
    if (!desc_ok(d)) return 0;
    const size_t main_ws = dcv_mfma_conv2d_supported(d, 2) ? dcv_mfma_conv2d_ws(d, 2) : dcv_direct_conv2d_wgrad_ws(d);
    return dcv_align_up(main_ws, 256) + db_ws(d);
}

int dcv_conv2d_fwd(const dcv_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && x && w && y);
    if (d->groups != 1) return DCV_ERR_UNSUPPORTED;
    if (dcv_mfma_conv2d_supported(d, 0)) return dcv_mfma_conv2d_fwd(d, x, w, bias, y, ws, ws_bytes, dcv_stream(stream));
    return dcv_direct_conv2d_fwd(d, x, w, bias, y, ws, ws_bytes, dcv_stream(stream));
}

int dcv_conv2d_dgrad(const dcv_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && dy && w && dx);
    if (d->groups != 1) return DCV_ERR_UNSUPPORTED;
    if (d->x_layout != DCV_LAYOUT_NHWC) return DCV_ERR_UNSUPPORTED;   // dx is always produced NHWC
    if (dcv_mfma_conv2d_supported(d, 1)) return dcv_mfma_conv2d_dgrad(d, dy, w, dx, ws, ws_bytes, dcv_stream(stream));
    return dcv_direct_conv2d_dgrad(d, dy, w, dx, ws, ws_bytes, dcv_stream(stream));
}

size_t dcv_conv2d_filter_bytes(const dcv_conv2d_desc* d, int op) {
    if (!desc_ok(d) || d->groups != 1 || op < 0 || op > 4) return 0;
    dcv_prep_entry e[DCV_PREP_PIECES];
    if (prep_entries(d, op, e) == 0) return 0;
    if (op == 2 || op == 3 || op == 4) return dcv_align_up((size_t)e[0].total * sizeof(float), 256);
    return op == 0 ? dcv_conv2d_fwd_workspace_bytes(d) : dcv_conv2d_dgrad_workspace_bytes(d);
}

int dcv_conv2d_prepare_filters(int count, const dcv_conv2d_desc* const* descs, const int* ops, const float* const* w, void* const* filters, void* stream) {
    DCV_CHECK_ARG(count >= 0 && (count == 0 || (descs && ops && w && filters)));
    dcv_prep_list l{};
    int max_total = 0;
    auto flush = [&]() {
        if (l.count == 0) return;
        int gx = (max_total + 255) / 256; if (gx > 64) gx = 64; if (gx < 1) gx = 1;
        hipLaunchKernelGGL(prep_multi_kernel, dim3((unsigned)gx, (unsigned)l.count), dim3(256), 0, dcv_stream(stream), l);
        l.count = 0; max_total = 0;
    };
    for (int k = 0; k < count; ++k) {
        DCV_CHECK_ARG(desc_ok(descs[k]) && w[k] && ops[k] >= 0 && ops[k] <= 4);
        dcv_prep_entry pieces[DCV_PREP_PIECES];
        const int np = prep_entries(descs[k], ops[k], pieces);
        if (np == 0) continue;                                    // this (layer, op) reads the OIHW filter directly
        DCV_CHECK_ARG(filters[k] != nullptr);
        for (int q = 0; q < np; ++q) {
            dcv_prep_entry e = pieces[q];
            if (e.total == 0) continue;
            e.w = w[k]; e.dst = reinterpret_cast<float*>(filters[k]) + e.dst_off;
            l.e[l.count++] = e;
            if (e.total > max_total) max_total = e.total;
            if (l.count == DCV_PREP_MAX) flush();
        }
    }
    flush();
    DCV_RETURN_LAUNCH();
}

int dcv_adamw_prep_build(const dcv_tensor_list* lists, int n_lists, int n_filters, const dcv_conv2d_desc* const* descs, const int* ops,
                         const float* const* weights, void* const* filters, dcv_adamw_prep_table* out) {
    DCV_CHECK_ARG(lists && n_lists > 0 && out && n_filters >= 0 && (n_filters == 0 || (descs && ops && weights && filters)));
    dcv_adamw_prep_table t{};
    dcv_prep_entry* preps = reinterpret_cast<dcv_prep_entry*>(t.prep);
    bool used[256] = {false};
    if (n_filters > 256) return DCV_ERR_UNSUPPORTED;
    for (int li = 0; li < n_lists; ++li) {
        const dcv_tensor_list& tl = lists[li];
        DCV_CHECK_ARG(tl.count > 0 && tl.count <= DCV_MAX_TENSORS);
        for (int k = 0; k < tl.count; ++k) {
            DCV_CHECK_ARG(tl.param[k] && tl.grad[k] && tl.state1[k] && tl.state2[k] && tl.numel[k] > 0);
            if (tl.state3[k]) return DCV_ERR_UNSUPPORTED;                            // amsgrad: the multi-tensor kernel
            // the prepared layouts of this tensor
            dcv_prep_entry mine[DCV_OPT_MAX_PREPS]; int nm = 0;
            for (int f = 0; f < n_filters; ++f) {
                if (weights[f] != tl.param[k]) continue;
                DCV_CHECK_ARG(desc_ok(descs[f]) && ops[f] >= 0 && ops[f] <= 4 && filters[f]);
                used[f] = true;
                dcv_prep_entry pieces[DCV_PREP_PIECES];
                const int np = prep_entries(descs[f], ops[f], pieces);
                for (int q = 0; q < np; ++q) {
                    if (pieces[q].total == 0) continue;
                    if (nm == DCV_OPT_MAX_PREPS) return DCV_ERR_UNSUPPORTED;
                    dcv_prep_entry e = pieces[q];
                    e.w = tl.param[k]; e.dst = reinterpret_cast<float*>(filters[f]) + e.dst_off;
                    mine[nm++] = e;
                }
            }
            const long long numel = tl.numel[k];
            // a Linear weight with its one row-wise layout splits into groups of rows (rows are independent); everything else is one block
            const bool rows_split = numel > DCV_OPT_MAX_NUMEL && nm == 1 && mine[0].kind == 1 && !mine[0].dgrad && mine[0].kcout > 1 && numel % mine[0].kcout == 0
                                    && numel / mine[0].kcout <= DCV_OPT_MAX_NUMEL && (long long)mine[0].total == numel;
            if (numel > DCV_OPT_MAX_NUMEL && !rows_split) return DCV_ERR_UNSUPPORTED;
            const long long row = rows_split ? numel / mine[0].kcout : numel;
            const int rows_total = rows_split ? mine[0].kcout : 1, rows_per = rows_split ? (int)(DCV_OPT_MAX_NUMEL / row) : 1;
            for (int r0 = 0; r0 < rows_total; r0 += rows_per) {
                const int rows = r0 + rows_per <= rows_total ? rows_per : rows_total - r0;
                if (t.items == DCV_OPT_MAX_ITEMS || t.preps + nm > DCV_OPT_MAX_PREPS) return DCV_ERR_UNSUPPORTED;
                dcv_opt_item& it = t.item[t.items++];
                const long long off = (long long)r0 * row;
                it.param = tl.param[k] + off; it.grad = tl.grad[k] + off; it.state1 = tl.state1[k] + off; it.state2 = tl.state2[k] + off;
                it.numel = (int)((long long)rows * row);
                it.prep_first = (short)t.preps; it.prep_count = (short)nm;
                for (int q = 0; q < nm; ++q) {
                    dcv_prep_entry e = mine[q];
                    if (rows_split) { e.w += off; e.dst += off; e.kcout = rows; e.total = it.numel; }
                    preps[t.preps++] = e;
                }
            }
        }
    }
    for (int f = 0; f < n_filters; ++f) if (!used[f]) return DCV_ERR_UNSUPPORTED;    // a filter of a weight which is not updated here would go stale
    *out = t;
    return DCV_OK;
}

int dcv_adamw_step_prep_slabs(const dcv_adamw_prep_table* table, int n_segments, const dcv_slab_segment* segments, const float* hyper, void* stream) {
    DCV_CHECK_ARG(table && hyper && table->items > 0 && table->items <= DCV_OPT_MAX_ITEMS && table->preps >= 0 && table->preps <= DCV_OPT_MAX_PREPS && n_segments >= 0 && (n_segments == 0 || segments));
    if (n_segments > DCV_MAX_SLAB_SEGMENTS) return DCV_ERR_UNSUPPORTED;
    const dcv_prep_entry* preps = reinterpret_cast<const dcv_prep_entry*>(table->prep);
    for (int q = 0; q < table->preps; ++q) if (!dcv_prep_scatterable(preps[q])) return DCV_ERR_UNSUPPORTED;
    TailTable tt{};
    bool covered[DCV_OPT_MAX_ITEMS] = {};
    int blocks = 0;
    for (int i = 0; i < n_segments; ++i) {
        const dcv_slab_segment& g = segments[i];
        if (!g.src || !g.dst || g.chunks < 1 || g.stride < g.count) return DCV_ERR_INVALID_ARG;
        if (g.tr_a < 0 || g.tr_b < 0 || ((g.tr_a > 0) != (g.tr_b > 0)) || (g.tr_a > 0 && g.count % ((unsigned long long)g.tr_a * g.tr_b) != 0)) return DCV_ERR_INVALID_ARG;
        if (g.count == 0) continue;
        const int k = tt.slabs.count;
        const int ws = g.count >= SLAB_WIDE_COUNT ? 5 : 3;          // (as dcv_launch_slab_reduce: the summation tree depends on it)
        tt.slabs.block_start[k] = blocks; tt.slabs.wshift[k] = ws; tt.slabs.seg[k] = g;
        tt.item_of[k] = -1;
        for (int it = 0; it < table->items; ++it) {
            const dcv_opt_item& o = table->item[it];
            if (g.dst >= o.grad && g.dst < o.grad + o.numel) {
                // the segment writes (a part of) this item's gradient: all of it must lie inside the item
                if (g.dst + g.count > o.grad + o.numel) return DCV_ERR_UNSUPPORTED;
                tt.item_of[k] = (short)it; covered[it] = true;
                break;
            }
        }
        // a tensor split into several items (the rows of a Linear weight) whose gradient one segment writes as a whole: not representable here
        if (tt.item_of[k] < 0)
            for (int it = 0; it < table->items; ++it) { const dcv_opt_item& o = table->item[it]; if (o.grad >= g.dst && o.grad < g.dst + g.count) return DCV_ERR_UNSUPPORTED; }
        ++tt.slabs.count;
        blocks += (int)dcv_ceil_div((size_t)g.count, (size_t)1 << ws);
    }
    tt.slabs.block_start[tt.slabs.count] = blocks;
    tt.slab_blocks = blocks;
    // an item only partly covered by segments would be updated twice or not at all: every covered item must be covered entirely by ONE segment
    for (int k = 0; k < tt.slabs.count; ++k)
        if (tt.item_of[k] >= 0 && (tt.slabs.seg[k].dst != table->item[tt.item_of[k]].grad || (long long)tt.slabs.seg[k].count != table->item[tt.item_of[k]].numel)) return DCV_ERR_UNSUPPORTED;
    for (int it = 0; it < table->items; ++it) if (!covered[it]) tt.direct[tt.direct_count++] = (short)it;
    const int total = blocks + tt.direct_count;
    if (total <= 0) return DCV_OK;
    hipLaunchKernelGGL(adamw_prep_slabs_kernel, dim3((unsigned)total), dim3(1024), 0, dcv_stream(stream), *table, tt, hyper);
    DCV_RETURN_LAUNCH();
}

int dcv_adamw_step_prep(const dcv_adamw_prep_table* table, const float* hyper, void* stream) {
    DCV_CHECK_ARG(table && hyper && table->items > 0 && table->items <= DCV_OPT_MAX_ITEMS && table->preps >= 0 && table->preps <= DCV_OPT_MAX_PREPS);
    hipLaunchKernelGGL(adamw_prep_kernel, dim3((unsigned)table->items), dim3(1024), 0, dcv_stream(stream), *table, hyper);
    DCV_RETURN_LAUNCH();
}

int dcv_conv2d_fwd_stats_rows(const dcv_conv2d_desc* d) {
    if (!desc_ok(d) || d->groups != 1 || dcv_conv2d_filter_bytes(d, 0) == 0) return 0;
    return dcv_mfma_conv2d_supported(d, 0) ? dcv_mfma_conv2d_fwd_stats_rows(d) : dcv_direct_conv2d_fwd_stats_rows(d);
}

int dcv_conv2d_fwd_stats_layout(const dcv_conv2d_desc* d, int* rows, int* rows_per_sample) {
    DCV_CHECK_ARG(rows && rows_per_sample);
    *rows = 0; *rows_per_sample = 0;
    if (!desc_ok(d) || d->groups != 1 || dcv_conv2d_filter_bytes(d, 0) == 0) return DCV_OK;
    *rows = dcv_mfma_conv2d_supported(d, 0) ? dcv_mfma_conv2d_fwd_stats_rows(d, rows_per_sample) : dcv_direct_conv2d_fwd_stats_rows(d, rows_per_sample);
    if (*rows_per_sample > 0 && *rows != d->n * *rows_per_sample) *rows_per_sample = 0;
    return DCV_OK;
}

int dcv_conv2d_fwd_prepared(const dcv_conv2d_desc* d, const float* x, const float* w, const void* filter, const float* bias, float* y, double* stats, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && x && w && y);
    if (d->groups != 1) return DCV_ERR_UNSUPPORTED;
    const size_t fb = dcv_conv2d_filter_bytes(d, 0);
    if (fb == 0) return stats ? DCV_ERR_UNSUPPORTED : dcv_direct_conv2d_fwd(d, x, w, bias, y, nullptr, 0, dcv_stream(stream));
    DCV_CHECK_ARG(filter != nullptr);
    if (dcv_mfma_conv2d_supported(d, 0)) return dcv_mfma_conv2d_fwd(d, x, w, bias, y, const_cast<void*>(filter), fb, dcv_stream(stream), true, stats);
    return dcv_direct_conv2d_fwd(d, x, w, bias, y, const_cast<void*>(filter), fb, dcv_stream(stream), true, stats);
}

int dcv_conv2d_dgrad_prepared(const dcv_conv2d_desc* d, const float* dy, const float* w, const void* filter, float* dx, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && dy && w && dx);
    if (d->groups != 1 || d->x_layout != DCV_LAYOUT_NHWC) return DCV_ERR_UNSUPPORTED;
    const size_t fb = dcv_conv2d_filter_bytes(d, 1);
    if (fb == 0) return dcv_direct_conv2d_dgrad(d, dy, w, dx, nullptr, 0, dcv_stream(stream));
    DCV_CHECK_ARG(filter != nullptr);
    if (dcv_mfma_conv2d_supported(d, 1)) return dcv_mfma_conv2d_dgrad(d, dy, w, dx, const_cast<void*>(filter), fb, dcv_stream(stream), true);
    return dcv_direct_conv2d_dgrad(d, dy, w, dx, const_cast<void*>(filter), fb, dcv_stream(stream), true);
}

int dcv_conv2d_dgrad_prepared_add(const dcv_conv2d_desc* d, const float* dy, const float* w, const void* filter, const float* addend, float* dx, void* stream) {
    DCV_CHECK_ARG(desc_ok(d) && dy && w && dx && addend);
    if (d->groups != 1 || d->x_layout != DCV_LAYOUT_NHWC || dcv_mfma_conv2d_supported(d, 1)) return DCV_ERR_UNSUPPORTED;
    const size_t fb = dcv_conv2d_filter_bytes(d, 1);
    if (fb == 0) return DCV_ERR_UNSUPPORTED;
    DCV_CHECK_ARG(filter != nullptr);
    return dcv_direct_conv2d_dgrad(d, dy, w, dx, const_cast<void*>(filter), fb, dcv_stream(stream), true, addend);
}

int dcv_conv2d_wgrad_fuses_bias(const dcv_conv2d_desc* d) {
    if (!desc_ok(d) || d->groups != 1 || dcv_mfma_conv2d_supported(d, 2)) return 0;
    return dcv_direct_conv2d_wgrad_fuses_bias(d) ? 1 : 0;
}

static int conv2d_wgrad_impl(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, void* stream,
                             dcv_slab_segment* defer, int* ndefer) {
    DCV_CHECK_ARG(desc_ok(d) && x && dy && dw);
    if (d->groups != 1) return DCV_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < dcv_conv2d_wgrad_workspace_bytes(d)) return DCV_ERR_WORKSPACE;
    const bool mfma = dcv_mfma_conv2d_supported(d, 2);
    const size_t main_ws = dcv_align_up(mfma ? dcv_mfma_conv2d_ws(d, 2) : dcv_direct_conv2d_wgrad_ws(d), 256);
    bool db_done = false;
    int rc = mfma ? dcv_mfma_conv2d_wgrad(d, x, dy, dw, ws, main_ws, dcv_stream(stream))
                  : dcv_direct_conv2d_wgrad(d, x, dy, dw, db, &db_done, ws, main_ws, dcv_stream(stream), defer, ndefer);
    if (rc != DCV_OK) return rc;
    if (db && !db_done) rc = dcv_channel_sum(dy, db, (size_t)d->n * d->ho * d->wo, d->cout, reinterpret_cast<char*>(ws) + main_ws, db_ws(d), stream);
    return rc;
}

int dcv_conv2d_wgrad(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
    return conv2d_wgrad_impl(d, x, dy, dw, db, ws, ws_bytes, stream, nullptr, nullptr);
}

// at most 64 jobs (bounded stack tables); APPENDS its segments behind the *n_segments already there
static int wgrad_batch_part(int count, const dcv_wgrad_job* jobs, dcv_slab_segment* segments, int* n_segments, void* stream) {
    bool eligible[64], taken[64];
    for (int i = 0; i < count; ++i) {
        const dcv_wgrad_job& j = jobs[i];
        DCV_CHECK_ARG(desc_ok(&j.desc) && j.x && j.dy && j.dw);
        if (j.desc.groups != 1) return DCV_ERR_UNSUPPORTED;
        if (!j.ws || j.ws_bytes < dcv_conv2d_wgrad_workspace_bytes(&j.desc)) return DCV_ERR_WORKSPACE;
        eligible[i] = !dcv_mfma_conv2d_supported(&j.desc, 2);
    }
    int rc = dcv_direct_conv2d_wgrad_multi(count, jobs, eligible, taken, segments, n_segments, dcv_stream(stream));
    for (int i = 0; i < count && rc == DCV_OK; ++i) {
        if (taken[i]) continue;
        const dcv_wgrad_job& j = jobs[i];
        int n = 0;
        rc = conv2d_wgrad_impl(&j.desc, j.x, j.dy, j.dw, j.db, j.ws, j.ws_bytes, stream, segments + *n_segments, &n);
        *n_segments += n;
    }
    return rc;
}

// `segments` (room for 2 * count entries) is an OUT array: every call starts it at 0.  (It used to append behind whatever *n_segments held: a caller
// re-issuing the call with the same objects — a launch list replaying a captured step — wrote past the array, further each time.)
int dcv_conv2d_wgrad_batch(int count, const dcv_wgrad_job* jobs, dcv_slab_segment* segments, int* n_segments, void* stream) {
    DCV_CHECK_ARG(count >= 0 && (count == 0 || (jobs && segments)) && n_segments);
    *n_segments = 0;
    for (int i0 = 0; i0 < count; i0 += 64) {
        const int rc = wgrad_batch_part(count - i0 < 64 ? count - i0 : 64, jobs + i0, segments, n_segments, stream);
        if (rc != DCV_OK) return rc;
    }
    return DCV_OK;
}

int dcv_conv2d_wgrad_deferred(const dcv_conv2d_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes,
                              dcv_slab_segment* segments, int* n_segments, void* stream) {
    DCV_CHECK_ARG(segments && n_segments);
    *n_segments = 0;
    return conv2d_wgrad_impl(d, x, dy, dw, db, ws, ws_bytes, stream, segments, n_segments);
}

}  // extern "C"
