// allreduce.hip — one-shot mean all-reduce of a SMALL gradient bucket across the GPUs of one node (the dcv_allreduce_* family of include/deepcv_hip.h).
// Replaces the hook-driven bucket all-reduce of DistributedDataParallel (reference src/deepcv/meta/ignite_training.py:373-390, collective C1 of
// SURVEY.md §2.3) for the stock classifier, whose whole gradient is ONE 68 KB bucket that only exists when the backward pass has ended: nothing to overlap
// with, so what counts is latency, and a library collective is a kernel launch of its own plus a ring / tree protocol sized for megabytes.
//
// Protocol (push, one hop, no collective library): every rank owns an INBOX in device memory, mapped into every peer process by hipIpc (dmabuf).
//   inbox = header [64 words] | slots [2 parities][world sources][max_floats] of 8-byte {epoch, value} granules
// A launch of rank r  (1) stores each of its bucket values as one granule into EVERY peer's inbox at [parity][r][i] — plain 8-byte system-scope stores over
// xGMI: the tag travels with the data, no fence, no flag;  (2) polls its OWN inbox (local memory) for the granules of every other rank until their tag
// shows the epoch (bounded spin) and adds the world values in RANK ORDER, its own in its place: every rank computes the same bits;  (3) writes sum / world
// over its bucket.  The epoch is a counter in the inbox header which the launch advances itself when its last block is done (a replayed launch list or
// hipGraph needs no per-step argument); two parities because a fast rank may push step k + 1 while a slow one still reads step k (it cannot get further
// ahead: step k + 1 completes nowhere before every rank has pushed it).
// Memory model (round 4): a launch polls words which OTHER devices store while it runs.  HIP promises cross-agent visibility inside a running kernel only for
// fine-grained (or uncached) allocations — coarse-grained memory (plain hipMalloc) is only coherent at kernel boundaries, and that a poll of it sees a peer's
// store at all is hardware behaviour, not a guarantee.  The inbox is therefore allocated FINE-GRAINED (hipExtMallocWithFlags(hipDeviceMallocFinegrained);
// DCV_ALLREDUCE_MEM=uncached / coarse select the other kinds for experiments), and all accesses to it are system-scope atomics.
// Waiting: a rank which is late (loader stall, checkpoint, first-replay verification) must not poison anybody — DistributedDataParallel would simply wait.  The
// wait is bounded by TIME (s_memrealtime, 100 MHz), by default 600 s (`DCV_ALLREDUCE_TIMEOUT_S`, read by dcv_allreduce_oneshot at every call); only a peer
// which is really gone ends it.  Then the WHOLE launch fails together: the thread which gave up raises the error word in the inbox header (and, when
// dcv_allreduce_set_error_word registered one, in a pinned host word the training loop polls without synchronising), every thread of every block which sees
// it stops waiting, and every element written after that is NaN.
// Roofline: latency — one xGMI store propagation (~2 us) + the kernel boundary; 136 KB per peer and step.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef unsigned long long ar_u64;
typedef __attribute__((address_space(1))) ar_u64 ar_gu64;
typedef __attribute__((address_space(1))) unsigned ar_gu32;

constexpr int AR_HEADER_WORDS = 64;          // [0] epoch of the last completed launch, [1] blocks done in the running launch, [2] error word (1 + rank which never arrived),
                                             // [4..5] address of a host-visible copy of the error word (0: none)
constexpr int AR_T = 256;
constexpr int AR_POLLS_PER_CLOCK_CHECK = 64; // the deadline and the launch's error word are looked at every that many polls

struct ar_peers { ar_u64* inbox[DCV_ALLREDUCE_MAX_RANKS]; };

__global__ __launch_bounds__(AR_T) void allreduce_oneshot_kernel(float* bucket, size_t count, int rank, int world, unsigned* header, ar_peers peers, size_t max_floats, float inv_world,
                                                                 unsigned long long timeout_ticks) {
    const unsigned epoch = __hip_atomic_load((ar_gu32*)header, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;      // (advanced below, when EVERY block of this launch is done)
    const size_t parity = (size_t)(epoch & 1u) * (size_t)world * max_floats;
    ar_u64* mine = peers.inbox[rank];
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    bool failed = __hip_atomic_load((ar_gu32*)(header + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;          // an earlier launch failed: this bucket is not to be trusted either
    for (size_t i0 = (size_t)blockIdx.x * AR_T; i0 < count; i0 += (size_t)gridDim.x * AR_T) {
        const size_t i = i0 + threadIdx.x;
        const bool live = i < count;
        const float v = live ? bucket[i] : 0.f;
        if (live) {
            const ar_u64 g = ((ar_u64)epoch << 32) | (ar_u64)__float_as_uint(v);
            for (int p = 0; p < world; ++p)
                if (p != rank) __hip_atomic_store((ar_gu64*)(peers.inbox[p] + parity + (size_t)rank * max_floats + i), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        float acc = 0.f;
        for (int q = 0; q < world; ++q) {                                            // rank order: the same sum on every rank
            float vq = v;
            if (q != rank && live && !failed) {
                const ar_gu64* src = (const ar_gu64*)(mine + parity + (size_t)q * max_floats + i);
                ar_u64 g = 0;
                bool got = false;
                for (unsigned polls = 1; ; ++polls) {
                    g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    got = (unsigned)(g >> 32) == epoch;
                    if (got) break;
                    if (polls % AR_POLLS_PER_CLOCK_CHECK == 0u) {
                        // somebody else of this launch gave up already, or the peer has been silent for the whole bound
                        if (__hip_atomic_load((ar_gu32*)(header + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
                        if (__builtin_amdgcn_s_memrealtime() - t_start > timeout_ticks) break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!got) {
                    failed = true;
                    unsigned expected = 0u;
                    if (__hip_atomic_compare_exchange_strong((ar_gu32*)(header + 2), &expected, (unsigned)q + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                        const unsigned long long host_word = *reinterpret_cast<const unsigned long long*>(header + 4);
                        if (host_word) __hip_atomic_store((ar_gu32*)host_word, (unsigned)q + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                vq = __uint_as_float((unsigned)g);
            }
            acc += vq;
        }
        if (live) bucket[i] = failed ? __builtin_nanf("") : acc * inv_world;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add((ar_gu32*)(header + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {                                                 // the last block: nobody of this launch reads the epoch any more
            __hip_atomic_store((ar_gu32*)(header + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((ar_gu32*)header, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// kind of memory the inbox lives in (see the memory-model note at the top)
inline int ar_mem_kind() {
    const char* e = getenv("DCV_ALLREDUCE_MEM");
    if (e && !strcmp(e, "coarse")) return 0;
    if (e && !strcmp(e, "uncached")) return 2;
    return 1;
}

inline unsigned long long ar_timeout_ticks() {
    const char* e = getenv("DCV_ALLREDUCE_TIMEOUT_S");
    double s = e ? atof(e) : 600.0;
    if (!(s > 0.0)) s = 600.0;
    return (unsigned long long)(s * 1e8);                                            // s_memrealtime counts at 100 MHz
}

}  // namespace

extern "C" {

size_t dcv_allreduce_inbox_bytes(int world, size_t max_floats) {
    if (world < 1 || world > DCV_ALLREDUCE_MAX_RANKS) return 0;
    return (size_t)AR_HEADER_WORDS * 4 + 2 * (size_t)world * max_floats * sizeof(ar_u64);
}

// The inbox must be an allocation of its own (hipIpc exports whole allocations): the one place where this library allocates device memory.
int dcv_allreduce_alloc(size_t bytes, void** inbox) {
    DCV_CHECK_ARG(inbox && bytes >= (size_t)AR_HEADER_WORDS * 4);
    void* p = nullptr;
    const int kind = ar_mem_kind();
    const hipError_t rc = kind == 0 ? hipMalloc(&p, bytes) : hipExtMallocWithFlags(&p, bytes, kind == 2 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    if (rc != hipSuccess) { (void)hipGetLastError(); return DCV_ERR_WORKSPACE; }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return DCV_ERR_LAUNCH; }
    *inbox = p;
    return DCV_OK;
}

int dcv_allreduce_free(void* inbox) {
    DCV_CHECK_ARG(inbox);
    return hipFree(inbox) == hipSuccess ? DCV_OK : DCV_ERR_LAUNCH;
}

int dcv_allreduce_export(void* inbox, unsigned char* handle) {
    DCV_CHECK_ARG(inbox && handle);
    static_assert(sizeof(hipIpcMemHandle_t) == DCV_ALLREDUCE_HANDLE_BYTES, "the C-ABI's handle size is HIP's");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, inbox) != hipSuccess) { (void)hipGetLastError(); return DCV_ERR_UNSUPPORTED; }
    memcpy(handle, &h, sizeof(h));
    return DCV_OK;
}

int dcv_allreduce_import(const unsigned char* handle, void** peer_inbox) {
    DCV_CHECK_ARG(handle && peer_inbox);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return DCV_ERR_UNSUPPORTED; }
    *peer_inbox = p;
    return DCV_OK;
}

int dcv_allreduce_unmap(void* peer_inbox) {
    DCV_CHECK_ARG(peer_inbox);
    return hipIpcCloseMemHandle(peer_inbox) == hipSuccess ? DCV_OK : DCV_ERR_LAUNCH;
}

int dcv_allreduce_oneshot(float* bucket, size_t count, int rank, int world, void* const* inboxes, size_t max_floats, void* stream) {
    DCV_CHECK_ARG(bucket && inboxes && world >= 1 && world <= DCV_ALLREDUCE_MAX_RANKS && rank >= 0 && rank < world && count <= max_floats);
    if (count == 0) return DCV_OK;
    ar_peers peers{};
    for (int p = 0; p < world; ++p) {
        DCV_CHECK_ARG(inboxes[p]);
        peers.inbox[p] = reinterpret_cast<ar_u64*>(reinterpret_cast<char*>(inboxes[p]) + (size_t)AR_HEADER_WORDS * 4);
    }
    // few enough blocks to be co-resident with anything else on the chip (the polls of one rank wait for the pushes of the others, never for its own blocks)
    unsigned grid = (unsigned)dcv_ceil_div(count, AR_T);
    if (grid > 128u) grid = 128u;
    hipLaunchKernelGGL(allreduce_oneshot_kernel, dim3(grid), dim3(AR_T), 0, dcv_stream(stream), bucket, count, rank, world, reinterpret_cast<unsigned*>(inboxes[rank]), peers, max_floats,
                       1.f / (float)world, ar_timeout_ticks());
    DCV_RETURN_LAUNCH();
}

// `host_word`: a 4-byte word of pinned (device-visible) host memory, 0 at registration, which a failing launch sets to 1 + the rank it never heard from: the
// training loop can look at it every step without synchronising anything.  nullptr unregisters.  Synchronises (like dcv_allreduce_status): setup time only.
int dcv_allreduce_set_error_word(void* inbox, unsigned* host_word) {
    DCV_CHECK_ARG(inbox);
    const unsigned long long v = (unsigned long long)host_word;
    if (hipMemcpy(reinterpret_cast<char*>(inbox) + 16, &v, sizeof(v), hipMemcpyHostToDevice) != hipSuccess) return DCV_ERR_LAUNCH;
    return DCV_OK;
}

int dcv_allreduce_status(const void* inbox, void* stream, int* err) {
    DCV_CHECK_ARG(inbox && err);
    unsigned h[3] = {0, 0, 0};
    if (hipMemcpyAsync(h, inbox, sizeof(h), hipMemcpyDeviceToHost, dcv_stream(stream)) != hipSuccess) return DCV_ERR_LAUNCH;
    if (hipStreamSynchronize(dcv_stream(stream)) != hipSuccess) return DCV_ERR_LAUNCH;
    *err = (int)h[2];
    return DCV_OK;
}

}  // extern "C"
