// p256_group_kernels.hip — kernels for grouping a generic batch by public key inside the step
// (p256_group.h) and for running stage B over the two compacted index lists.
//
//   k_group_insert / k_group_assign / k_group_split : one lane per tuple; global atomics only
//   k_keytab_bases   : one lane per grouped key  (256 doublings: the latency floor of a fresh key)
//   k_keytab_window  : four lanes per (key, window): 32 mixed additions + Montgomery-trick normalisation each,
//                      Jacobian intermediates parked in a private 16 KiB strip of HBM
//   k_verify_keyed_list / k_verify_generic_list : stage B over grp_idx / ung_idx, one accept BYTE per tuple
//   k_pack_bitmap    : accept bytes -> LSB-first bitmap
// Launch sizes that depend on device-side counters use the upper bound; surplus lanes exit at once.
#include <hip/hip_runtime.h>

#include "p256_group.h"
#include "p256_kernels.h"

namespace sbv {

__global__ __launch_bounds__(256) void k_group_insert(const uint8_t* __restrict__ tuples, size_t n, GroupState g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) group_insert_lane(tuples, i, g);
}
__global__ __launch_bounds__(256) void k_group_assign(size_t n, GroupState g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) group_assign_lane(i, g);
}
// Same result as group_split_lane, but with ONE atomic per wavefront and list instead of one per lane:
// a million atomicAdds on two words serialise at ~11 ns each (11 ms per batch, measured); a ballot +
// prefix popcount needs 2 x n/64 of them.
__global__ __launch_bounds__(256) void k_group_split(size_t n, GroupState g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n;
    u32 s = SBV_GROUP_NONE;
    if (active) s = g.slot_of[g.rep[i]];
    const bool ung = active && s == SBV_GROUP_NONE;
    const bool grp = active && s != SBV_GROUP_NONE;
    const unsigned long long mu = __ballot(ung), mg = __ballot(grp);
    const int lane = threadIdx.x & 63;
    u32 base_u = 0, base_g = 0;
    if (lane == 0) {
        if (mu) base_u = atomicAdd(&g.counters[2], (u32)__popcll(mu));
        if (mg) base_g = atomicAdd(&g.counters[1], (u32)__popcll(mg));
    }
    base_u = __shfl(base_u, 0, 64);
    base_g = __shfl(base_g, 0, 64);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (ung) g.ung_idx[base_u + (u32)__popcll(mu & below)] = (u32)i;
    if (grp) {
        g.slots[i] = s;
        g.grp_idx[base_g + (u32)__popcll(mg & below)] = (u32)i;
    }
}

__device__ __forceinline__ u32 group_count(const GroupState& g) {
    const u32 c = g.counters[0];
    return c < g.max_groups ? c : g.max_groups;
}

__global__ __launch_bounds__(64) void k_keytab_bases(const uint8_t* __restrict__ tuples, GroupState g, apt* __restrict__ bases,
                                                     uint8_t* __restrict__ valid) {
    const u32 k = blockIdx.x * 64 + threadIdx.x;
    if (k < group_count(g)) keytab_bases_lane(tuples, k, g, bases, valid);
}

__global__ __launch_bounds__(64) void k_keytab_window(GroupState g, const apt* __restrict__ bases, u32* __restrict__ tmp,
                                                      apt* __restrict__ ktab) {
    const u32 lane = blockIdx.x * 64 + threadIdx.x;          // = (key * 33 + window) * 4 + part
    const u32 kw = lane / SBV_KEYTAB_PARTS, part = lane % SBV_KEYTAB_PARTS;
    if (kw / SBV_GTAB_WINDOWS >= group_count(g)) return;
    keytab_window_lane(bases[kw], (int)part, tmp + (size_t)lane * SBV_KEYTAB_TMP_DWORDS, ktab + (size_t)kw * SBV_GTAB_PER_WINDOW);
}

__global__ __launch_bounds__(SBV_VERIFY_BLOCK) void k_verify_keyed_list(Scratch s, GroupState g, const apt* __restrict__ ktab,
                                                                       const uint8_t* __restrict__ kvalid,
                                                                       const apt* __restrict__ g16, uint8_t* __restrict__ acc) {
    const u32 L = blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    if (L >= g.counters[1]) return;
    const u32 t = g.grp_idx[L];
    acc[t] = verify_lane_keyed<false>(s, t, g.slots[t], group_count(g), ktab, kvalid, g16) ? 1 : 0;
}

__global__ __launch_bounds__(SBV_VERIFY_BLOCK) void k_verify_generic_list(Scratch s, GroupState g, u32* __restrict__ qtab,
                                                                         const apt* __restrict__ g16, uint8_t* __restrict__ acc) {
    const u32 L = blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    if (L >= g.counters[2]) return;
    const u32 t = g.ung_idx[L];
    acc[t] = verify_lane<false>(s, t, qtab + (size_t)L * (SBV_QTAB_ENTRIES * 40), g16) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_pack_bitmap(const uint8_t* __restrict__ acc, size_t n, uint8_t* __restrict__ bitmap) {
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;           // bitmap byte
    if (b >= ((n + 7) >> 3)) return;
    u32 v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = b * 8 + k;
        if (i < n && acc[i]) v |= 1u << k;
    }
    bitmap[b] = (uint8_t)v;
}

// Enqueue stage B with in-step grouping.  Stage A (k_p256_prep) is already enqueued on `stream`.
// Two streams: the grouping + table-building chain (low occupancy, latency-bound: 33 lanes per key)
// runs on `side` concurrently with stage A and with the generic kernel over the ungrouped list; only
// the registered-key kernel has to wait for the tables.
//
//   stream: [prep] ........ wait(split) generic_list ....... wait(tables) keyed_list  pack
//   side  : insert assign split | bases ------- windows ------|
hipError_t launch_p256_verify_grouped(const uint8_t* d_tuples, const Scratch& s, size_t n, const GroupBuffers& b,
                                      u32* d_qtab, const apt* d_g16, uint8_t* d_bitmap, hipStream_t stream, hipStream_t side,
                                      hipEvent_t ev_fork, hipEvent_t ev_split, hipEvent_t ev_tables, hipEvent_t prof_k0,
                                      hipEvent_t prof_k1) {
    if (n == 0) return hipSuccess;
    GroupState g;
    g.ht = b.ht; g.ht_mask = b.ht_mask; g.rep = b.rep; g.cnt = b.cnt; g.slot_of = b.slot_of; g.group_rep = b.group_rep;
    g.counters = b.counters; g.grp_idx = b.grp_idx; g.ung_idx = b.ung_idx; g.slots = b.slots;
    g.max_groups = b.max_groups; g.min_count = b.min_count;
    hipError_t e;
    // `side` may not touch the group buffers before everything already enqueued on `stream` (the previous
    // batch's readers of those buffers) has run; ev_fork was recorded by the caller BEFORE stage A.
    if ((e = hipStreamWaitEvent(side, ev_fork, 0)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(b.ht, 0, ((size_t)b.ht_mask + 1) * sizeof(u32), side)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(b.cnt, 0, n * sizeof(u32), side)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(b.counters, 0, 4 * sizeof(u32), side)) != hipSuccess) return e;
    const unsigned gn = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_group_insert, dim3(gn), dim3(256), 0, side, d_tuples, n, g);
    hipLaunchKernelGGL(k_group_assign, dim3(gn), dim3(256), 0, side, n, g);
    hipLaunchKernelGGL(k_group_split, dim3(gn), dim3(256), 0, side, n, g);
    if ((e = hipEventRecord(ev_split, side)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_keytab_bases, dim3((b.max_groups + 63) / 64), dim3(64), 0, side, d_tuples, g, b.bases, b.kvalid);
    hipLaunchKernelGGL(k_keytab_window, dim3((b.max_groups * SBV_GTAB_WINDOWS * SBV_KEYTAB_PARTS + 63) / 64), dim3(64), 0, side, g, b.bases, b.tmp, b.ktab);
    if ((e = hipEventRecord(ev_tables, side)) != hipSuccess) return e;
    const unsigned gv = (unsigned)((n + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK);
    if ((e = hipStreamWaitEvent(stream, ev_split, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_verify_generic_list, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, d_qtab, d_g16, b.acc);
    if ((e = hipStreamWaitEvent(stream, ev_tables, 0)) != hipSuccess) return e;
    if (prof_k0 && (e = hipEventRecord(prof_k0, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_verify_keyed_list, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, b.ktab, b.kvalid, d_g16, b.acc);
    if (prof_k1 && (e = hipEventRecord(prof_k1, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_pack_bitmap, dim3((unsigned)(((n + 7) / 8 + 255) / 256)), dim3(256), 0, stream, b.acc, n, d_bitmap);
    return hipGetLastError();
}

}  // namespace sbv
