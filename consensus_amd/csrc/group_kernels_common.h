// group_kernels_common.h — the curve-independent kernels of the in-step key grouping, shared by the P-256 and
// the Ed25519 pipelines (each translation unit gets its own static copy; no relocatable device code needed).
#pragma once
#include <hip/hip_runtime.h>

#include "p256_group.h"

namespace sbv {

__device__ __forceinline__ u32 group_count(const GroupState& g) {
    const u32 c = g.counters[0];
    return c < g.max_groups ? c : g.max_groups;
}

// The insert kernel of every scheme: one lane per tuple finds its key's representative; the counts go to g.cnt AGGREGATED per
// workgroup (round 6).  One atomic per tuple was fine for the headline's 1 024 signers (1 000 increments per counter), but a
// consenter batch — configs[3]: 550 000 signatures by 16 keys whose representatives are tuples 0..15, i.e. 16 counters in ONE cache
// line — serialised 34 000 device-scope atomics per counter in one memory channel: k_group_insert 1.40 ms and, stalled behind it,
// stage A 1.43 ms of a 3.2 ms call (profiles/r06/kernel_stats_replay_r06l.csv).  Now: the lanes of a wavefront that share a
// representative elect a leader (ballot / readlane loop, one round per distinct representative), the leaders meet in a 512-entry
// LDS table keyed by the representative, and the workgroup flushes one atomic per distinct representative.
template <int STRIDE, int OFF, int WORDS>
static __device__ __forceinline__ void group_insert_block(const uint8_t* __restrict__ tuples, size_t n, const GroupState& g) {
    __shared__ u32 skey[512], scnt[512];
    const unsigned tid = threadIdx.x;
    for (unsigned t = tid; t < 512; t += blockDim.x) { skey[t] = 0xFFFFFFFFu; scnt[t] = 0; }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + tid;
    u32 mine = 0xFFFFFFFFu;
    bool want = false;
    if (i < n) {
        mine = group_find_rep_t<STRIDE, OFF, WORDS>(tuples, i, g);
        want = group_sampled((u32)i, g.sample_mask);
    }
    unsigned long long todo = __ballot(want);
    const int lane = (int)(tid & 63u);
    if (todo) {
        // A wavefront whose first representative is nobody else's (the headline batch: 64 consecutive tuples of 64 signers) gains nothing
        // from the rounds below — 64 of them, one lane at work in each, on the path to the G phase (+0.25 ms per 2^20 step when every
        // wavefront took them, profiles/r06/bench_driver_flags_r06m.json): its lanes count for themselves, as rounds 1-5 did.
        const int first = __ffsll((long long)todo) - 1;
        const u32 r0 = (u32)__shfl((int)mine, first, 64);
        if (__popcll(__ballot(want && mine == r0)) < 2) {
            if (want) atomicAdd(&g.cnt[mine], 1u);
            todo = 0;
        }
    }
    while (todo) {                                          // wave-uniform loop: one round per distinct representative
        const int leader = __ffsll((long long)todo) - 1;
        const u32 r = (u32)__shfl((int)mine, leader, 64);
        const unsigned long long same = __ballot(want && mine == r);
        if (lane == leader) {
            const u32 c = (u32)__popcll(same);
            u32 h = (r * 0x9E3779B1u) >> 23;                // 9 bits
            for (int probes = 0; probes < 512; ++probes) {
                const u32 prev = atomicCAS(&skey[h], 0xFFFFFFFFu, r);
                if (prev == 0xFFFFFFFFu || prev == r) { atomicAdd(&scnt[h], c); break; }
                h = (h + 1u) & 511u;
            }
        }
        todo &= ~same;
    }
    __syncthreads();
    for (unsigned t = tid; t < 512; t += blockDim.x)
        if (skey[t] != 0xFFFFFFFFu) atomicAdd(&g.cnt[skey[t]], scnt[t]);
}

// kc: the scheme's persistent key-table cache (kc.enabled = 0 when it is off): cached keys are grouped whatever their count
// in this batch.  STRIDE / OFF / WORDS = the tuple format's key location (p256_group.h: group_insert_lane_t)
template <int STRIDE, int OFF, int WORDS>
static __global__ __launch_bounds__(256) void k_group_assign_t(const uint8_t* __restrict__ tuples, size_t n, GroupState g, KeyCache kc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) group_assign_lane_t<STRIDE, OFF, WORDS>(tuples, i, g, kc);
}
static __global__ __launch_bounds__(256) void k_group_assign(const uint8_t* __restrict__ tuples, size_t n, GroupState g, KeyCache kc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) group_assign_lane(tuples, i, g, kc);
}

// Every group of the batch finds its table slot (p256_group.h: persistent key-table cache), in two small launches of
// 64-lane workgroups.  The lookup is read-only: everything in the table was inserted by earlier batches, i.e. by earlier
// kernels; the insert places the misses (atomics only; the keys of one batch are distinct, so nobody needs to read what a
// neighbour just wrote).  (One 1024-lane workgroup did both with a barrier in between; it had to wait ~130 us for a whole CU
// to drain while stage A and the split kernel filled the device, at the head of the table-building chain.)
template <int STRIDE, int OFF, int WORDS>
static __global__ __launch_bounds__(64) void k_key_cache_lookup_t(const uint8_t* __restrict__ tuples, GroupState g, KeyCache kc,
                                                                  u32* __restrict__ tslot, uint8_t* __restrict__ cold) {
    const u32 k = blockIdx.x * 64 + threadIdx.x;
    if (k == 0) { kc.count[1] = 0; kc.count[2] = 0; }       // hits / misses of this batch: counted by the insert kernel
    if (k >= group_count(g)) return;
    key_cache_phase_lookup<STRIDE, OFF, WORDS>(tuples, g, kc, k, tslot, cold);
}
template <int STRIDE, int OFF, int WORDS>
static __global__ __launch_bounds__(64) void k_key_cache_insert_t(const uint8_t* __restrict__ tuples, GroupState g, KeyCache kc,
                                                                  u32* __restrict__ tslot) {
    const u32 k = blockIdx.x * 64 + threadIdx.x;
    if (k >= group_count(g)) return;
    key_cache_phase_insert<STRIDE, OFF, WORDS>(tuples, g, kc, k, tslot);
}

// ---- hot keys: the scheme-independent kernels (p256_group.h "hot keys" / "life cycle"; round 6: shared with the Ed25519 step) ----------
// wtab: the pool of wide combs (a scheme's own entry type: opaque here); kwide == nullptr = feature off / no pool
struct HotKeys { const void* wtab; u32* kwide; u32* khits; u32* hot; u32* plist; u32 cache_cap, wide_cap, promote_min; u32* wowner; u32* elist; };
static __device__ __forceinline__ u32 promote_live(const u32* hot) { const u32 c = hot[1]; return c < SBV_PROMOTE_MAX ? c : SBV_PROMOTE_MAX; }
// Life cycle of the hot keys (p256_group.h, round 6): the clock sweep and the evictions.
static __global__ __launch_bounds__(256) void k_hot_decay(HotKeys hk) {
    const u32 slot = blockIdx.x * 256 + threadIdx.x;
    if (slot < hk.cache_cap) hot_decay_lane(slot, hk.khits);
}
static __global__ __launch_bounds__(256) void k_promote_select(GroupState g, const u32* __restrict__ tslot, const uint8_t* __restrict__ kvalid, HotKeys hk) {
    const u32 groups = group_count(g);
    for (u32 k = blockIdx.x * 256 + threadIdx.x; k < groups; k += gridDim.x * 256)
        group_promote_select_lane(k, tslot, kvalid, hk.cache_cap, hk.kwide, hk.khits, hk.promote_min, hk.wide_cap, hk.hot, hk.plist, hk.elist);
}
// ONE workgroup: for each slot that found the pool full, its 1024 lanes scan the owners for the coldest comb not handed out in this
// batch, lane 0 merges and commits (hot_evict_commit: the hysteresis, kwide of the victim, wowner, the plist entry the builder reads).
#define SBV_HOT_POOL_MAX 4096u
static __global__ __launch_bounds__(1024) void k_promote_evict(HotKeys hk) {
    __shared__ u32 taken[SBV_HOT_POOL_MAX / 32];
    __shared__ u32 sh_h[16], sh_w[16];
    __shared__ u32 entries;
    const u32 tid = threadIdx.x;
    const u32 ncand = hk.hot[3] < SBV_PROMOTE_MAX ? hk.hot[3] : SBV_PROMOTE_MAX;
    if (ncand == 0 || hk.wide_cap > SBV_HOT_POOL_MAX) return;             // uniform
    for (u32 i = tid; i < SBV_HOT_POOL_MAX / 32; i += 1024) taken[i] = 0;
    if (tid == 0) entries = hk.hot[1] < SBV_PROMOTE_MAX ? hk.hot[1] : SBV_PROMOTE_MAX;
    __syncthreads();
    {   // A full pool of keys about as hot as the candidates is the steady state of a signer set larger than the pool: 64 candidates per
        // batch, none of which may take a comb, cost 64 scans = 0.1 ms on the stream the next batch's sort waits on
        // (profiles/r06/timeline_p256_half_hot_r06aw.txt).  One scan settles it: if the coldest owner is too hot for the HOTTEST candidate
        // (hot_evict_ok), it is too hot for every candidate.
        u32 mc = 0;
        for (u32 c = tid; c < ncand; c += 1024) { const u32 h = hk.khits[hk.elist[c]]; mc = h > mc ? h : mc; }
        u32 bh, bw;
        hot_evict_scan(hk.khits, hk.wowner, taken, hk.wide_cap, hk.cache_cap, tid, 1024u, bh, bw);
        for (int off = 32; off >= 1; off >>= 1) {
            const u32 om = (u32)__shfl_xor((int)mc, off, 64), oh = (u32)__shfl_xor((int)bh, off, 64);
            mc = om > mc ? om : mc;
            bh = oh < bh ? oh : bh;
        }
        if ((tid & 63) == 0) { sh_h[tid >> 6] = bh; sh_w[tid >> 6] = mc; }
        __syncthreads();
        for (int i = 0; i < 16; ++i) { bh = sh_h[i] < bh ? sh_h[i] : bh; mc = sh_w[i] > mc ? sh_w[i] : mc; }
        __syncthreads();
        if (bh == 0xFFFFFFFFu || !hot_evict_ok(mc, bh)) return;          // uniform: every lane holds the same two numbers
    }
    for (u32 c = 0; c < ncand; ++c) {
        u32 bh, bw;
        hot_evict_scan(hk.khits, hk.wowner, taken, hk.wide_cap, hk.cache_cap, tid, 1024u, bh, bw);
        for (int off = 32; off >= 1; off >>= 1) {
            const u32 oh = (u32)__shfl_xor((int)bh, off, 64), ow = (u32)__shfl_xor((int)bw, off, 64);
            if (hot_evict_better(oh, ow, bh, bw)) { bh = oh; bw = ow; }
        }
        if ((tid & 63) == 0) { sh_h[tid >> 6] = bh; sh_w[tid >> 6] = bw; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < 16; ++i) if (hot_evict_better(sh_h[i], sh_w[i], bh, bw)) { bh = sh_h[i]; bw = sh_w[i]; }
            entries = hot_evict_commit(hk.elist[c], bh, bw, hk.khits, hk.kwide, hk.wowner, taken, entries, hk.plist);
        }
        __syncthreads();
    }
    if (tid == 0) hk.hot[1] = entries;
}
static __global__ __launch_bounds__(64) void k_promote_publish(const u32* __restrict__ plist, const u32* __restrict__ hot, u32* __restrict__ kwide, u32* __restrict__ wowner) {
    const u32 i = threadIdx.x;
    if (i >= promote_live(hot)) return;
    const u32 slot = plist[2 * i];
    if (slot != 0xFFFFFFFFu) { kwide[slot] = plist[2 * i + 1]; wowner[plist[2 * i + 1]] = slot; }
}

static __global__ __launch_bounds__(256) void k_pack_bitmap(const uint8_t* __restrict__ acc, size_t n, uint8_t* __restrict__ bitmap) {
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;           // bitmap byte
    if (b >= ((n + 7) >> 3)) return;
    u32 v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = b * 8 + k;
        if (i < n && acc[i] == 1) v |= 1u << k;       // exactly "accepted": a tuple left in a transient state (SBV_ED_PENDING) by a step that never finished it is a REJECT
    }
    bitmap[b] = (uint8_t)v;
}

// Compaction shared by both curves: `ung` / `grp` / `rejected` are this lane's class; one atomicAdd per
// workgroup and list (atomicAdds on one word serialise at ~11 ns each on MI355X).
__device__ __forceinline__ void group_split_emit(size_t i, u32 s, bool ung, bool grp, bool rejected, const GroupState& g) {
    __shared__ u32 sh_cnt[3][4];
    __shared__ u32 sh_base[2];
    const unsigned long long mu = __ballot(ung), mg = __ballot(grp), mr = __ballot(rejected);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh_cnt[0][wave] = (u32)__popcll(mu); sh_cnt[1][wave] = (u32)__popcll(mg); sh_cnt[2][wave] = (u32)__popcll(mr); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 tu = sh_cnt[0][0] + sh_cnt[0][1] + sh_cnt[0][2] + sh_cnt[0][3];
        const u32 tg = sh_cnt[1][0] + sh_cnt[1][1] + sh_cnt[1][2] + sh_cnt[1][3];
        const u32 tr = sh_cnt[2][0] + sh_cnt[2][1] + sh_cnt[2][2] + sh_cnt[2][3];
        sh_base[0] = tu ? atomicAdd(&g.counters[2], tu) : 0u;
        sh_base[1] = tg && !g.sorted ? atomicAdd(&g.counters[1], tg) : 0u;      // sorted: the counting sort owns counters[1] and grp_idx
        if (tr) atomicAdd(&g.counters[3], tr);
    }
    __syncthreads();
    u32 base_u = sh_base[0], base_g = sh_base[1];
    for (int w = 0; w < wave; ++w) { base_u += sh_cnt[0][w]; base_g += sh_cnt[1][w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (ung) g.ung_idx[base_u + (u32)__popcll(mu & below)] = (u32)i;
    if (ung || rejected) g.slots[i] = SBV_GROUP_NONE;
    if (grp) {
        g.slots[i] = s;
        if (!g.sorted) g.grp_idx[base_g + (u32)__popcll(mg & below)] = (u32)i;
    }
}

// Compaction of one class with one atomicAdd per workgroup (256 lanes): returns this lane's position in the list, valid
// where `mine` is set.
__device__ __forceinline__ u32 group_compact_pos(bool mine, u32* counter) {
    __shared__ u32 cp_cnt[4];
    __shared__ u32 cp_base;
    const unsigned long long m = __ballot(mine);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) cp_cnt[wave] = (u32)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 t = cp_cnt[0] + cp_cnt[1] + cp_cnt[2] + cp_cnt[3];
        cp_base = t ? atomicAdd(counter, t) : 0u;
    }
    __syncthreads();
    u32 base = cp_base;
    for (int w = 0; w < wave; ++w) base += cp_cnt[w];
    return base + (u32)__popcll(m & ((1ull << lane) - 1ull));
}
// key-sorted step, pass 1 (group_classify_lane): the group of every tuple; ungrouped tuples become candidates
// (P-256: list = ung_cand, counter = counters[4]; Ed25519 has no key check in front of the one-lane kernel: ung_idx, counters[2])
static __global__ __launch_bounds__(256) void k_group_classify(size_t n, GroupState g, u32* __restrict__ list, u32* __restrict__ counter) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n;
    u32 s = SBV_GROUP_NONE;
    if (active) { s = g.slot_of[g.rep[i]]; g.slots[i] = s; }
    const bool cand = active && s == SBV_GROUP_NONE;
    const u32 pos = group_compact_pos(cand, counter);
    if (cand) list[pos] = (u32)i;
}

// ---- key-sorted grouped list (p256_group.h: group_sort_*) ----------------------------------------------------------------
// One workgroup of 1024 lanes per SBV_SORT_TILE tuples; dynamic LDS = one u32 per group.  A tile sees each of ~1000 keys a
// handful of times, so the LDS histogram turns a million global atomics on ~1000 hot words into groups x tiles of them.
#define SBV_SORT_PER_LANE 8
#define SBV_SORT_TILE (1024 * SBV_SORT_PER_LANE)
// Groups one LDS histogram holds (64 KiB of dynamic LDS).  A batch with more groups (round 5: up to 65 536) takes plain global atomics
// instead, which is what the histogram degenerates to when a tile sees each key once.  The kernels decide ON THE DEVICE (the group
// count is known there only): with few hot keys the direct form serialises on a handful of words (16 consenter keys: 0.4 ms per 2^18
// tuples, measured in round 5).  One launch either way: as two kernels per step the empty one still cost its launch on the sort's chain.
#define SBV_SORT_LDS_GROUPS 16384u
static __global__ __launch_bounds__(1024) void k_group_sort_count(size_t n, GroupState g) {
    extern __shared__ u32 sort_lh[];
    const u32 groups = group_count(g);
    if (groups == 0) return;
    const size_t base = (size_t)blockIdx.x * SBV_SORT_TILE;
    if (groups > SBV_SORT_LDS_GROUPS) {           // more groups than one histogram holds: plain global atomics (wave-uniform branch)
#pragma unroll
        for (int q = 0; q < SBV_SORT_PER_LANE; ++q) {
            const size_t i = base + (size_t)q * 1024 + threadIdx.x;
            if (i < n) group_sort_count_lane(i, g);
        }
        return;
    }
    for (u32 k = threadIdx.x; k < groups; k += 1024) sort_lh[k] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SBV_SORT_PER_LANE; ++q) {
        const size_t i = base + (size_t)q * 1024 + threadIdx.x;
        if (i < n) {
            const u32 s = g.slots[i];
            if (s < groups) atomicAdd(&sort_lh[s], 1u);
        }
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < groups; k += 1024) {
        const u32 c = sort_lh[k];
        if (c) atomicAdd(&g.gcount[k], c);
    }
}
// one workgroup: exclusive scan of the group counts
static __global__ __launch_bounds__(1024) void k_group_sort_scan(GroupState g) {
    __shared__ u32 part[1024];
    const u32 groups = group_count(g);
    const u32 rows = group_sort_rows(groups), P = group_sort_positions(groups);      // the runs' order in the list: p256_group.h
    const u32 per = (P + 1023) / 1024;
    const u32 lo = threadIdx.x * per;
    u32 hi = lo + per;
    if (hi > P) hi = P;
    u32 sum = 0;
    for (u32 p = lo; p < hi; ++p) { const u32 k = group_sort_group_at(p, rows); if (k < groups) sum += g.gcount[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (u32 d = 1; d < 1024; d <<= 1) {            // Hillis-Steele inclusive scan
        const u32 v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    u32 run = part[threadIdx.x] - sum;
    for (u32 p = lo; p < hi; ++p) { const u32 k = group_sort_group_at(p, rows); if (k < groups) { g.gcursor[k] = run; run += g.gcount[k]; } }
    if (threadIdx.x == 1023) g.counters[1] = part[1023];
}
static __global__ __launch_bounds__(1024) void k_group_sort_scatter(size_t n, GroupState g) {
    extern __shared__ u32 sort_lh[];
    const u32 groups = group_count(g);
    if (groups == 0) return;
    const size_t base = (size_t)blockIdx.x * SBV_SORT_TILE;
    if (groups > SBV_SORT_LDS_GROUPS) {
#pragma unroll
        for (int q = 0; q < SBV_SORT_PER_LANE; ++q) {
            const size_t i = base + (size_t)q * 1024 + threadIdx.x;
            if (i < n) group_sort_scatter_lane(i, g);
        }
        return;
    }
    for (u32 k = threadIdx.x; k < groups; k += 1024) sort_lh[k] = 0;
    __syncthreads();
    u32 grp[SBV_SORT_PER_LANE], rank[SBV_SORT_PER_LANE];
#pragma unroll
    for (int q = 0; q < SBV_SORT_PER_LANE; ++q) {
        const size_t i = base + (size_t)q * 1024 + threadIdx.x;
        grp[q] = SBV_GROUP_NONE;
        rank[q] = 0;
        if (i < n) {
            const u32 s = g.slots[i];
            if (s < groups) { grp[q] = s; rank[q] = atomicAdd(&sort_lh[s], 1u); }
        }
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < groups; k += 1024) {
        const u32 c = sort_lh[k];
        if (c) sort_lh[k] = atomicAdd(&g.gcursor[k], c);      // the tile's piece of group k's run
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SBV_SORT_PER_LANE; ++q) {
        if (grp[q] == SBV_GROUP_NONE) continue;
        const u32 L = sort_lh[grp[q]] + rank[q];
        g.grp_idx[L] = (u32)(base + (size_t)q * 1024 + threadIdx.x);
        g.grp_of[L] = grp[q];
    }
}

}  // namespace sbv
