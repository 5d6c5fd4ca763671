// p256_group_kernels.hip — kernels for grouping a generic batch by public key inside the step
// (p256_group.h) and for running stage B over the two compacted index lists.
//
//   k_group_insert / k_group_assign / k_group_split : one lane per tuple; global atomics only
//   k_keytab_bases   : one lane per grouped key  (256 doublings: the latency floor of a fresh key), in chunks of windows
//   k_keytab_window  : a few lanes per (key, window): additions + Montgomery-trick normalisation each,
//                      Jacobian intermediates parked in a private strip of HBM
//   k_gphase_generic : u1*G for every tuple (independent of the keys) + the generic stage B over ung_idx
//   k_verify_keyed_q : the key-comb additions over grp_idx; one accept BYTE per tuple from either kernel
//   k_pack_bitmap    : accept bytes -> LSB-first bitmap
// Launch sizes that depend on device-side counters use the upper bound; surplus lanes exit at once.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "group_kernels_common.h"
#include "p256_kernels.h"
#include "p256_keytab29.h"

// waves per SIMD the comb kernels of the carry-free field are compiled for: the mixed addition keeps ~160 live registers
// (XYZZ accumulator 36, two table entries in flight 32, 17 64-bit columns 34, temporaries); at 3 waves (168 VGPRs) hipcc
// spills ~60 dwords per lane in the Q phase, at 2 waves nothing, and tools/febench measures the addition chain no slower
// at 2 waves than at 3 (the arithmetic has no carry chains to hide).
// The G phase of the key-sorted step also exists as a kernel of its own (k_gphase_sorted): without the one-lane kernel's code
// in the same launch it needs 168 VGPRs and spills nothing at 3 waves per SIMD, and while the table kernels hold registers
// beside it two of its waves still fit on a SIMD instead of one (2^20 cold 3.44 -> 3.35 ms, profiles/r03/ab_gphase_split_r03t.jsonl;
// 4 waves: 77 spilled dwords, 3.78 ms).  The ungrouped list then needs a launch in front of it, which costs 0.05 ms of a
// 0.9 ms step at 2^18: GroupSync::gsplit_min picks the form by batch size.
#ifndef SBV_GPHASE_WAVES
#define SBV_GPHASE_WAVES 3
#endif
#ifndef SBV_COMB29_WAVES
#define SBV_COMB29_WAVES 2
#endif
// Q phase (round 6): the chunks' launches and the wide pass fit 168 VGPRs without a spill once each form is a kernel of its own
// (k_verify_keyed_q<MODE, LAST>) — 3 waves per SIMD like the G phase; the rows-only pass (two table entries in flight per window) stays at 2
#ifndef SBV_QPHASE_WAVES
#define SBV_QPHASE_WAVES 3
#endif

namespace sbv {

__global__ __launch_bounds__(256) void k_group_insert(const uint8_t* __restrict__ tuples, size_t n, GroupState g) {
    group_insert_block<160, 96, 16>(tuples, n, g);
}
// Same result as group_split_lane (compaction: group_split_emit)
__global__ __launch_bounds__(256) void k_group_split(const uint8_t* __restrict__ tuples, size_t n, GroupState g, uint8_t* __restrict__ acc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n;
    u32 s = SBV_GROUP_NONE;
    if (active) s = g.slot_of[g.rep[i]];
    bool ung = active && s == SBV_GROUP_NONE;
    const bool grp = active && s != SBV_GROUP_NONE;
    bool key_rejected = false;
    if (ung) {                      // key filter of group_split_lane
        if (!tuple_key_ok(tuples, i)) { acc[i] = 0; ung = false; key_rejected = true; }
    }
    group_split_emit(i, s, ung, grp, key_rejected, g);
}

// key-sorted step, pass 2 (group_keycheck_lane): pointFromAffine on the compacted candidates -> ung_idx, or rejected
__global__ __launch_bounds__(256) void k_group_keycheck(const uint8_t* __restrict__ tuples, GroupState g, uint8_t* __restrict__ acc) {
    const u32 L = blockIdx.x * 256 + threadIdx.x;
    const u32 cands = g.counters[4];
    if (blockIdx.x * 256u >= cands) return;            // whole workgroup idle (uniform: the barriers below are not reached by anyone)
    const bool active = L < cands;
    u32 i = 0;
    bool ok = false;
    if (active) {
        i = g.ung_cand[L];
        ok = tuple_key_ok(tuples, i);
        if (!ok) acc[i] = 0;
    }
    const unsigned long long mr = __ballot(active && !ok);
    if ((threadIdx.x & 63) == 0 && mr) atomicAdd(&g.counters[3], (u32)__popcll(mr));
    const u32 pos = group_compact_pos(active && ok, &g.counters[2]);
    if (active && ok) g.ung_idx[pos] = i;
}

// ---- per-batch key tables on the carry-free field (p256_keytab29.h) -----------------------------------------------------
// tmp layout (u32 words): one strip of SBV_KT29_WINDOW_TMP words per (key, window), shared by the rows and the fill kernel
// (same stream, never concurrent).
#define SBV_KT29_WINDOW_TMP (7 * SBV_KT29_FILL_TMP_WORDS)
// The table kernels are a few hundred lanes of latency-bound chains on the critical path of the step, sharing SIMDs with
// the throughput kernels (G phase, Q phase, stage A): raise their wave priority so the arbiter issues them first.
#ifndef SBV_TABLE_PRIO
#define SBV_TABLE_PRIO 0
#endif
static __device__ __forceinline__ void table_prio() {
#if SBV_TABLE_PRIO > 0
    __builtin_amdgcn_s_setprio(SBV_TABLE_PRIO);
#endif
}

// Exchange policy of the chain on the device: one lane of a quad; a value of lane `src` of the quad reaches all four lanes
// by DPP quad_perm (v_mov_b32_dpp, a full-rate register move; control = the source lane in all four 2-bit fields).
struct keychain_quad_dev {
    static const int N = 1;
    kchain s[1];
    int r;
    __device__ __forceinline__ int role(int) const { return r; }
    __device__ __forceinline__ void bcast(fe29 out[1], const fe29 in[1], int src) const {
        SBV_UNROLL
        for (int l = 0; l < 9; ++l) {
            const int v = in[0].v[l];
            out[0].v[l] = src == 0 ? __builtin_amdgcn_mov_dpp(v, 0x00, 0xF, 0xF, true)
                        : src == 1 ? __builtin_amdgcn_mov_dpp(v, 0x55, 0xF, 0xF, true)
                                   : __builtin_amdgcn_mov_dpp(v, 0xAA, 0xF, 0xF, true);
        }
    }
};
// The table kernels run on BOUNDED grids (round 5): a launch holds at most SBV_TABLE_GRID_BLOCKS workgroups of 64 lanes and every
// lane walks its share of the work items in a grid-stride loop.  The number of groups is known on the device only, and a batch may
// now hold up to 65 536 of them (sbv_p256_set_grouping): a grid sized for the capacity would dispatch ~140 000 empty workgroups per
// launch on the headline batch (1024 groups), and scratch indexed by (key, window) would take 44 GB.  With the bound, the headline
// batch still runs every item in one pass (its 2 176 fill workgroups fit), a 16 384-key batch loops, and the per-lane scratch of the
// rows / fill steps is indexed by the RESIDENT lane: 2 parities x 4096 x 64 lanes x 675 words = 1.4 GB.
#define SBV_TABLE_GRID_BLOCKS 4096u
// lanes = groups x 4 (p256_keytab29.h: keychain29_run); a quad lives or exits as a whole
__global__ __launch_bounds__(64) void k_keytab29_chain(const uint8_t* __restrict__ tuples, GroupState g, u32* __restrict__ jstate,
                                                       u32* __restrict__ bases, uint8_t* __restrict__ valid,
                                                       const u32* __restrict__ tslot, const uint8_t* __restrict__ cold,
                                                       int j_first, int j_last, u32 rec_mask) {
    const u32 groups = group_count(g);
    for (u32 lane = blockIdx.x * 64 + threadIdx.x; (lane >> 2) < groups; lane += gridDim.x * 64) {
        const u32 k = lane >> 2;
        if (!cold[k]) continue;
        if (j_first > 0 && valid[tslot[k]] == 0) continue;      // the first chunk found the key invalid: no table (p256_keytab29.h)
        table_prio();
        keychain_quad_dev q;
        q.r = (int)(lane & 3u);
        keychain29_run(q, tuples, k, g, jstate, bases, valid + tslot[k], j_first, j_last, rec_mask);
    }
}
// lanes = groups x j_count x 2: lane 0 of a window builds the babies b B_j, b = 1..8, lane 1 the giants 16 a B_j, a = 1..8
// tmp: this launch's scratch, SBV_KT29_ROWS_TMP_WORDS words per resident lane
#ifndef SBV_ROWS_WAVES
#define SBV_ROWS_WAVES 2
#endif
__global__ __launch_bounds__(64, SBV_ROWS_WAVES) void k_keytab29_rows(GroupState g, const u32* __restrict__ bases, u32* __restrict__ tmp,
                                                      apt* __restrict__ ktab, apt* __restrict__ ntab, const u32* __restrict__ tslot,
                                                      const uint8_t* __restrict__ cold, const uint8_t* __restrict__ kvalid, int j_first, int j_count) {
    const u32 total = group_count(g) * (u32)j_count * 2u;               // <= 65 536 x 33 x 2: fits 32 bits
    for (u32 base = blockIdx.x * 64; base < total; base += gridDim.x * 64) {        // the loop state is wave-uniform: it lives in scalar registers
        const u32 lane = base + threadIdx.x;
        if (lane >= total) break;
        const u32 which = lane & 1u;
        const u32 kw = lane >> 1;
        const u32 key = kw / (u32)j_count, j = (u32)j_first + kw % (u32)j_count;
        if (!cold[key] || kvalid[tslot[key]] == 0) continue;            // nothing to build, or a key that is no point: no table
        if (which == 1 && j == SBV_GTAB_WINDOWS - 1) continue;          // the top window has no giants
        table_prio();
        const size_t w = (size_t)key * SBV_GTAB_WINDOWS + j;
        u32* t = tmp + (size_t)(blockIdx.x * 64 + threadIdx.x) * SBV_KT29_ROWS_TMP_WORDS;
        keytab29_rows_lane(bases + w * (SBV_KT29_POINTS_PER_WINDOW * SBV_KT29_REC_WORDS), (int)which, j == SBV_GTAB_WINDOWS - 1, t,
                           ktab + ((size_t)tslot[key] * SBV_GTAB_WINDOWS + j) * SBV_GTAB_PER_WINDOW,
                           ntab + ((size_t)tslot[key] * SBV_GTAB_WINDOWS + j) * SBV_NTAB_PER_WINDOW);
    }
}

// The fill step (symmetric form, p256_keytab29.h): lanes = groups x j_count x 8, lane a - 1 of a window fills both sides of giant 16 a.
// Measured against round 3's one-sided fill (15 denominators per lane, babies 1..16 from the rows step) in round 4
// (profiles/r04/ab_chunk0_symfill_r04a.jsonl): cold 2^20 3.34 -> 3.22 ms, 2^19 2.23 -> 2.08, 2^18 1.77 -> 1.62; the one-sided
// kernel and the two wide forms of round 3 (one lane per entry; rows split over lanes — both measured slower) left the library.
// Round 5: only for the groups that earn a full table (needfill, p256_group.h: group_table_class_lane); tmp: 72 words per resident lane.
#define SBV_KT29_FILL_LANE_WORDS (8 * 9)
__global__ __launch_bounds__(64) void k_keytab29_fill_sym(GroupState g, u32* __restrict__ tmp, apt* __restrict__ ktab,
                                                          const u32* __restrict__ tslot, const uint8_t* __restrict__ needfill,
                                                          const uint8_t* __restrict__ kvalid, int j_first, int j_count) {
    const u32 total = group_count(g) * (u32)j_count * 8u;               // <= 65 536 x 33 x 8: fits 32 bits
    for (u32 base = blockIdx.x * 64; base < total; base += gridDim.x * 64) {
        const u32 lane = base + threadIdx.x;
        if (lane >= total) break;
        const u32 r = lane & 7u;
        const u32 kw = lane >> 3;
        const u32 key = kw / (u32)j_count, j = (u32)j_first + kw % (u32)j_count;
        if (j == SBV_GTAB_WINDOWS - 1 || !needfill[key] || kvalid[tslot[key]] == 0) continue;
        table_prio();
        u32* t = tmp + (size_t)(blockIdx.x * 64 + threadIdx.x) * SBV_KT29_FILL_LANE_WORDS;
        keytab29_fill_sym_lane(1 + (int)r, t, ktab + ((size_t)tslot[key] * SBV_GTAB_WINDOWS + j) * SBV_GTAB_PER_WINDOW);
    }
}
// The same with ONE inversion per window (round 6): a workgroup takes SBV_FILL_SHARED_BLOCK / 8 windows; every lane leaves the product of its
// eight denominators in LDS, the first lanes turn the windows' 8 products each into their inverses with one inversion per window
// (keytab29_fill_group_inverses: the inversions of the workgroup's wavefronts packed into one), every lane walks back with its own.
// 256 lanes = 4 wavefronts = 32 windows per workgroup: 512 packs the inversions 8 : 1 but needs eight free wave slots on ONE CU beside the G
// phase (cold 3.25-3.28 ms), 128 packs 2 : 1 (3.24-3.28); 256: 3.20-3.21 (profiles/r06/ab_fill_block_r06bg.jsonl; one per lane: 3.27-3.32)
#ifndef SBV_FILL_SHARED_BLOCK
#define SBV_FILL_SHARED_BLOCK 256
#endif
__global__ __launch_bounds__(SBV_FILL_SHARED_BLOCK) void k_keytab29_fill_shared(GroupState g, u32* __restrict__ tmp, apt* __restrict__ ktab,
                                                          const u32* __restrict__ tslot, const uint8_t* __restrict__ needfill,
                                                          const uint8_t* __restrict__ kvalid, int j_first, int j_count) {
    __shared__ u32 sh[(SBV_FILL_SHARED_BLOCK / 8) * 16 * 9];
    const u32 total = group_count(g) * (u32)j_count * 8u;
    const u32 tid = threadIdx.x;
    u32* w = sh + (tid >> 3) * (16 * 9);
    for (u32 base = blockIdx.x * SBV_FILL_SHARED_BLOCK; base < total; base += gridDim.x * SBV_FILL_SHARED_BLOCK) {      // uniform in the workgroup
        const u32 lane = base + tid;
        const u32 r = lane & 7u;
        const u32 kw = lane >> 3;
        bool active = lane < total;
        u32 key = 0, j = 0;
        if (active) {
            key = kw / (u32)j_count; j = (u32)j_first + kw % (u32)j_count;
            active = !(j == SBV_GTAB_WINDOWS - 1 || !needfill[key] || kvalid[tslot[key]] == 0);
        }
        if (!__syncthreads_or(active ? 1 : 0)) continue;            // nothing to fill in these 64 windows (cached keys)
        u32* t = tmp + (size_t)(blockIdx.x * SBV_FILL_SHARED_BLOCK + tid) * SBV_KT29_FILL_LANE_WORDS;
        apt* row = active ? ktab + ((size_t)tslot[key] * SBV_GTAB_WINDOWS + j) * SBV_GTAB_PER_WINDOW : nullptr;
        fe29 acc = f29_one();
        if (active) { table_prio(); keytab29_fill_sym_acc(1 + (int)r, t, row, acc); }
        f29_store_raw(w + r * 9, acc);
        __syncthreads();
        if (tid < SBV_FILL_SHARED_BLOCK / 8) keytab29_fill_group_inverses(sh + tid * (16 * 9));
        __syncthreads();
        if (active) {
            fe29 inv;
            f29_load_raw(inv, w + r * 9);
            keytab29_fill_sym_finish(1 + (int)r, t, row, inv);
        }
        __syncthreads();                                             // the next round rewrites sh
    }
}
// scratch words of ONE launch of the rows / fill kernels (the launcher gives the two table streams a region each)
#define SBV_TABLE_TMP_WORDS ((size_t)SBV_TABLE_GRID_BLOCKS * 64 * SBV_KT29_ROWS_TMP_WORDS)

// Table classes of the batch's groups and, at the end of the step, what the slots hold (p256_group.h).  One lane per group.
// hot: the wide-comb state of the cache slots (p256_group.h: hot keys); kwide == nullptr = feature off / no pool
__global__ __launch_bounds__(256) void k_group_table_class(GroupState g, const u32* __restrict__ tslot, const uint8_t* __restrict__ cold,
                                                           const uint8_t* __restrict__ kfull, u32 table_slots, u32 full_min,
                                                           uint8_t* __restrict__ full, uint8_t* __restrict__ needfill,
                                                           HotKeys hk, uint8_t* __restrict__ wide) {
    const u32 groups = group_count(g);
    for (u32 k = blockIdx.x * 256 + threadIdx.x; k < groups; k += gridDim.x * 256) {
        group_table_class_lane(k, g, tslot, cold, kfull, table_slots, full_min, full, needfill);
        if (hk.kwide && g.sorted) group_hot_class_lane(k, g, tslot, cold, hk.cache_cap, hk.kwide, hk.khits, wide);
        else wide[k] = 0;
        if (full[k]) atomicAdd(&g.counters[5], 1u);           // sbv_p256_last_table_classes; and [5] == groups tells the rows-only pass that it has no wavefront
        if (needfill[k]) atomicAdd(&g.counters[6], 1u);
        if (wide[k]) atomicAdd(&g.counters[8], 1u);           // [8] == 0 tells the wide pass the same
    }
}
// ---- promotion of hot cache slots to wide combs (p256_group.h: hot keys; p256_widetab29.h: the builder of the registered path) ----------
// select (one lane per group) -> bases (the 2 x 17 base points of each promotion, gathered from the key's 8-bit table) -> chains + fill
// (the builder's lanes, for the promotions this batch really made) -> publish (kwide[slot] = index: later batches take the wide pass)
__global__ __launch_bounds__(64) void k_promote_bases(const u32* __restrict__ plist, const u32* __restrict__ hot, const apt* __restrict__ ktab, apt* __restrict__ pbases) {
    const u32 lane = blockIdx.x * 64 + threadIdx.x;
    const u32 W = (257 + SBV_HOT_BITS - 1) / SBV_HOT_BITS;          // 17
    const u32 i = lane / (2 * W), e = lane % (2 * W);
    if (i < promote_live(hot)) promote_base_lane(i, e, plist, ktab, pbases);
}
__global__ __launch_bounds__(64) void k_promote_chains(const apt* __restrict__ pbases, const u32* __restrict__ plist, const u32* __restrict__ hot, widebuild w,
                                                       size_t stride, u32* __restrict__ tmp, apt* __restrict__ wtab) {
    const u32 lane = blockIdx.x * 64 + threadIdx.x;
    const u32 per_key = (u32)w.windows * 2u;
    const u32 i = lane / per_key, j = (lane % per_key) >> 1, role = lane & 1u;
    if (i >= promote_live(hot) || plist[2 * i] == 0xFFFFFFFFu) return;
    const apt* kb = pbases + (size_t)i * per_key;
    widetab_chain_role(w, kb + j, kb + w.windows + j, (int)role, tmp + (size_t)lane * widebuild_chain_len(w) * SBV_WIDETAB_REC_WORDS,
                       wtab + (size_t)plist[2 * i + 1] * stride + (size_t)j * w.per_window);
}
__global__ __launch_bounds__(256) void k_promote_fill(const u32* __restrict__ plist, const u32* __restrict__ hot, widebuild w, size_t stride, apt* __restrict__ wtab) {
    const size_t lane = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 chunks = widebuild_fill_chunks(w);
    const size_t per_window = (size_t)(w.giants - 1) * chunks;
    const size_t per_key = per_window * (size_t)w.windows;
    const u32 i = (u32)(lane / per_key);
    if (i >= promote_live(hot) || plist[2 * i] == 0xFFFFFFFFu) return;
    const size_t r = lane % per_key;
    const u32 j = (u32)(r / per_window);
    const size_t q = r % per_window;
    const u32 gi = 1u + (u32)(q / chunks), c = (u32)(q % chunks);
    widetab_fill_lane(w, gi, 1u + c * SBV_WIDETAB_T, wtab + (size_t)plist[2 * i + 1] * stride + (size_t)j * w.per_window);
}
__global__ __launch_bounds__(256) void k_group_table_mark(GroupState g, const u32* __restrict__ tslot, const uint8_t* __restrict__ cold,
                                                          const uint8_t* __restrict__ full, const uint8_t* __restrict__ needfill, u32 table_slots,
                                                          uint8_t* __restrict__ kfull) {
    const u32 groups = group_count(g);
    for (u32 k = blockIdx.x * 256 + threadIdx.x; k < groups; k += gridDim.x * 256) group_table_mark_lane(k, tslot, cold, full, needfill, table_slots, kfull);
}
// One launch, two jobs.  Blocks [0, generic_blocks): the generic stage B (doubling kernel) over the
// ungrouped list — keys that repeat too rarely for a table; a 2.3 ms serial chain per lane on ~5 % of the
// tuples, so it has to start as early as possible and run BESIDE the throughput work, at the same
// 3 waves/SIMD register budget (on its own stream it ran at 234 VGPRs and squeezed the kernel it
// overlapped down to one wave per SIMD).  Remaining blocks: u1 * G for every tuple of the batch.
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_COMB29_WAVES) void k_gphase_generic(const uint8_t* __restrict__ tuples, Scratch s, size_t n, GroupState g, u32* __restrict__ qtab,
                                                                       const apt* __restrict__ g16, gcomb g16r,
                                                                       u32* __restrict__ gacc, uint8_t* __restrict__ acc,
                                                                       unsigned generic_blocks, size_t first, size_t end) {
    if (blockIdx.x < generic_blocks) {
        const u32 L = blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
        if (L >= g.counters[2]) return;
        const u32 t = g.ung_idx[L];
        const bool v = s.rec ? verify29_lane_generic_rec(s, tuples, t, qtab + (size_t)L * SBV_QTAB29_WORDS, g16r)
                             : verify29_lane_generic(s, t, qtab + (size_t)L * SBV_QTAB29_WORDS, g16r);
        acc[t] = v ? 1 : 0;
        return;
    }
    if (group_count(g) == 0) return;            // no key repeats often enough (e.g. all-distinct keys): nothing will read gacc
    const size_t i = first + (size_t)(blockIdx.x - generic_blocks) * SBV_VERIFY_BLOCK + threadIdx.x;      // this launch: tuples [first, end)
    if (g.sorted) {                             // lanes of the key-sorted list: only grouped tuples, accumulator parked at the lane's position
        if (i < g.counters[1]) gphase29_lane_sorted(s, g.grp_idx[i], i, g16r, gacc);
        return;
    }
    if (i < end) gphase29_lane(s, i, g16r, gacc);
}

// The G phase of the key-sorted step alone: lane L of the sorted list
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_GPHASE_WAVES) void k_gphase_sorted(Scratch s, GroupState g, gcomb g16r, u32* __restrict__ gacc) {
    const size_t i = (size_t)blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    if (i < g.counters[1]) gphase29_lane_sorted(s, g.grp_idx[i], i, g16r, gacc);
}

// Q phase over the grouped list: windows [j0, j1) of the per-batch key combs.
// Round 5 — one instantiation per table class, every wavefront served by exactly one of them (all three evaluate the same predicate on
// the same data):
//   MODE 2 (wide)    ALL lanes' keys own a wide comb (hot cache slots, p256_group.h): u2 * Q in 17 additions from the 16-bit comb, ONE launch
//                    right behind the G phase — it needs no table of this batch;
//   MODE 0 (full)    otherwise, ALL lanes' keys own a full 8-bit comb (wide or not): the launches of the chunks, one addition per window, as before;
//   MODE 1 (narrow)  any other wavefront — a key with rows only, or a mix at the seam of two runs: ONE launch behind the last rows, windows
//                    0..32 from the compact rows (babies and giants), two additions per window.
// A lane whose key pointFromAffine refuses (or that has no slot) is "dead": rejected whatever is added, it never decides its wavefront's class.
// (a key with a wide comb keeps its 8-bit table, FULL OR ROWS ONLY — promotion does not ask which: in a mixed wavefront it counts as
// what that table is)
__device__ __forceinline__ int q_wave_class(bool dead, bool w, bool f) {
    return group_wave_class(wave_all(dead), wave_all(dead || w), wave_all(dead || f));      // the rule itself: p256_group.h, shared with tests/emul
}
// Round 6: LAST (the launch that turns the sum into a verdict) is a template parameter, and the compaction-order form is a kernel of its
// own (k_verify_keyed_q_list).  With `last` a run-time flag one kernel held both continuations of the comb loop — park the accumulator,
// or compare R.x with r — and the compaction-order copy of the loop beside them: 247 VGPRs, 2 waves per SIMD (128 spilled dwords at 3,
// rounds 3-5).  Each of the forms alone is the G phase's loop with another table: 168 VGPRs, nothing spilled, 3 waves per SIMD.
template <int MODE, bool LAST>
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, MODE == SBV_Q_NARROW ? SBV_COMB29_WAVES : SBV_QPHASE_WAVES) void k_verify_keyed_q(Scratch s, GroupState g, const apt* __restrict__ ktab,
                                                                    const uint8_t* __restrict__ kvalid, const u32* __restrict__ tslot,
                                                                    const uint8_t* __restrict__ full, const uint8_t* __restrict__ wide, widekeys wk, u32* __restrict__ wstat,
                                                                    u32 table_slots, u32* __restrict__ gacc,
                                                                    uint8_t* __restrict__ acc, int j0, int j1) {
    // The rows-only and the wide pass are launched beside the chunks' launches for every batch, over the whole grouped list; in most
    // batches — the headline's 1 024 hot signers, a consenter replay — no group is of their class, and every wavefront used to load its
    // lane's group, slot and class bytes to find that out (16 384 wavefronts alive for 0.8 ms beside the Q phase).  The
    // class kernel counts the groups of each class: one scalar load per workgroup settles it.
    if (MODE == SBV_Q_NARROW && g.counters[5] >= group_count(g)) return;      // every group owns a full table: no wavefront can be rows-only
    if (MODE == SBV_Q_WIDE && g.counters[8] == 0) return;                     // no group may take the wide pass
    // Key-sorted list: consecutive blocks hold consecutive keys.  The dispatcher deals workgroups round-robin over the
    // 8 XCDs (block b -> XCD b % 8, MI355X_MICROARCH.md "Workgroup dispatch"; a speed assumption only), so block b takes
    // logical block (b % 8) * per + b / 8: every XCD walks its own contiguous eighth of the list and a key's comb rows are
    // fetched into ONE L2.  `per` comes from the live lane count, not the launch's upper bound, so the eighths are even.
    // (The runs of the list come in the order 0, 8, 16, ..., 1, 9, ... of their groups since round 6 — p256_group.h: group_sort_group_at —
    // so that a class of groups, e.g. the hot keys, is spread over all eight XCDs.  Dealing the list out in 64 chunks per XCD instead
    // was measured too: the Q launch 936 -> 996 us, profiles/r06/ab_p256_chunked_xcd_r06an.jsonl.)
    const u32 lanes = g.counters[1];
    const u32 per = ((lanes + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK + 7) >> 3;
    const u32 local = blockIdx.x >> 3;
    if (local >= per) return;
    const u32 L = ((blockIdx.x & 7u) * per + local) * SBV_VERIFY_BLOCK + threadIdx.x;
    if (L >= lanes) return;
    const u32 t = g.grp_idx[L];
    const u32 grp = g.grp_of[L];
    const bool known = grp < group_count(g);
    const u32 ts = known ? tslot[grp] : SBV_GROUP_NONE;
    const bool dead = !(ts < table_slots) || kvalid[ts < table_slots ? ts : 0u] == 0;     // no slot, or a key that is no point: "reject" whatever is added
    const int cls = q_wave_class(dead, !dead && wide[grp] != 0, !dead && full[grp] != 0);
    if (cls == SBV_Q_NONE) { if (MODE == SBV_Q_FULL && LAST) acc[t] = 0; return; }                 // rejected without touching a table (there is none)
    if (cls != MODE) return;                                                              // another instantiation's wavefront
    if (MODE != SBV_Q_FULL) {                                                             // statistics only: lanes of the rows-only / the wide pass
        const unsigned long long am = __ballot(true);
        if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)am) - 1u) atomicAdd(MODE == SBV_Q_NARROW ? &g.counters[7] : wstat, (u32)__popcll(am));
    }
    if (MODE == SBV_Q_WIDE) {
        u256 u2, r;
        rec_load256(u2, s.rec, t, SBV_REC_U2);
        xyzz R;
        gacc29_load(R, gacc, s.cap, L);
        wide_qphase29_point(R, u2, wk, dead ? 0u : wk.idx[ts]);                           // a dead lane of a wide wavefront walks comb 0: its verdict is false anyway
        rec_load256(r, s.rec, t, SBV_REC_R);
        acc[t] = !dead && s.rec[(size_t)t * SBV_REC_WORDS + SBV_REC_OK] != 0 && pt29_rx_matches(R, r) ? 1 : 0;
        return;
    }
    const bool v = qphase29_lane_sorted<MODE == SBV_Q_NARROW>(s, t, L, ts, table_slots, ktab, kvalid, gacc, j0, j1, LAST);
    if (LAST) acc[t] = v ? 1 : 0;
}
// ---- the LDS-staged form of the chunks' launches (round 6; VERDICT r5 #3, north_star "LDS-staged ... tables") ------------------------
// A workgroup of the key-sorted list nearly always holds ONE key (256 lanes in runs of ~1000).  Such a workgroup stages the window's
// row — 128 entries, 8 KB — in LDS with two coalesced 16-byte loads per lane instead of 256 random 64-byte gathers through the vector
// cache, two buffers (16 KB per workgroup, 48 KB per CU at 3 workgroups), one barrier per window: row j + 1 is fetched into registers
// while the additions of row j run, and written to the other buffer behind them.  A workgroup at the seam of two keys (or with a
// rows-only / wide wavefront) takes the ordinary path.  Opt-in (SBV_QPHASE_LDS=1): measured against the gathers in
// profiles/r06/ab_qphase_lds_*; DESIGN.md section 7 has the verdict.
template <bool LAST>
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_QPHASE_WAVES) void k_verify_keyed_q_lds(Scratch s, GroupState g, const apt* __restrict__ ktab,
                                                                    const uint8_t* __restrict__ kvalid, const u32* __restrict__ tslot,
                                                                    const uint8_t* __restrict__ full, const uint8_t* __restrict__ wide,
                                                                    u32 table_slots, u32* __restrict__ gacc,
                                                                    uint8_t* __restrict__ acc, int j0, int j1) {
    struct alignas(16) q4 { u32 x, y, z, w; };
    __shared__ q4 rows[2][SBV_GTAB_PER_WINDOW * 4];          // 2 x 128 entries x 64 B
    __shared__ u32 wave_slot[SBV_VERIFY_BLOCK / 64];
    __shared__ u32 blk_top;
    const u32 lanes = g.counters[1];
    const u32 per = ((lanes + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK + 7) >> 3;
    const u32 local = blockIdx.x >> 3;
    if (local >= per) return;                                               // whole workgroup: no barrier has been reached
    const u32 tid = threadIdx.x;
    const u32 L = ((blockIdx.x & 7u) * per + local) * SBV_VERIFY_BLOCK + tid;
    const bool in = L < lanes;
    const u32 t = in ? g.grp_idx[L] : 0u;
    const u32 grp = in ? g.grp_of[L] : 0xFFFFFFFFu;
    const bool known = in && grp < group_count(g);
    const u32 ts = known ? tslot[grp] : SBV_GROUP_NONE;
    const bool dead = !(ts < table_slots) || kvalid[ts < table_slots ? ts : 0u] == 0;
    const int cls = q_wave_class(dead, !dead && wide[known ? grp : 0u] != 0, !dead && full[known ? grp : 0u] != 0);
    // what this wavefront wants from the workgroup: 0xFFFFFFFE = nothing (every lane dead), a slot = its live lanes all use that
    // slot's full table, SBV_GROUP_NONE = anything else (two keys, a rows-only or a wide wavefront)
    u32 want = 0xFFFFFFFEu;
    if (cls != SBV_Q_NONE) {
        const unsigned long long live = __ballot(!dead);
        const u32 first = (u32)__shfl((int)ts, __ffsll((long long)live) - 1, 64);
        want = cls == SBV_Q_FULL && __all(dead || ts == first) ? first : SBV_GROUP_NONE;
    }
    if ((tid & 63u) == 0) wave_slot[tid >> 6] = want;
    if (tid == 0) blk_top = 0;
    __syncthreads();
    u32 slot = 0xFFFFFFFEu;
    bool uniform = true;
    SBV_UNROLL
    for (int w = 0; w < SBV_VERIFY_BLOCK / 64; ++w) {
        const u32 v = wave_slot[w];
        if (v == 0xFFFFFFFEu) continue;
        if (v == SBV_GROUP_NONE || (slot != 0xFFFFFFFEu && v != slot)) uniform = false;
        slot = v;
    }
    if (!uniform || slot == 0xFFFFFFFEu) {                                  // the ordinary path, wavefront by wavefront (no barrier below)
        if (!in) return;
        if (cls == SBV_Q_NONE) { if (LAST) acc[t] = 0; return; }
        if (cls != SBV_Q_FULL) return;
        const bool v = qphase29_lane_sorted<false>(s, t, L, ts, table_slots, ktab, kvalid, gacc, j0, j1, LAST);
        if (LAST) acc[t] = v ? 1 : 0;
        return;
    }
    // ---- one key for the whole workgroup: rows through LDS ------------------------------------------------------------------------
    const bool livel = !dead;                                               // dead and out-of-range lanes walk along (barriers) and add nothing
    const apt* qtab = ktab + (size_t)slot * (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW);
    u256 u2in;
    SBV_UNROLL
    for (int w = 0; w < 8; ++w) u2in.v[w] = 0;
    if (livel) rec_load256(u2in, s.rec, t, SBV_REC_U2);
    const bool flip = (u2in.v[7] >> 31) != 0;
    u256 u2, nmu;
    (void)sub256(nmu, sc_n(), u2in);
    select256(u2, flip, nmu, u2in);
    u288 k;
    gcomb_recode(k, u2, 8, SBV_GTAB_WINDOWS);
    u32 idx; bool neg, skip;
    if (j1 == SBV_GTAB_WINDOWS) {                                           // the carry window only if a lane of the WORKGROUP carries
        gcomb_digit(k, 8, SBV_GTAB_WINDOWS - 1, idx, neg, skip);
        if (wave_any(livel && !skip) && (tid & 63u) == 0) atomicOr(&blk_top, 1u);
    }
    xyzz R;
    pt29_set_inf(R);
    if (livel) gacc29_load(R, gacc, s.cap, L);
    const q4* grow = reinterpret_cast<const q4*>(qtab + (size_t)j0 * SBV_GTAB_PER_WINDOW);
    q4 pa = grow[2 * tid], pb = grow[2 * tid + 1];
    rows[j0 & 1][2 * tid] = pa; rows[j0 & 1][2 * tid + 1] = pb;
    __syncthreads();                                                        // row j0 and blk_top are in place
    if (j1 == SBV_GTAB_WINDOWS && blk_top == 0) j1 = SBV_GTAB_WINDOWS - 1;
    SBV_NOUNROLL
    for (int j = j0; j < j1; ++j) {
        const bool more = j + 1 < j1;
        if (more) {
            const q4* nrow = reinterpret_cast<const q4*>(qtab + (size_t)(j + 1) * SBV_GTAB_PER_WINDOW);
            pa = nrow[2 * tid]; pb = nrow[2 * tid + 1];
        }
        gcomb_digit(k, 8, j, idx, neg, skip);
        if (livel && !skip) {
            const q4* e = &rows[j & 1][idx * 4];
            raw_apt cur;
            const q4 a = e[0], b = e[1], c = e[2], d = e[3];
            cur.w[0] = a.x; cur.w[1] = a.y; cur.w[2] = a.z; cur.w[3] = a.w; cur.w[4] = b.x; cur.w[5] = b.y; cur.w[6] = b.z; cur.w[7] = b.w;
            cur.w[8] = c.x; cur.w[9] = c.y; cur.w[10] = c.z; cur.w[11] = c.w; cur.w[12] = d.x; cur.w[13] = d.y; cur.w[14] = d.z; cur.w[15] = d.w;
            apt29 q;
            raw_apt_unpack(q, cur);
            pt29_madd(R, q, neg != flip);
        }
        if (more) { rows[(j + 1) & 1][2 * tid] = pa; rows[(j + 1) & 1][2 * tid + 1] = pb; }
        __syncthreads();
    }
    if (!in) return;
    if (dead) { if (LAST) acc[t] = 0; return; }
    if (!LAST) {
        size_t Ls = L;
        asm volatile("" : "+v"(Ls));
        gacc29_store(gacc, s.cap, Ls, R);
        return;
    }
    u256 r;
    rec_load256(r, s.rec, t, SBV_REC_R);
    acc[t] = s.rec[(size_t)t * SBV_REC_WORDS + SBV_REC_OK] != 0 && pt29_rx_matches(R, r) ? 1 : 0;
}

// compaction order (SBV_GROUP_SORT=0): no wide pass (the class kernel leaves wide[] empty)
template <int MODE>
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_COMB29_WAVES) void k_verify_keyed_q_list(Scratch s, GroupState g, const apt* __restrict__ ktab,
                                                                    const uint8_t* __restrict__ kvalid, const u32* __restrict__ tslot,
                                                                    const uint8_t* __restrict__ full, u32 table_slots, u32* __restrict__ gacc,
                                                                    uint8_t* __restrict__ acc, int j0, int j1, int last) {
    if (MODE == SBV_Q_NARROW && g.counters[5] >= group_count(g)) return;
    const u32 L = blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    if (L >= g.counters[1]) return;
    const u32 t = g.grp_idx[L];
    const u32 grp = g.slots[t];                                        // group of this tuple -> its table slot (cache or per-batch area)
    const bool known = grp < group_count(g);
    const u32 ts = known ? tslot[grp] : SBV_GROUP_NONE;
    const bool dead = !(ts < table_slots) || kvalid[ts < table_slots ? ts : 0u] == 0;
    const int cls = q_wave_class(dead, false, !dead && full[grp] != 0);
    if (cls == SBV_Q_NONE) { if (MODE == SBV_Q_FULL && last) acc[t] = 0; return; }
    if (cls != MODE) return;
    if (MODE == SBV_Q_NARROW) {
        const unsigned long long am = __ballot(true);
        if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)am) - 1u) atomicAdd(&g.counters[7], (u32)__popcll(am));
    }
    const bool v = qphase29_lane<MODE == SBV_Q_NARROW>(s, t, ts, table_slots, ktab, kvalid, gacc, j0, j1, last != 0);
    if (last) acc[t] = v ? 1 : 0;
}

// Latency form of the key-sorted step (GroupSync::coop_max: batches up to 2^15 tuples; measured in round 4,
// profiles/r04/ab_coop_r04a.jsonl: warm 2^10 0.30 -> 0.19 ms, 2^12 0.34 -> 0.22, 2^14 0.37 -> 0.27, 2^15 0.39 -> 0.33): SBV_COOP_LANES lanes per
// grouped tuple, every lane sums every SBV_COOP_LANES-th of the 13 + 33 comb terms of u1 * G + u2 * Q from the comb of G and
// the key's table (p256_comb29.h: keyed29_partial_lane, the registered-key latency kernel's lane) and the partial sums meet
// in a butterfly of exact XYZZ additions: one launch ~10 additions deep instead of the G phase and two Q launches (45 deep).
// For warm batches of a few thousand tuples, where every kernel of the step runs at the latency of one lane.
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, 2) void k_group_coop(Scratch s, GroupState g, const apt* __restrict__ ktab,
                                                                    const uint8_t* __restrict__ kvalid, const u32* __restrict__ tslot,
                                                                    u32 table_slots, gcomb g16r, uint8_t* __restrict__ acc) {
    const size_t lane_g = (size_t)blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    const size_t L = lane_g / SBV_COOP_LANES;
    const int sub = (int)(lane_g % SBV_COOP_LANES);
    const bool active = L < g.counters[1];
    xyzz R;
    pt29_set_inf(R);
    bool ok = false;
    u32 t = 0;
    if (active) {
        t = g.grp_idx[L];
        const u32 grp = g.grp_of[L];
        u32 slot = grp < group_count(g) ? tslot[grp] : SBV_GROUP_NONE;
        ok = slot < table_slots;
        if (!ok) slot = 0;
        ok = ok && kvalid[slot] != 0 && s.rec[(size_t)t * SBV_REC_WORDS + SBV_REC_OK] != 0;
        u256 u1, u2;
        rec_load256(u1, s.rec, t, SBV_REC_U1);
        rec_load256(u2, s.rec, t, SBV_REC_U2);
        keyed29_partial_lane(R, u1, u2, ktab + (size_t)slot * (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW), g16r, sub);
    }
    SBV_NOUNROLL
    for (int off = SBV_COOP_LANES / 2; off >= 1; off >>= 1) {      // after log2(lanes) exchanges every lane of the group holds the whole sum
        xyzz P;
        SBV_UNROLL
        for (int l = 0; l < 9; ++l) {
            P.X.v[l] = __shfl_xor(R.X.v[l], off, 64);
            P.Y.v[l] = __shfl_xor(R.Y.v[l], off, 64);
            P.ZZ.v[l] = __shfl_xor(R.ZZ.v[l], off, 64);
            P.ZZZ.v[l] = __shfl_xor(R.ZZZ.v[l], off, 64);
        }
        P.inf = __shfl_xor(R.inf ? 1 : 0, off, 64) != 0;
        pt29_add(R, P);
    }
    if (active && sub == 0) {
        u256 r;
        rec_load256(r, s.rec, t, SBV_REC_R);
        acc[t] = ok && pt29_rx_matches(R, r) ? 1 : 0;
    }
}

// Enqueue stage B with in-step grouping.  Stage A (k_p256_prep) is enqueued here too, on `stream`.
// Three streams (a process gets 4 hardware queues by default; a further stream aliases one of them and serialises the step,
// measured twice).  Only the G phase and the Q phase are throughput work; everything else is a low-occupancy, latency-bound
// chain (four lanes per key / a few lanes per window / the rare ungrouped tuples), so the chains run beside each other and
// beside the throughput kernels:
//
//   stream: prep | wait(split) { generic stage B over the ungrouped list, G phase } wait(tables c) Q-phase chunk c ... pack
//   side_a: insert assign cache | chain chunk 0 | chain chunk 1 | ... | wait(G phase) wide pass
//   side_b:        wait(assign) classify keycheck sort | wait(chain c) rows + fill of chunk c   (odd chunks: side_t, if given) | rows-only pass | promotions
//
// Forms that were built, measured and rejected in rounds 2-4 are no longer in the library (DESIGN.md section 7 keeps the
// numbers): stage A + G phase in slices, table pieces finer than the Q chunks, an uneven first chunk, one lane per table entry,
// fill rows split over lanes, an own stream for the one-lane kernel.
hipError_t launch_p256_verify_grouped(const uint8_t* d_tuples, const Scratch& s_in, size_t n, const GroupBuffers& b,
                                      u32* d_qtab, const apt* d_g16, const gcomb& d_g16r, uint8_t* d_bitmap, hipStream_t stream,
                                      const GroupSync& y, hipEvent_t after_prep, hipEvent_t* prof, int* prof_pairs) {
    if (n == 0) return hipSuccess;
    GroupState g;
    g.ht = b.ht; g.ht_mask = b.ht_mask; g.rep = b.rep; g.cnt = b.cnt; g.slot_of = b.slot_of; g.group_rep = b.group_rep;
    g.counters = b.counters; g.grp_idx = b.grp_idx; g.ung_idx = b.ung_idx; g.slots = b.slots;
    g.max_groups = b.max_groups; g.seed = b.seed;
    g.gcount = b.gcount; g.gcursor = b.gcount ? b.gcount + b.max_groups : nullptr; g.grp_of = b.grp_of; g.ung_cand = b.ung_cand;
    // key-sorted grouped list: needs the per-tuple records of stage A; one LDS word per group while the histogram fits (<= 16 384
    // groups), plain global atomics beyond (the kernels decide: group_kernels_common.h)
    const size_t sort_lds = (size_t)(b.max_groups > SBV_SORT_LDS_GROUPS ? SBV_SORT_LDS_GROUPS : b.max_groups) * sizeof(u32);
    g.sorted = y.sorted && s_in.rec && b.gcount && b.grp_of && b.ung_cand ? 1u : 0u;
    // Stage A writes EITHER the per-tuple records (key-sorted step: every reader takes them) OR the limb-major planes
    Scratch s = s_in;
    if (!g.sorted) s.rec = nullptr;
    if (b.sample_shift >= 0) group_set_sampling(g, b.min_count, (u32)b.sample_shift);
    else group_set_threshold(g, b.min_count);
    const int chunks = y.chunks < 1 ? 1 : (y.chunks > SBV_GROUP_MAX_CHUNKS ? SBV_GROUP_MAX_CHUNKS : y.chunks);
    const bool coop = g.sorted && y.coop_max && n <= y.coop_max;      // k_group_coop instead of the G phase and the Q launches
    // table classes (p256_group.h): the coop launch reads any entry of a row, so its batches (<= 2^15 tuples) fill every table
    const u32 full_min = coop ? 0u : b.full_min;
    const u32 table_slots = b.kc.cap + b.max_groups;
    // hot keys (p256_group.h): only with the cache on, a pool to promote into and the key-sorted list
    const bool hot_on = b.wtab && b.kwide && b.kc.enabled && g.sorted;
    const HotKeys hk = {b.wtab, hot_on ? b.kwide : nullptr, b.khits, b.hot, b.plist, b.kc.cap, b.wide_cap, b.promote_min, b.wowner, b.elist};
    const widekeys wk = hot_on ? widekeys_make(b.wtab, b.kwide, SBV_HOT_BITS) : widekeys_none();
    static const bool q_lds = [] { const char* v = getenv("SBV_QPHASE_LDS"); return v && v[0] == '1'; }();      // the LDS-staged form of the chunks' launches (opt-in)
    hipError_t e;
#define SBV_TRY(x) do { if ((e = (x)) != hipSuccess) return e; } while (0)
    // The side streams may not touch the group buffers before everything already enqueued on `stream` (the
    // previous batch's readers of those buffers) has run; ev_fork was recorded by the caller BEFORE stage A.
    SBV_TRY(hipStreamWaitEvent(y.side_a, y.ev_fork, 0));
    // ... nor before the previous batch's tail on side_b (table marks, promotion select: the readers of tslot and of the class bytes) is
    // through (an event never recorded yet waits for nothing)
    SBV_TRY(hipStreamWaitEvent(y.side_a, y.ev_promoted, 0));
    SBV_TRY(hipMemsetAsync(b.ht, 0, ((size_t)b.ht_mask + 1) * sizeof(u32), y.side_a));
    SBV_TRY(hipMemsetAsync(b.cnt, 0, n * sizeof(u32), y.side_a));
    SBV_TRY(hipMemsetAsync(b.counters, 0, SBV_GROUP_COUNTERS * sizeof(u32), y.side_a));
    if (g.sorted) SBV_TRY(hipMemsetAsync(b.gcount, 0, (size_t)b.max_groups * sizeof(u32), y.side_a));
    const unsigned gn = (unsigned)((n + 255) / 256);
    const unsigned gv = (unsigned)((n + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK);
    hipLaunchKernelGGL(k_group_insert, dim3(gn), dim3(256), 0, y.side_a, d_tuples, n, g);
    hipLaunchKernelGGL(k_group_assign, dim3(gn), dim3(256), 0, y.side_a, d_tuples, n, g, b.kc);
    SBV_TRY(hipEventRecord(y.ev_assign, y.side_a));
    hipLaunchKernelGGL((k_key_cache_lookup_t<160, 96, 16>), dim3((b.max_groups + 63) / 64), dim3(64), 0, y.side_a, d_tuples, g, b.kc, b.tslot, b.cold);
    hipLaunchKernelGGL((k_key_cache_insert_t<160, 96, 16>), dim3((b.max_groups + 63) / 64), dim3(64), 0, y.side_a, d_tuples, g, b.kc, b.tslot);
    SBV_TRY(hipEventRecord(y.ev_cache, y.side_a));        // tslot / cold are final: the table classes (side_b) need them
    // stage A
    const size_t pbt = prep_block_tuples(n);
    const unsigned pblocks = (unsigned)((n + pbt - 1) / pbt);
    SBV_TRY(launch_p256_prep_blocks(d_tuples, n, s, stream, 0, pblocks));
    // side_b: split
    SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_assign, 0));
    if (hot_on) SBV_TRY(hipMemsetAsync(b.hot + 1, 0, 3 * sizeof(u32), y.side_b));        // promotions and wide-pass lanes of THIS batch: behind the previous batch's builder, which reads hot[1]
    if (!g.sorted) hipLaunchKernelGGL(k_group_split, dim3(gn), dim3(256), 0, y.side_b, d_tuples, n, g, b.acc);
    if (g.sorted) {             // classify, check the ungrouped candidates' keys, counting sort of the grouped tuples by key;
                                // ev_split then also stands for "the sorted list is final"
        hipLaunchKernelGGL(k_group_classify, dim3(gn), dim3(256), 0, y.side_b, n, g, b.ung_cand, b.counters + 4);
        hipLaunchKernelGGL(k_group_keycheck, dim3(gn), dim3(256), 0, y.side_b, d_tuples, g, b.acc);
        const unsigned tiles = (unsigned)((n + SBV_SORT_TILE - 1) / SBV_SORT_TILE);
        hipLaunchKernelGGL(k_group_sort_count, dim3(tiles), dim3(1024), sort_lds, y.side_b, n, g);
        hipLaunchKernelGGL(k_group_sort_scan, dim3(1), dim3(1024), 0, y.side_b, g);
    }
    if (g.sorted) {
        const unsigned tiles = (unsigned)((n + SBV_SORT_TILE - 1) / SBV_SORT_TILE);
        hipLaunchKernelGGL(k_group_sort_scatter, dim3(tiles), dim3(1024), sort_lds, y.side_b, n, g);
    }
    SBV_TRY(hipEventRecord(y.ev_split, y.side_b));
    // table classes: after the exact counts (gcount survives the scan; the scatter moves gcursor only) and after the cache assigned the
    // slots; behind the scatter, so that the G phase (which waits for the sorted list, not for the classes) starts a launch earlier
    SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_cache, 0));
    hipLaunchKernelGGL(k_group_table_class, dim3(64), dim3(256), 0, y.side_b, g, b.tslot, b.cold, b.kfull, table_slots, full_min, b.full, b.needfill, hk, b.wide);
    SBV_TRY(hipEventRecord(y.ev_class, y.side_b));
    // The generic stage B over the ungrouped list (keys that repeat too rarely for a table: a 2.3 ms chain per lane, so it starts
    // as early as possible and runs beside the throughput work) and the G phase.  group_count() is final after the assignment;
    // the ungrouped list (and the sorted list) after the split.
    SBV_TRY(hipStreamWaitEvent(stream, y.ev_assign, 0));
    SBV_TRY(hipStreamWaitEvent(stream, y.ev_split, 0));
    if (after_prep) SBV_TRY(hipEventRecord(after_prep, stream));
    if (coop) {                  // the ungrouped list only: the coop launch below does the G phase's job too
        hipLaunchKernelGGL(k_gphase_generic, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, stream, d_tuples, s, n, g, d_qtab, d_g16, d_g16r, b.gacc, b.acc, gv, (size_t)0, n);
    } else if (g.sorted && y.gsplit_min && n >= y.gsplit_min) {
        hipLaunchKernelGGL(k_gphase_generic, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, stream, d_tuples, s, n, g, d_qtab, d_g16, d_g16r, b.gacc, b.acc, gv, (size_t)0, n);
        hipLaunchKernelGGL(k_gphase_sorted, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, d_g16r, b.gacc);
    } else {
        hipLaunchKernelGGL(k_gphase_generic, dim3(gv + gv), dim3(SBV_VERIFY_BLOCK), 0, stream, d_tuples, s, n, g, d_qtab, d_g16, d_g16r, b.gacc, b.acc, gv, (size_t)0, n);
    }
    SBV_TRY(hipEventRecord(y.ev_generic, stream));         // the G phase is enqueued: the accumulators of the rows-only and the wide pass are final behind this event
    // Chunks of windows: the chain on side_a, rows + fill on side_b (odd chunks on side_t), the Q phase on stream.
    for (int c = 0; c < chunks; ++c) {
        const int j_first = SBV_GTAB_WINDOWS * c / chunks, j_end = SBV_GTAB_WINDOWS * (c + 1) / chunks;   // [j_first, j_end)
        const int j_count = j_end - j_first;
        const bool on_t = y.tstreams > 1 && y.side_t && (c & 1);
        hipStream_t tb = on_t ? y.side_t : y.side_b;     // rows + fill of this chunk
        u32* ttmp = b.tmp + (on_t ? SBV_TABLE_TMP_WORDS : 0);          // per-lane scratch of this stream's table kernels (never two launches of one stream at a time)
        auto bounded = [](size_t lanes) { const size_t blocks = (lanes + 63) / 64; return (unsigned)(blocks < SBV_TABLE_GRID_BLOCKS ? (blocks ? blocks : 1) : SBV_TABLE_GRID_BLOCKS); };
        hipLaunchKernelGGL(k_keytab29_chain, dim3(bounded((size_t)b.max_groups * 4)), dim3(64), 0, y.side_a, d_tuples, g, b.jstate, b.bases,
                           b.kvalid, b.tslot, b.cold, j_first, j_end - 1, 0x11u);
        SBV_TRY(hipEventRecord(y.ev_bases[c], y.side_a));
        if (hot_on && !coop && c + 1 == chunks) {
            // The wide pass: the wavefronts whose keys all own a wide comb — 17 additions from the 16-bit combs, no table of this batch
            // needed.  On side_a behind the LAST chain (in front of them it would hold up the tables of the batch's cold keys; on
            // `stream` it ran before the chunks' launches instead of beside them: with a quarter of the lanes wide it is a 0.87 ms
            // latency chain at low occupancy, timeline_hot_4096_r05q.txt), as soon as the G phase is done.
            SBV_TRY(hipStreamWaitEvent(y.side_a, y.ev_generic, 0));
            SBV_TRY(hipStreamWaitEvent(y.side_a, y.ev_class, 0));
            hipLaunchKernelGGL((k_verify_keyed_q<SBV_Q_WIDE, true>), dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, y.side_a, s, g, b.ktab, b.kvalid, b.tslot, b.full, b.wide, wk, b.hot + 2,
                               table_slots, b.gacc, b.acc, 0, SBV_GTAB_WINDOWS);
            SBV_TRY(hipEventRecord(y.ev_wide, y.side_a));
        }
        SBV_TRY(hipStreamWaitEvent(tb, y.ev_bases[c], 0));
        if (on_t) SBV_TRY(hipStreamWaitEvent(tb, y.ev_class, 0));      // side_b has it in stream order
        hipLaunchKernelGGL(k_keytab29_rows, dim3(bounded((size_t)b.max_groups * j_count * 2)), dim3(64), 0, tb, g, b.bases, ttmp, b.ktab, b.ntab, b.tslot, b.cold, b.kvalid, j_first, j_count);
        // one inversion per window (default since round 6: Q launch 926-939 -> 872-883 us beside it, cold step -0.5 %, profiles/r06/ab_fill_*_r06bd.jsonl); SBV_FILL_SHARED=0: one per lane
        static const bool fill_shared = [] { const char* e = getenv("SBV_FILL_SHARED"); return !e || atoi(e) != 0; }();
        if (fill_shared) hipLaunchKernelGGL(k_keytab29_fill_shared, dim3((bounded((size_t)b.max_groups * j_count * 8) + (SBV_FILL_SHARED_BLOCK / 64) - 1u) / (SBV_FILL_SHARED_BLOCK / 64)), dim3(SBV_FILL_SHARED_BLOCK), 0, tb, g, ttmp, b.ktab, b.tslot,
                                            b.needfill, b.kvalid, j_first, j_count);
        else hipLaunchKernelGGL(k_keytab29_fill_sym, dim3(bounded((size_t)b.max_groups * j_count * 8)), dim3(64), 0, tb, g, ttmp, b.ktab, b.tslot, b.needfill, b.kvalid, j_first, j_count);
        SBV_TRY(hipEventRecord(y.ev_tables[c], tb));
        SBV_TRY(hipStreamWaitEvent(stream, y.ev_tables[c], 0));
        const bool last = c + 1 == chunks;
        if (coop) {              // every table first, then the one launch
            if (!last) continue;
            if (prof) SBV_TRY(hipEventRecord(prof[0], stream));
            const size_t lanes = n * SBV_COOP_LANES;
            hipLaunchKernelGGL(k_group_coop, dim3((unsigned)((lanes + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK)), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, b.ktab,
                               b.kvalid, b.tslot, table_slots, d_g16r, b.acc);
            if (prof) SBV_TRY(hipEventRecord(prof[1], stream));
            continue;
        }
        if (last) {
            // The wavefronts with a key that has rows only: all 33 windows, two additions per window — a ~60-addition chain per lane.
            // On side_b (idle once the last fill is done), BESIDE the last chunk's launch instead of behind it: needs the G phase and
            // the rows of every chunk (side_b has its own in stream order and waits for the other table stream's).
            SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_generic, 0));
            for (int cc = 0; cc < chunks; ++cc) SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_tables[cc], 0));
            if (g.sorted) hipLaunchKernelGGL((k_verify_keyed_q<SBV_Q_NARROW, true>), dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, y.side_b, s, g, b.ntab, b.kvalid, b.tslot, b.full, b.wide, wk, b.hot + 2,
                                             table_slots, b.gacc, b.acc, 0, SBV_GTAB_WINDOWS);
            else hipLaunchKernelGGL(k_verify_keyed_q_list<SBV_Q_NARROW>, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, y.side_b, s, g, b.ntab, b.kvalid, b.tslot, b.full,
                                    table_slots, b.gacc, b.acc, 0, SBV_GTAB_WINDOWS, 1);
            SBV_TRY(hipEventRecord(y.ev_narrow, y.side_b));
        }
        if (prof) SBV_TRY(hipEventRecord(prof[2 * c], stream));
        if (!g.sorted) hipLaunchKernelGGL(k_verify_keyed_q_list<SBV_Q_FULL>, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, b.ktab, b.kvalid, b.tslot, b.full,
                                          table_slots, b.gacc, b.acc, j_first, j_end, last ? 1 : 0);
        else if (q_lds && last) hipLaunchKernelGGL((k_verify_keyed_q_lds<true>), dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, b.ktab, b.kvalid, b.tslot, b.full, b.wide,
                                                   table_slots, b.gacc, b.acc, j_first, j_end);
        else if (q_lds) hipLaunchKernelGGL((k_verify_keyed_q_lds<false>), dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, b.ktab, b.kvalid, b.tslot, b.full, b.wide,
                                           table_slots, b.gacc, b.acc, j_first, j_end);
        else if (last) hipLaunchKernelGGL((k_verify_keyed_q<SBV_Q_FULL, true>), dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, b.ktab, b.kvalid, b.tslot, b.full, b.wide, wk, b.hot + 2,
                                          table_slots, b.gacc, b.acc, j_first, j_end);
        else hipLaunchKernelGGL((k_verify_keyed_q<SBV_Q_FULL, false>), dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, b.ktab, b.kvalid, b.tslot, b.full, b.wide, wk, b.hot + 2,
                                table_slots, b.gacc, b.acc, j_first, j_end);
        if (prof) SBV_TRY(hipEventRecord(prof[2 * c + 1], stream));
    }
    if (!coop) SBV_TRY(hipStreamWaitEvent(stream, y.ev_narrow, 0));
    if (hot_on && !coop) SBV_TRY(hipStreamWaitEvent(stream, y.ev_wide, 0));
    // `stream` has the verdict bytes: pack them; everything else is the batch's tail and runs on side_b, off the caller's path
    SBV_TRY(hipEventRecord(y.ev_promote, stream));
    hipLaunchKernelGGL(k_pack_bitmap, dim3((unsigned)(((n + 7) / 8 + 255) / 256)), dim3(256), 0, stream, b.acc, n, d_bitmap);
    SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_promote, 0));
    hipLaunchKernelGGL(k_group_table_mark, dim3(64), dim3(256), 0, y.side_b, g, b.tslot, b.cold, b.full, b.needfill, table_slots, b.kfull);
    if (hot_on) {
        // the life cycle (round 6): the clock sweep every SBV_HOT_DECAY_EVERY-th batch
        if (b.hot_tick % SBV_HOT_DECAY_EVERY == SBV_HOT_DECAY_EVERY - 1) hipLaunchKernelGGL(k_hot_decay, dim3((b.kc.cap + 255) / 256), dim3(256), 0, y.side_b, hk);
        // promotions (p256_group.h: hot keys): which slots (the evictions when the pool is full), and their base points (tiny launches
        // that read tslot and the tables)
        hipLaunchKernelGGL(k_promote_select, dim3(64), dim3(256), 0, y.side_b, g, b.tslot, b.kvalid, hk);
        hipLaunchKernelGGL(k_promote_evict, dim3(1), dim3(1024), 0, y.side_b, hk);
        const widebuild wb = widebuild_make(SBV_HOT_BITS);
        hipLaunchKernelGGL(k_promote_bases, dim3((SBV_PROMOTE_MAX * 2 * (u32)wb.windows + 63) / 64), dim3(64), 0, y.side_b, b.plist, b.hot, b.ktab, b.pbases);
    }
    // ev_promoted: the tail no longer reads tslot, the class bytes or the per-batch tables — the next batch's side_a (which rewrites them)
    // waits for it; the builder and the publication follow on side_b, in front of the next batch's own side_b work (its table classes
    // read kwide, its hot[] reset sits there too) and of its wide pass (behind the G phase, which waits for that side_b work)
    SBV_TRY(hipEventRecord(y.ev_promoted, y.side_b));
    if (hot_on) {
        const widebuild wb = widebuild_make(SBV_HOT_BITS);
        const size_t stride = gcomb_entries(SBV_HOT_BITS);
        hipLaunchKernelGGL(k_promote_chains, dim3((SBV_PROMOTE_MAX * (u32)wb.windows * 2u + 63) / 64), dim3(64), 0, y.side_b, b.pbases, b.plist, b.hot, wb, stride, b.ptmp, b.wtab);
        const size_t fill_lanes = (size_t)SBV_PROMOTE_MAX * wb.windows * (wb.giants - 1) * widebuild_fill_chunks(wb);
        hipLaunchKernelGGL(k_promote_fill, dim3((unsigned)((fill_lanes + 255) / 256)), dim3(256), 0, y.side_b, b.plist, b.hot, wb, stride, b.wtab);
        hipLaunchKernelGGL(k_promote_publish, dim3(1), dim3(64), 0, y.side_b, b.plist, b.hot, b.kwide, b.wowner);
    }
#undef SBV_TRY
    if (prof && prof_pairs) *prof_pairs = coop ? 1 : chunks;
    return hipGetLastError();
}

}  // namespace sbv
