// ed25519_group_kernels.hip — the grouped step of the Ed25519 variant (ed25519_group.h), same stream layout
// as the P-256 one (p256_group_kernels.hip):
//
//   stream: [S]B for every tuple ........ wait(split, tables c) Q-phase chunk c ... wait(generic) pack
//   side_a: insert assign | bases chunk 0 | bases chunk 1 | ... | wait(split) one-lane kernel over the ungrouped list
//   side_b:        wait(assign) split | wait(bases c) windows chunk c ...
//
// No stage A here (Ed25519 has no scalar inversion), so the G phase starts at once.  Unlike P-256 the ungrouped
// list is never empty on the headline-shaped batch (half of all corrupted keys still decompress), so its 2.5 ms
// chain runs on side_a where it gates only the final pack.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ed25519_group.h"
#include "ed25519_kernels.h"
#include "group_kernels_common.h"
#include "p256_kernels.h"

namespace sbv {

#ifndef SBV_ED_QPHASE_WAVES
#define SBV_ED_QPHASE_WAVES 3
#endif
#ifndef SBV_ED_GROUP_WAVES
#define SBV_ED_GROUP_WAVES 2      // waves/SIMD the comb kernels are compiled for: 234 VGPRs and no scratch at 2; 168 + 268 B of spills at 3, same speed (profiles/r02/ed25519_ab_r02.txt)
#endif

// The ungrouped list is a few hundred wavefronts of one long serial chain each (key check, then the one-lane kernel) beside the
// throughput kernels; once a scheme's combs are cached it is the step's critical path.  SBV_ED_CHAIN_PRIO > 0 raises those wavefronts'
// priority at their SIMD's arbiter (s_setprio).
#ifndef SBV_ED_CHAIN_PRIO
#define SBV_ED_CHAIN_PRIO 0
#endif
static __device__ __forceinline__ void ed_chain_prio() {
#if SBV_ED_CHAIN_PRIO > 0
    __builtin_amdgcn_s_setprio(SBV_ED_CHAIN_PRIO);
#endif
}

__global__ __launch_bounds__(256) void k_ed_group_insert(const uint8_t* __restrict__ tuples, size_t n, GroupState g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    (void)i;
    group_insert_block<128, 64, 8>(tuples, n, g);
}

// Same result as ed_group_split_lane (compaction: group_split_emit)
__global__ __launch_bounds__(256) void k_ed_group_split(size_t n, GroupState g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n;
    u32 s = SBV_GROUP_NONE;
    if (active) s = g.slot_of[g.rep[i]];
    group_split_emit(i, s, active && s == SBV_GROUP_NONE, active && s != SBV_GROUP_NONE, false, g);
}

// only the groups whose tables are built in this batch (cold: not found in the key-table cache).  Four lanes per key (ed25519_group.h:
// edchain_run); a quad lives or exits as a whole.
struct edchain_quad_dev {
    static const int N = 1;
    ept s[1];
    int r;
    __device__ __forceinline__ int role(int) const { return r; }
    __device__ __forceinline__ void bcast(fe25 out[1], const fe25 in[1], int src) const {
        SBV_UNROLL
        for (int l = 0; l < 10; ++l) {
            const int v = in[0].v[l];
            out[0].v[l] = src == 0 ? __builtin_amdgcn_mov_dpp(v, 0x00, 0xF, 0xF, true)
                        : src == 1 ? __builtin_amdgcn_mov_dpp(v, 0x55, 0xF, 0xF, true)
                        : src == 2 ? __builtin_amdgcn_mov_dpp(v, 0xAA, 0xF, 0xF, true)
                                   : __builtin_amdgcn_mov_dpp(v, 0xFF, 0xF, 0xF, true);
        }
    }
};
__global__ __launch_bounds__(64) void k_ed_keytab_bases(const uint8_t* __restrict__ tuples, GroupState g, u32* __restrict__ jbases,
                                                        uint8_t* __restrict__ valid, const u32* __restrict__ tslot, const uint8_t* __restrict__ cold,
                                                        int j_first, int j_last) {
    const u32 lane = blockIdx.x * 64 + threadIdx.x;
    const u32 k = lane >> 2;
    if (k >= group_count(g) || !cold[k]) return;
    edchain_quad_dev q;
    q.r = (int)(lane & 3u);
    edchain_run(q, tuples, k, g, jbases, valid + tslot[k], j_first, j_last);
}

// lanes = groups x j_count x parts
__global__ __launch_bounds__(64) void k_ed_keytab_window(GroupState g, const u32* __restrict__ jbases, u32* __restrict__ tmp,
                                                         aniels* __restrict__ ktab, const u32* __restrict__ tslot, const uint8_t* __restrict__ cold,
                                                         int j_first, int j_count, int parts) {
    const u32 lane = blockIdx.x * 64 + threadIdx.x;
    const u32 part = lane % (u32)parts;
    const u32 kw = lane / (u32)parts;
    const u32 key = kw / (u32)j_count, j = (u32)j_first + kw % (u32)j_count;
    if (key >= group_count(g) || !cold[key]) return;
    const size_t w = (size_t)key * SBV_ED_KEY_WINDOWS + j;
    ed_keytab_window_lane(jbases + w * SBV_ED_JBASE_DWORDS, (int)part, parts,
                          tmp + w * (size_t)(SBV_ED_KEY_PER_WINDOW * SBV_ED_WINDOW_TMP_WORDS) + (size_t)part * (SBV_ED_KEY_PER_WINDOW / parts) * SBV_ED_WINDOW_TMP_WORDS,
                          ktab + ((size_t)tslot[key] * SBV_ED_KEY_WINDOWS + j) * SBV_ED_KEY_PER_WINDOW);
}

// The one-lane kernel (ed25519_verify_lane) over the ungrouped list, at the 3 waves/SIMD budget of the throughput
// kernels it runs beside.  A ~2.5 ms serial chain per lane: it goes on side_a behind the last bases so that it
// gates nothing but the final pack.
#ifndef SBV_ED_ONE_WAVES
#define SBV_ED_ONE_WAVES SBV_ED_GROUP_WAVES
#endif
// ungxy (may be null): the keys as the key check decompressed them (key-sorted step)
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_ED_ONE_WAVES) void k_ed_generic_list(const uint8_t* __restrict__ tuples, GroupState g,
                                                                        u32* __restrict__ qtab, const aniels* __restrict__ btab,
                                                                        uint8_t* __restrict__ acc, const u32* __restrict__ ungxy) {
    const u32 L = blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    if (L >= g.counters[2]) return;
    ed_chain_prio();
    const u32 t = g.ung_idx[L];
    acc[t] = ed25519_verify_lane(EdGlobalTuple{ed_tuple_words(tuples, t)}, qtab + (size_t)L * (SBV_ED_QTAB_ENTRIES * SBV_ED_PT_WORDS), btab,
                                 ungxy ? ungxy + (size_t)L * SBV_ED_UNGXY_WORDS : nullptr) ? 1 : 0;
}

// key-sorted step, pass 2: the ungrouped candidates whose key is a point -> ung_idx (ed25519_group.h: ed_group_keycheck_lane)
__global__ __launch_bounds__(256) void k_ed_keycheck(const uint8_t* __restrict__ tuples, GroupState g, uint8_t* __restrict__ acc, u32* __restrict__ ungxy) {
    const u32 L = blockIdx.x * 256 + threadIdx.x;
    const u32 cands = g.counters[4];
    if (blockIdx.x * 256u >= cands) return;            // whole workgroup idle (uniform: the barriers below are not reached by anyone)
    ed_chain_prio();
    const bool active = L < cands;
    u32 i = 0;
    bool ok = false;
    ept A;
    if (active) {
        i = g.ung_cand[L];
        ok = ed_tuple_key_load(tuples, i, A);
        if (!ok) acc[i] = 0;
    }
    const unsigned long long mr = __ballot(active && !ok);
    if ((threadIdx.x & 63) == 0 && mr) atomicAdd(&g.counters[3], (u32)__popcll(mr));
    const u32 pos = group_compact_pos(active && ok, &g.counters[2]);
    if (active && ok) {
        g.ung_idx[pos] = i;
        if (ungxy) { fe25_store_raw(ungxy + (size_t)pos * SBV_ED_UNGXY_WORDS, A.X); fe25_store_raw(ungxy + (size_t)pos * SBV_ED_UNGXY_WORDS + 10, A.Y); }
    }
}
// The ungrouped list on four lanes per tuple (ed25519_group.h: ed25519_verify_quad): 4 x as many wavefronts, each a chain a third as long
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_ED_GROUP_WAVES) void k_ed_generic_quad(const uint8_t* __restrict__ tuples, GroupState g, const u32* __restrict__ ungxy,
                                                                        u32* qtab, const aniels* __restrict__ btab, uint8_t* __restrict__ acc) {
    const u32 lane = blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    const u32 L = lane >> 2;
    if (L >= g.counters[2]) return;                   // a quad lives or leaves as a whole
    const u32 t = g.ung_idx[L];
    edchain_quad_dev q;
    q.r = (int)(lane & 3u);
    bool v[1];
    ed25519_verify_quad(q, EdGlobalTuple{ed_tuple_words(tuples, t)}, ungxy + (size_t)L * SBV_ED_UNGXY_WORDS, qtab + (size_t)L * (SBV_ED_QTAB_ENTRIES * SBV_ED_PT_WORDS), btab, v);
    if (q.r == 0) acc[t] = v[0] ? 1 : 0;
}

// [S]B for every tuple of the batch
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_ED_GROUP_WAVES) void k_ed_gphase(const uint8_t* __restrict__ tuples, size_t n, edcomb btab,
                                                                  u32* __restrict__ gacc, size_t cap, uint8_t* __restrict__ okb, int tuple_major) {
    const size_t i = (size_t)blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    if (i < n) ed_gphase_lane(tuples, i, btab, gacc, cap, okb, tuple_major != 0);
}

// Round 6: the key-sorted form and the compaction-order form are kernels of their own, LAST is a template parameter (what the P-256 Q
// phase gained from the same split: each form alone needs far fewer registers than the kernel that held them all)
// hot keys: does every (active) lane of this wavefront belong to a group whose cache slot owns a 16-bit comb?
__device__ __forceinline__ bool ed_wave_is_wide(u32 grp, const GroupState& g, const uint8_t* __restrict__ wide) {
    const bool w = grp < group_count(g) && wide[grp] != 0;
    return __ballot(w) == __ballot(true);
}
template <bool LAST>
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_ED_QPHASE_WAVES) void k_ed_qphase(const uint8_t* __restrict__ tuples, GroupState g,
                                                                  const aniels* __restrict__ ktab, const uint8_t* __restrict__ kvalid,
                                                                  const u32* __restrict__ tslot, u32 table_slots,
                                                                  u32* __restrict__ gacc, size_t cap, const uint8_t* __restrict__ okb,
                                                                  uint8_t* __restrict__ acc, int j0, int j1, const uint8_t* __restrict__ wide) {
    // key-sorted list, XCD-aware block order (see k_verify_keyed_q): block b takes logical block (b % 8) * per + b / 8
    const u32 lanes = g.counters[1];
    const u32 per = ((lanes + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK + 7) >> 3;
    const u32 local = blockIdx.x >> 3;
    if (local >= per) return;
    const u32 L = ((blockIdx.x & 7u) * per + local) * SBV_VERIFY_BLOCK + threadIdx.x;
    if (L >= lanes) return;
    const u32 t = g.grp_idx[L];
    const u32 grp = g.grp_of[L];
    if (wide && ed_wave_is_wide(grp, g, wide)) return;                // every lane's key owns a 16-bit comb: k_ed_qphase_wide's wavefront
    const bool v = ed_qphase_lane(tuples, t, grp < group_count(g) ? tslot[grp] : SBV_GROUP_NONE, table_slots, ktab, kvalid, gacc, cap, okb, j0, j1, LAST, true);
    if (LAST) acc[t] = v ? SBV_ED_PENDING : 0;
}
// The wide pass (ed25519_group.h: hot keys): the wavefronts of the key-sorted list whose lanes ALL belong to promoted cache slots —
// [k](-A) in 16 additions from the slot's comb, one launch that needs no table of this batch; it leaves the tuple pending for k_ed_finish
// as the last chunk's launch does for everybody else.  Same block order as k_ed_qphase: the two kernels split the same wavefronts.
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_ED_GROUP_WAVES) void k_ed_qphase_wide(const uint8_t* __restrict__ tuples, GroupState g,
                                                                  const uint8_t* __restrict__ kvalid, const u32* __restrict__ tslot,
                                                                  const uint8_t* __restrict__ wide, const u32* __restrict__ kwide,
                                                                  const uint8_t* __restrict__ wtab, u32* __restrict__ wstat,
                                                                  u32* __restrict__ gacc, const uint8_t* __restrict__ okb, uint8_t* __restrict__ acc) {
    if (g.counters[8] == 0) return;                                   // no group of this batch owns a comb
    const u32 lanes = g.counters[1];
    const u32 per = ((lanes + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK + 7) >> 3;
    const u32 local = blockIdx.x >> 3;
    if (local >= per) return;
    const u32 L = ((blockIdx.x & 7u) * per + local) * SBV_VERIFY_BLOCK + threadIdx.x;
    if (L >= lanes) return;
    const u32 t = g.grp_idx[L];
    const u32 grp = g.grp_of[L];
    if (!ed_wave_is_wide(grp, g, wide)) return;
    const unsigned long long am = __ballot(true);
    if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)am) - 1u) atomicAdd(wstat, (u32)__popcll(am));     // statistics: lanes of the wide pass
    const u32 slot = tslot[grp];                                      // a cache slot: wide[grp] says so
    const bool v = ed_qphase_wide_lane(tuples, t, kvalid[slot] != 0, wtab + (size_t)kwide[slot] * SBV_ED_HOT_COMB_BYTES, gacc, okb);
    acc[t] = v ? SBV_ED_PENDING : 0;
}
// Hot keys, per batch: the groups' tuple counts go to their cache slots' hit counters, and wide[k] = the slot owns a comb
// (p256_group.h: group_hot_class_lane); counters[8] = how many such groups (0 = the wide pass has no wavefront)
__global__ __launch_bounds__(256) void k_ed_hot_class(GroupState g, const u32* __restrict__ tslot, const uint8_t* __restrict__ cold, HotKeys hk,
                                                      uint8_t* __restrict__ wide) {
    const u32 groups = group_count(g);
    for (u32 k = blockIdx.x * 256 + threadIdx.x; k < groups; k += gridDim.x * 256) {
        group_hot_class_lane(k, g, tslot, cold, hk.cache_cap, hk.kwide, hk.khits, wide);
        if (wide[k]) atomicAdd(&g.counters[8], 1u);
    }
}
// The builder of this batch's promotions: lane (promotion i, window j, part) writes 32 entries of comb plist[2 i + 1] from the base point
// the slot's 8-bit comb holds (ed25519_group.h: ed_widetab_lane).  A bounded grid walks the promotions' 16 x 1024 lanes each; scratch
// is per resident lane.
__global__ __launch_bounds__(64) void k_ed_promote_window(const u32* __restrict__ plist, const u32* __restrict__ hot, const aniels* __restrict__ ktab,
                                                          u32* __restrict__ tmp, uint8_t* __restrict__ wtab) {
    const u32 total = promote_live(hot) * (SBV_ED_HOT_WINDOWS * SBV_ED_HOT_PARTS);
    const u32 gid = blockIdx.x * 64 + threadIdx.x;
    u32* mine = tmp + (size_t)gid * SBV_ED_HOT_TMP_WORDS;
    for (u32 item = gid; item < total; item += gridDim.x * 64) {
        const u32 part = item % SBV_ED_HOT_PARTS, j = (item / SBV_ED_HOT_PARTS) % SBV_ED_HOT_WINDOWS, i = item / (SBV_ED_HOT_PARTS * SBV_ED_HOT_WINDOWS);
        const u32 slot = plist[2 * i];
        if (slot == 0xFFFFFFFFu) continue;
        ed_widetab_lane(ktab + (size_t)slot * SBV_ED_KEYTAB_ENTRIES, j, part, mine, wtab + (size_t)plist[2 * i + 1] * SBV_ED_HOT_COMB_BYTES);
    }
}
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_ED_GROUP_WAVES) void k_ed_qphase_list(const uint8_t* __restrict__ tuples, GroupState g,
                                                                  const aniels* __restrict__ ktab, const uint8_t* __restrict__ kvalid,
                                                                  const u32* __restrict__ tslot, u32 table_slots,
                                                                  u32* __restrict__ gacc, size_t cap, const uint8_t* __restrict__ okb,
                                                                  uint8_t* __restrict__ acc, int j0, int j1, int last) {
    const u32 L = blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    if (L >= g.counters[1]) return;
    const u32 t = g.grp_idx[L];
    const u32 grp = g.slots[t];
    const bool v = ed_qphase_lane(tuples, t, grp < group_count(g) ? tslot[grp] : SBV_GROUP_NONE, table_slots, ktab, kvalid, gacc, cap, okb, j0, j1, last != 0);
    if (last) acc[t] = v ? SBV_ED_PENDING : 0;
}

// encode(R) == R_enc for the tuples the last Q chunk left pending: a lane takes SBV_ED_FINISH_T consecutive tuples and ONE
// inversion (ed25519_group.h: ed_finish_lane).  Few registers, 2^20 / 8 lanes: every wavefront of it is resident at once.
__global__ __launch_bounds__(256) void k_ed_finish(const uint8_t* __restrict__ tuples, size_t n, u32* __restrict__ gacc, size_t cap,
                                                   uint8_t* __restrict__ acc, int tuple_major) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * SBV_ED_FINISH_T;
    if (i0 < n) ed_finish_lane(tuples, n, i0, gacc, cap, acc, tuple_major != 0);
}

hipError_t launch_ed25519_verify_grouped(const uint8_t* d_tuples, size_t n, const GroupBuffers& b, const EdGroupBuffers& eb,
                                         u32* d_qtab, const aniels* d_btab, const edcomb& bcomb, uint8_t* d_bitmap, hipStream_t stream,
                                         const GroupSync& y, hipEvent_t* prof, int* prof_pairs) {
    if (n == 0) return hipSuccess;
    GroupState g;
    g.ht = b.ht; g.ht_mask = b.ht_mask; g.rep = b.rep; g.cnt = b.cnt; g.slot_of = b.slot_of; g.group_rep = b.group_rep;
    g.counters = b.counters; g.grp_idx = b.grp_idx; g.ung_idx = b.ung_idx; g.slots = b.slots;
    g.max_groups = b.max_groups; g.seed = b.seed;
    g.gcount = b.gcount; g.gcursor = b.gcount ? b.gcount + b.max_groups : nullptr; g.grp_of = b.grp_of; g.ung_cand = b.ung_cand;
    // key-sorted grouped list (p256_group.h): the Q phase walks runs of equal keys; the accumulator is tuple-major so that the
    // G phase can still start at once, in tuple order
    const size_t sort_lds = (size_t)b.max_groups * sizeof(u32);
    g.sorted = y.sorted && b.gcount && b.grp_of && sort_lds <= 64 * 1024 ? 1u : 0u;
    group_set_threshold(g, b.min_count);
    // 2 chunks (the context's setting) also after the finish left the Q phase: 3 / 4 chunks 4.41-4.51 / 4.36-4.45 ms against 4.19-4.33 per cold 2^20
    // step, 3.33-3.35 / 3.38-3.42 against 3.20-3.23 warm (profiles/r04/ab_ed_chunks_r04o.jsonl)
    const int chunks = y.chunks < 1 ? 1 : (y.chunks > SBV_GROUP_MAX_CHUNKS ? SBV_GROUP_MAX_CHUNKS : y.chunks);
    const int parts = SBV_KEYTAB_PARTS_DEFAULT;     // lanes per (key, window) of the table kernel (2 / 4 / 8 / 16 measured in round 2)
    hipError_t e;
#define SBV_TRY(x) do { if ((e = (x)) != hipSuccess) return e; } while (0)
    // ev_fork was recorded by the caller on `stream` before anything of this batch (see the P-256 launcher)
    SBV_TRY(hipStreamWaitEvent(y.side_a, y.ev_fork, 0));
    // the P-256 step's tail (table marks, promotion select: p256_group_kernels.hip) reads tslot / cold, which this scheme's grouping shares
    SBV_TRY(hipStreamWaitEvent(y.side_a, y.ev_promoted, 0));
    SBV_TRY(hipMemsetAsync(b.ht, 0, ((size_t)b.ht_mask + 1) * sizeof(u32), y.side_a));
    SBV_TRY(hipMemsetAsync(b.cnt, 0, n * sizeof(u32), y.side_a));
    SBV_TRY(hipMemsetAsync(b.counters, 0, SBV_GROUP_COUNTERS * sizeof(u32), y.side_a));
    if (g.sorted) SBV_TRY(hipMemsetAsync(b.gcount, 0, (size_t)b.max_groups * sizeof(u32), y.side_a));
    SBV_TRY(hipMemsetAsync(b.acc, 0, n, y.side_a));      // no stale "pending" marker can survive a batch that was cut short
    const unsigned gn = (unsigned)((n + 255) / 256);
    const unsigned gv = (unsigned)((n + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK);
    hipLaunchKernelGGL(k_ed_group_insert, dim3(gn), dim3(256), 0, y.side_a, d_tuples, n, g);
    hipLaunchKernelGGL((k_group_assign_t<128, 64, 8>), dim3(gn), dim3(256), 0, y.side_a, d_tuples, n, g, eb.kc);
    SBV_TRY(hipEventRecord(y.ev_assign, y.side_a));
    hipLaunchKernelGGL((k_key_cache_lookup_t<128, 64, 8>), dim3((b.max_groups + 63) / 64), dim3(64), 0, y.side_a, d_tuples, g, eb.kc, b.tslot, b.cold);
    hipLaunchKernelGGL((k_key_cache_insert_t<128, 64, 8>), dim3((b.max_groups + 63) / 64), dim3(64), 0, y.side_a, d_tuples, g, eb.kc, b.tslot);
    const u32 table_slots = eb.kc.cap + b.max_groups;
    // hot keys (ed25519_group.h): only with the cache on, a pool to promote into and the key-sorted list
    const bool hot_on = eb.wtab && eb.kwide && eb.wide && eb.kc.enabled && g.sorted;
    const HotKeys hk = {eb.wtab, hot_on ? eb.kwide : nullptr, eb.khits, eb.hot, eb.plist, eb.kc.cap, eb.wide_cap, eb.promote_min, eb.wowner, eb.elist};
    const uint8_t* wide = hot_on ? eb.wide : nullptr;
    if (hot_on) SBV_TRY(hipEventRecord(y.ev_cache, y.side_a));         // table slots assigned: the class kernel (side_b) reads them
    SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_assign, 0));
    if (hot_on) SBV_TRY(hipMemsetAsync(eb.hot + 1, 0, 3 * sizeof(u32), y.side_b));       // behind the previous batch's builder, which reads hot[1]
    if (g.sorted) {
        const unsigned tiles = (unsigned)((n + SBV_SORT_TILE - 1) / SBV_SORT_TILE);
        hipLaunchKernelGGL(k_group_classify, dim3(gn), dim3(256), 0, y.side_b, n, g, b.ung_cand, b.counters + 4);
        hipLaunchKernelGGL(k_group_sort_count, dim3(tiles), dim3(1024), sort_lds, y.side_b, n, g);
        hipLaunchKernelGGL(k_group_sort_scan, dim3(1), dim3(1024), 0, y.side_b, g);
        hipLaunchKernelGGL(k_group_sort_scatter, dim3(tiles), dim3(1024), sort_lds, y.side_b, n, g);
    } else {
        hipLaunchKernelGGL(k_ed_group_split, dim3(gn), dim3(256), 0, y.side_b, n, g);
    }
    if (hot_on) {
        SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_cache, 0));
        hipLaunchKernelGGL(k_ed_hot_class, dim3(64), dim3(256), 0, y.side_b, g, b.tslot, b.cold, hk, eb.wide);
    }
    SBV_TRY(hipEventRecord(y.ev_split, y.side_b));
    // stream: the G phase needs nothing but the tuples.  (Holding it back until the grouping's first kernels are through, so that the head
    // of the table pipeline does not share the CUs with it, was measured in round 6 and changes nothing: 3.84 against 3.81 ms cold.)
    hipLaunchKernelGGL(k_ed_gphase, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, stream, d_tuples, n, bcomb, b.gacc, b.gacc_cap, eb.okb, (int)g.sorted);
    if (hot_on) SBV_TRY(hipEventRecord(y.ev_class, stream));            // the G phase is enqueued: the wide pass (side_b) adds to its sums
    SBV_TRY(hipStreamWaitEvent(stream, y.ev_split, 0));
    // side_a: the doubling chains of every chunk, then the ungrouped list
    for (int c = 0; c < chunks; ++c) {
        const int j_first = SBV_ED_KEY_WINDOWS * c / chunks, j_end = SBV_ED_KEY_WINDOWS * (c + 1) / chunks;
        hipLaunchKernelGGL(k_ed_keytab_bases, dim3((b.max_groups * 4u + 63) / 64), dim3(64), 0, y.side_a, d_tuples, g, b.jbases, eb.kvalid,
                           b.tslot, b.cold, j_first, j_end - 1);
        SBV_TRY(hipEventRecord(y.ev_bases[c], y.side_a));
    }
    SBV_TRY(hipStreamWaitEvent(y.side_a, y.ev_split, 0));
    // the candidates whose key is a point -> ung_idx, right in front of the kernel that needs it: on side_b before the sort it held
    // up the first table windows (4.33 ms per cold 2^20 step against 4.08 here; without any key check 4.52: profiles/r04/ab_ed_keycheck_r04n.jsonl)
    // the ungrouped list on four lanes per tuple (the key check hands over the decompressed keys); SBV_ED_UNGROUPED_QUAD=0: the one-lane kernel
    static const bool quad_env = [] { const char* e = getenv("SBV_ED_UNGROUPED_QUAD"); return e && atoi(e) != 0; }();
    static const bool keep_keys = [] { const char* e = getenv("SBV_ED_UNGROUPED_KEYS"); return !e || atoi(e) != 0; }();
    const u32* ungxy = keep_keys && g.sorted ? eb.ungxy : nullptr;     // the key check keeps the keys it decompressed
    const bool quad = quad_env && ungxy;
    if (g.sorted) hipLaunchKernelGGL(k_ed_keycheck, dim3(gn), dim3(256), 0, y.side_a, d_tuples, g, b.acc, const_cast<u32*>(ungxy));
    // Once the combs are cached the one-lane kernel's chain (1.1 ms alone) is the step's critical path, and the key check in front of it
    // took 0.7 ms instead of 0.1 whenever its workgroups had to queue up behind those of the first Q launch (timeline_ed_hot_r06x.txt):
    // that launch now waits for the key check.  A cold step's first Q launch waits for tables anyway.
    static const bool keycheck_first = [] { const char* e = getenv("SBV_ED_KEYCHECK_FIRST"); return !e || atoi(e) != 0; }();
    if (g.sorted && keycheck_first) {
        SBV_TRY(hipEventRecord(y.ev_narrow, y.side_a));
        SBV_TRY(hipStreamWaitEvent(stream, y.ev_narrow, 0));
    }
    if (quad) hipLaunchKernelGGL(k_ed_generic_quad, dim3(4 * gv), dim3(SBV_VERIFY_BLOCK), 0, y.side_a, d_tuples, g, ungxy, d_qtab, d_btab, b.acc);
    else hipLaunchKernelGGL(k_ed_generic_list, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, y.side_a, d_tuples, g, d_qtab, d_btab, b.acc, ungxy);
    SBV_TRY(hipEventRecord(y.ev_generic, y.side_a));
    // (an uneven split — a first chunk of 6-12 windows so that the first Q launch starts earlier — was measured in round 5 and loses:
    // 4.11-4.46 ms per cold 2^20 step against 4.08-4.12, profiles/r05/ab_ed_chunk0_r05m.jsonl)
    for (int c = 0; c < chunks; ++c) {
        const int j_first = SBV_ED_KEY_WINDOWS * c / chunks, j_end = SBV_ED_KEY_WINDOWS * (c + 1) / chunks;
        const int j_count = j_end - j_first;
        hipStream_t tb = y.side_b;      // one table stream: the windows of the odd chunks on a second stream measured slower (round 4: 4.55 -> 4.67 ms)
        SBV_TRY(hipStreamWaitEvent(tb, y.ev_bases[c], 0));
        const size_t wl = (size_t)b.max_groups * j_count * parts;
        hipLaunchKernelGGL(k_ed_keytab_window, dim3((unsigned)((wl + 63) / 64)), dim3(64), 0, tb, g, b.jbases, b.tmp, eb.ktab,
                           b.tslot, b.cold, j_first, j_count, parts);
        SBV_TRY(hipEventRecord(y.ev_tables[c], tb));
        SBV_TRY(hipStreamWaitEvent(stream, y.ev_tables[c], 0));
        if (prof) SBV_TRY(hipEventRecord(prof[2 * c], stream));
        if (!g.sorted) hipLaunchKernelGGL(k_ed_qphase_list, dim3(gv), dim3(SBV_VERIFY_BLOCK), 0, stream, d_tuples, g, eb.ktab, eb.kvalid, b.tslot, table_slots, b.gacc, b.gacc_cap,
                                          eb.okb, b.acc, j_first, j_end, c + 1 == chunks ? 1 : 0);
        else if (c + 1 == chunks) hipLaunchKernelGGL(k_ed_qphase<true>, dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, stream, d_tuples, g, eb.ktab, eb.kvalid, b.tslot, table_slots,
                                                     b.gacc, b.gacc_cap, eb.okb, b.acc, j_first, j_end, wide);
        else hipLaunchKernelGGL(k_ed_qphase<false>, dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, stream, d_tuples, g, eb.ktab, eb.kvalid, b.tslot, table_slots,
                                b.gacc, b.gacc_cap, eb.okb, b.acc, j_first, j_end, wide);
        if (prof) SBV_TRY(hipEventRecord(prof[2 * c + 1], stream));
    }
    if (hot_on) {
        // The wide pass, on side_b behind this batch's table windows (none in a warm batch) and beside the chunks' launches: it needs the G
        // phase and the classes, no table.  On `stream` in front of the chunks' launches a batch with SOME hot signers ran three launches
        // in a row, each over a part of the lanes (half-hot 2^18: 1.45 ms against 1.31 from the 8-bit combs alone, ed_hot_sweep_r06al).
        // Like the first Q launch it lets the ungrouped list's key check go first.
        SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_class, 0));
        if (g.sorted && keycheck_first) SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_narrow, 0));
        hipLaunchKernelGGL(k_ed_qphase_wide, dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, y.side_b, d_tuples, g, eb.kvalid, b.tslot, wide, eb.kwide, eb.wtab,
                           eb.hot + 2, b.gacc, eb.okb, b.acc);
        SBV_TRY(hipEventRecord(y.ev_wide, y.side_b));
        SBV_TRY(hipStreamWaitEvent(stream, y.ev_wide, 0));
    }
    {   // the pending tuples' encodings: one inversion per SBV_ED_FINISH_T tuples
        const size_t fl = (n + SBV_ED_FINISH_T - 1) / SBV_ED_FINISH_T;
        hipLaunchKernelGGL(k_ed_finish, dim3((unsigned)((fl + 255) / 256)), dim3(256), 0, stream, d_tuples, n, b.gacc, b.gacc_cap, b.acc, (int)g.sorted);
    }
    if (hot_on) {
        // The tail on side_b, behind this batch's last table windows (the builder reads a promoted slot's 8-bit comb, which this very batch
        // may have built) and behind the wide pass (an evicted comb is rewritten) — stream order: the clock sweep, which slots get a comb, the
        // evictions; then — the next batch's side_a may rewrite tslot from here on — the combs themselves and their publication.
        // The next batch's class kernel queues up behind all of it on this stream: a comb is used only once it is complete.
        if (eb.hot_tick % SBV_HOT_DECAY_EVERY == SBV_HOT_DECAY_EVERY - 1) hipLaunchKernelGGL(k_hot_decay, dim3((eb.kc.cap + 255) / 256), dim3(256), 0, y.side_b, hk);
        hipLaunchKernelGGL(k_promote_select, dim3(64), dim3(256), 0, y.side_b, g, b.tslot, eb.kvalid, hk);
        hipLaunchKernelGGL(k_promote_evict, dim3(1), dim3(1024), 0, y.side_b, hk);
        SBV_TRY(hipEventRecord(y.ev_promoted, y.side_b));
        hipLaunchKernelGGL(k_ed_promote_window, dim3(SBV_ED_HOT_BUILD_BLOCKS), dim3(64), 0, y.side_b, eb.plist, eb.hot, eb.ktab, eb.ptmp, eb.wtab);
        hipLaunchKernelGGL(k_promote_publish, dim3(1), dim3(64), 0, y.side_b, eb.plist, eb.hot, eb.kwide, eb.wowner);
    }
    SBV_TRY(hipStreamWaitEvent(stream, y.ev_generic, 0));
    hipLaunchKernelGGL(k_pack_bitmap, dim3((unsigned)(((n + 7) / 8 + 255) / 256)), dim3(256), 0, stream, b.acc, n, d_bitmap);
#undef SBV_TRY
    if (prof && prof_pairs) *prof_pairs = chunks;
    return hipGetLastError();
}

}  // namespace sbv
