// k256_group_kernels.hip — the grouped step for secp256k1 (k256_group.h): the P-256 step's three-stream pipeline with this
// curve's kernels.  Grouping, classification, counting sort and the bitmap pack are the shared kernels of
// group_kernels_common.h; stage A is k_k256_prep (one lane per signature, also writes the per-tuple records).
//
//   stream: [prep] wait(sort) { generic stage B over the ungrouped list + G phase } wait(tables c) Q-phase chunk c ... pack
//   side_a: insert assign | chain chunk 0 | chain chunk 1
//   side_b:        wait(assign) classify keycheck sort | wait(chain c) rows fill of chunk c
//
// The curve has a comb pool and a persistent key-table cache of its own (KeyPool: slots [0, kc.cap) are kept between batches,
// [kc.cap, kc.cap + max_groups) are the per-batch area) — NOT the P-256 one: cache slots are found by the 64 key bytes, and a byte
// string can be a point of both curves, so a shared table would let a crafted key be verified against the other curve's comb.
#include <hip/hip_runtime.h>

#include <thread>
#include <vector>

#include "group_kernels_common.h"
#include "k256_group.h"
#include "p256_kernels.h"

namespace sbv {

struct KGlobalTupleG {
    const u32* p;
    __device__ __forceinline__ u32 operator[](int i) const { return p[i]; }
};
#define SBV_K256_PREP_T 8          // tuples per inversion in stage A of the grouped step
// T tuples per lane, one inversion per lane (k256_core.h: k256_prep_chunk); workgroup b covers tuples [b * 64 T, (b + 1) * 64 T),
// lane l of it tuples b * 64 T + l + 64 k: the lanes of a wavefront read 64 consecutive tuples in every pass
__global__ __launch_bounds__(64) void k_k256_prep_chunk(const uint8_t* __restrict__ tuples, size_t n, Scratch s, int T) {
    const size_t first = (size_t)blockIdx.x * 64 * (size_t)T + threadIdx.x;
    auto words = [&](size_t idx) { return KGlobalTupleG{reinterpret_cast<const u32*>(tuples + idx * 160)}; };
    k256_prep_chunk(words, n, s, first, (size_t)64, T);
}
__global__ __launch_bounds__(256) void k_k256_group_insert(const uint8_t* __restrict__ tuples, size_t n, GroupState g) {
    group_insert_block<160, 96, 16>(tuples, n, g);
}
__global__ __launch_bounds__(256) void k_k256_keycheck(const uint8_t* __restrict__ tuples, GroupState g, uint8_t* __restrict__ acc) {
    const u32 L = blockIdx.x * 256 + threadIdx.x;
    const u32 cands = g.counters[4];
    if (blockIdx.x * 256u >= cands) return;
    const bool active = L < cands;
    u32 i = 0;
    bool ok = false;
    if (active) {
        i = g.ung_cand[L];
        kfe x, y;
        ok = k256_key_load(tuples, i, x, y);
        if (!ok) acc[i] = 0;
    }
    const unsigned long long mr = __ballot(active && !ok);
    if ((threadIdx.x & 63) == 0 && mr) atomicAdd(&g.counters[3], (u32)__popcll(mr));
    const u32 pos = group_compact_pos(active && ok, &g.counters[2]);
    if (active && ok) g.ung_idx[pos] = i;
}

struct k256_quad_dev {
    static const int N = 1;
    kchain3 s[1];
    int r;
    __device__ __forceinline__ int role(int) const { return r; }
    __device__ __forceinline__ void bcast(kfe out[1], const kfe in[1], int src) const {
        SBV_UNROLL
        for (int l = 0; l < 9; ++l) {
            const int v = in[0].v[l];
            out[0].v[l] = src == 0 ? __builtin_amdgcn_mov_dpp(v, 0x00, 0xF, 0xF, true)
                        : src == 1 ? __builtin_amdgcn_mov_dpp(v, 0x55, 0xF, 0xF, true)
                                   : __builtin_amdgcn_mov_dpp(v, 0xAA, 0xF, 0xF, true);
        }
    }
};
// lanes = groups x 4; only the groups whose tables are built in this batch (cold); table slot of group k = tslot[k]
__global__ __launch_bounds__(64) void k_k256_chain(const uint8_t* __restrict__ tuples, GroupState g, u32* __restrict__ jstate, u32* __restrict__ bases,
                                                   uint8_t* __restrict__ valid, const u32* __restrict__ tslot, const uint8_t* __restrict__ cold,
                                                   int j_first, int j_last) {
    const u32 lane = blockIdx.x * 64 + threadIdx.x;
    const u32 k = lane >> 2;
    if (k >= group_count(g) || !cold[k]) return;
    k256_quad_dev q;
    q.r = (int)(lane & 3u);
    k256_chain_run(q, tuples, k, g, jstate, bases, valid + tslot[k], j_first, j_last);
}
__global__ __launch_bounds__(64, 2) void k_k256_rows(GroupState g, const u32* __restrict__ bases, u32* __restrict__ tmp, kapt* __restrict__ ktab,
                                                     const u32* __restrict__ tslot, const uint8_t* __restrict__ cold, int j_first, int j_count) {
    const u32 lane = blockIdx.x * 64 + threadIdx.x;
    const u32 which = lane & 1u, kw = lane >> 1;
    const u32 key = kw / (u32)j_count, j = (u32)j_first + kw % (u32)j_count;
    if (key >= group_count(g) || !cold[key]) return;
    if (which == 1 && j == SBV_GTAB_WINDOWS - 1) return;
    const size_t w = (size_t)key * SBV_GTAB_WINDOWS + j;
    k256_rows_lane(bases + w * SBV_K256_BASES_STRIDE, (int)which, j == SBV_GTAB_WINDOWS - 1,
                   tmp + w * SBV_K256_WINDOW_TMP + (size_t)which * SBV_K256_ROWS_TMP_WORDS,
                   ktab + ((size_t)tslot[key] * SBV_GTAB_WINDOWS + j) * SBV_GTAB_PER_WINDOW);
}
// lanes = groups x j_count x 7
__global__ __launch_bounds__(64) void k_k256_fill(GroupState g, u32* __restrict__ tmp, kapt* __restrict__ ktab, const u32* __restrict__ tslot,
                                                  const uint8_t* __restrict__ cold, int j_first, int j_count) {
    const u32 lane = blockIdx.x * 64 + threadIdx.x;
    const u32 r = lane % 7u, kw = lane / 7u;
    const u32 key = kw / (u32)j_count, j = (u32)j_first + kw % (u32)j_count;
    if (key >= group_count(g) || j == SBV_GTAB_WINDOWS - 1 || !cold[key]) return;
    const size_t w = (size_t)key * SBV_GTAB_WINDOWS + j;
    k256_fill_lane(1 + (int)r, tmp + w * SBV_K256_WINDOW_TMP + (size_t)r * SBV_K256_FILL_TMP_WORDS,
                   ktab + ((size_t)tslot[key] * SBV_GTAB_WINDOWS + j) * SBV_GTAB_PER_WINDOW);
}

// generic stage B over the ungrouped list (first blocks) + u1 * G over the key-sorted list
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, 2) void k_k256_gphase_generic(Scratch s, GroupState g, u32* __restrict__ qtab, const kapt* __restrict__ gtab,
                                                                           kgcomb gc, u32* __restrict__ gacc, uint8_t* __restrict__ acc, unsigned generic_blocks) {
    if (blockIdx.x < generic_blocks) {
        const u32 L = blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
        if (L >= g.counters[2]) return;
        const u32 t = g.ung_idx[L];
        acc[t] = k256_verify_lane(s, t, qtab + (size_t)L * SBV_QTAB29_WORDS, gtab) ? 1 : 0;
        return;
    }
    if (group_count(g) == 0) return;
    const size_t i = (size_t)(blockIdx.x - generic_blocks) * SBV_VERIFY_BLOCK + threadIdx.x;
    if (i < g.counters[1]) k256_gphase_lane_sorted(s, g.grp_idx[i], i, gc, gacc);
}
// (round 6: LAST is a template parameter, as in the P-256 Q phase — one continuation of the comb loop per kernel)
#ifndef SBV_K256_QPHASE_WAVES
#define SBV_K256_QPHASE_WAVES 2      // 3 waves: 17-25 spilled dwords (the Jacobian addition on 64-bit columns needs them), measured no faster in rounds 3-4
#endif
template <bool LAST>
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_K256_QPHASE_WAVES) void k_k256_qphase(Scratch s, GroupState g, const kapt* __restrict__ ktab, const uint8_t* __restrict__ kvalid,
                                                                   const u32* __restrict__ tslot, u32 table_slots,
                                                                   u32* __restrict__ gacc, uint8_t* __restrict__ acc, int j0, int j1) {
    // key-sorted list, XCD-aware block order (p256_group_kernels.hip: k_verify_keyed_q)
    const u32 lanes = g.counters[1];
    const u32 per = ((lanes + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK + 7) >> 3;
    const u32 local = blockIdx.x >> 3;
    if (local >= per) return;
    const u32 L = ((blockIdx.x & 7u) * per + local) * SBV_VERIFY_BLOCK + threadIdx.x;
    if (L >= lanes) return;
    const u32 t = g.grp_idx[L];
    const u32 grp = g.grp_of[L];
    const bool v = k256_qphase_lane_sorted(s, t, L, grp < group_count(g) ? tslot[grp] : SBV_GROUP_NONE, table_slots, ktab, kvalid, gacc, j0, j1, LAST);
    if (LAST) acc[t] = v ? 1 : 0;
}

// stage A + stage B of a grouped secp256k1 batch.  ev_fork must have been recorded on `stream` first.  kp: this curve's comb
// pool and key-table cache; b supplies the grouping arrays, the chain records, tslot / cold and the accumulators.
hipError_t launch_k256_verify_grouped(const uint8_t* d_tuples, const Scratch& s_in, size_t n, const GroupBuffers& b, const KeyPool& kp, u32* d_qtab,
                                      const kapt* d_gtab, const kapt* d_gcomb, int gcomb_bits, uint8_t* d_bitmap, hipStream_t stream, const GroupSync& y, hipEvent_t* prof, int* prof_pairs) {
    if (n == 0) return hipSuccess;
    GroupState g;
    g.ht = b.ht; g.ht_mask = b.ht_mask; g.rep = b.rep; g.cnt = b.cnt; g.slot_of = b.slot_of; g.group_rep = b.group_rep;
    g.counters = b.counters; g.grp_idx = b.grp_idx; g.ung_idx = b.ung_idx; g.slots = b.slots; g.max_groups = b.max_groups; g.seed = b.seed;
    g.gcount = b.gcount; g.gcursor = b.gcount + b.max_groups; g.grp_of = b.grp_of; g.ung_cand = b.ung_cand;
    g.sorted = 1u;
    group_set_threshold(g, b.min_count);
    Scratch s = s_in;
    s.rec = b.rec;
    const size_t sort_lds = (size_t)b.max_groups * sizeof(u32);
    kapt* ktab = reinterpret_cast<kapt*>(kp.ktab);
    uint8_t* kvalid = kp.kvalid;
    const u32 table_slots = kp.kc.cap + b.max_groups;
    hipError_t e;
#define SBV_TRY(x) do { if ((e = (x)) != hipSuccess) return e; } while (0)
    SBV_TRY(hipStreamWaitEvent(y.side_a, y.ev_fork, 0));
    // the P-256 step's tail (table marks, promotion select: p256_group_kernels.hip) reads tslot / cold, which this scheme's grouping shares
    SBV_TRY(hipStreamWaitEvent(y.side_a, y.ev_promoted, 0));
    SBV_TRY(hipMemsetAsync(b.ht, 0, ((size_t)b.ht_mask + 1) * sizeof(u32), y.side_a));
    SBV_TRY(hipMemsetAsync(b.cnt, 0, n * sizeof(u32), y.side_a));
    SBV_TRY(hipMemsetAsync(b.counters, 0, SBV_GROUP_COUNTERS * sizeof(u32), y.side_a));
    SBV_TRY(hipMemsetAsync(b.gcount, 0, (size_t)b.max_groups * sizeof(u32), y.side_a));
    const unsigned gn = (unsigned)((n + 255) / 256);
    const unsigned gv = (unsigned)((n + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK);
    hipLaunchKernelGGL(k_k256_group_insert, dim3(gn), dim3(256), 0, y.side_a, d_tuples, n, g);
    hipLaunchKernelGGL(k_group_assign, dim3(gn), dim3(256), 0, y.side_a, d_tuples, n, g, kp.kc);
    SBV_TRY(hipEventRecord(y.ev_assign, y.side_a));
    hipLaunchKernelGGL((k_key_cache_lookup_t<160, 96, 16>), dim3((b.max_groups + 63) / 64), dim3(64), 0, y.side_a, d_tuples, g, kp.kc, b.tslot, b.cold);
    hipLaunchKernelGGL((k_key_cache_insert_t<160, 96, 16>), dim3((b.max_groups + 63) / 64), dim3(64), 0, y.side_a, d_tuples, g, kp.kc, b.tslot);
    {   // stage A: SBV_K256_PREP_T tuples per lane share one inversion (Montgomery's trick).  Measured in round 4
        // (profiles/r04/ab_k256_prep_t_r04a.jsonl): 1 / 4 / 8 tuples per inversion 4.97 / 4.79 / 4.68 ms per 2^20 step.
        const size_t per_block = (size_t)64 * SBV_K256_PREP_T;
        hipLaunchKernelGGL(k_k256_prep_chunk, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(64), 0, stream, d_tuples, n, s, SBV_K256_PREP_T);
    }
    SBV_TRY(hipStreamWaitEvent(y.side_b, y.ev_assign, 0));
    hipLaunchKernelGGL(k_group_classify, dim3(gn), dim3(256), 0, y.side_b, n, g, b.ung_cand, b.counters + 4);
    hipLaunchKernelGGL(k_k256_keycheck, dim3(gn), dim3(256), 0, y.side_b, d_tuples, g, b.acc);
    const unsigned tiles = (unsigned)((n + SBV_SORT_TILE - 1) / SBV_SORT_TILE);
    hipLaunchKernelGGL(k_group_sort_count, dim3(tiles), dim3(1024), sort_lds, y.side_b, n, g);
    hipLaunchKernelGGL(k_group_sort_scan, dim3(1), dim3(1024), 0, y.side_b, g);
    hipLaunchKernelGGL(k_group_sort_scatter, dim3(tiles), dim3(1024), sort_lds, y.side_b, n, g);
    SBV_TRY(hipEventRecord(y.ev_split, y.side_b));
    SBV_TRY(hipStreamWaitEvent(stream, y.ev_split, 0));
    hipLaunchKernelGGL(k_k256_gphase_generic, dim3(2 * gv), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, d_qtab, d_gtab, kgcomb_make(d_gcomb, gcomb_bits), b.gacc, b.acc, gv);
    const int chunks = 2;
    for (int c = 0; c < chunks; ++c) {
        const int j_first = SBV_GTAB_WINDOWS * c / chunks, j_end = SBV_GTAB_WINDOWS * (c + 1) / chunks, j_count = j_end - j_first;
        hipLaunchKernelGGL(k_k256_chain, dim3((b.max_groups * 4 + 63) / 64), dim3(64), 0, y.side_a, d_tuples, g, b.jstate, b.bases, kvalid, b.tslot, b.cold, j_first, j_end - 1);
        SBV_TRY(hipEventRecord(y.ev_bases[c], y.side_a));
        hipStream_t tb = y.tstreams > 1 && y.side_t && (c & 1) ? y.side_t : y.side_b;     // as the P-256 step: rows + fill of chunk 1 do not queue behind chunk 0's
        SBV_TRY(hipStreamWaitEvent(tb, y.ev_bases[c], 0));
        const size_t wl = (size_t)b.max_groups * j_count * 2, fl = (size_t)b.max_groups * j_count * 7;
        hipLaunchKernelGGL(k_k256_rows, dim3((unsigned)((wl + 63) / 64)), dim3(64), 0, tb, g, b.bases, b.tmp, ktab, b.tslot, b.cold, j_first, j_count);
        hipLaunchKernelGGL(k_k256_fill, dim3((unsigned)((fl + 63) / 64)), dim3(64), 0, tb, g, b.tmp, ktab, b.tslot, b.cold, j_first, j_count);
        SBV_TRY(hipEventRecord(y.ev_tables[c], tb));
        SBV_TRY(hipStreamWaitEvent(stream, y.ev_tables[c], 0));
        if (prof) SBV_TRY(hipEventRecord(prof[2 * c], stream));
        if (c + 1 == chunks) hipLaunchKernelGGL(k_k256_qphase<true>, dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, ktab, kvalid, b.tslot, table_slots, b.gacc, b.acc, j_first, j_end);
        else hipLaunchKernelGGL(k_k256_qphase<false>, dim3((gv + 7u) & ~7u), dim3(SBV_VERIFY_BLOCK), 0, stream, s, g, ktab, kvalid, b.tslot, table_slots, b.gacc, b.acc, j_first, j_end);
        if (prof) SBV_TRY(hipEventRecord(prof[2 * c + 1], stream));
    }
    hipLaunchKernelGGL(k_pack_bitmap, dim3((unsigned)(((n + 7) / 8 + 255) / 256)), dim3(256), 0, stream, b.acc, n, d_bitmap);
#undef SBV_TRY
    if (prof && prof_pairs) *prof_pairs = chunks;
    return hipGetLastError();
}

}  // namespace sbv

namespace sbv {
// the `bits`-wide comb of G for the grouped step, one host thread per window (k256_core.h: k256_build_g_window_bits)
void host_build_k256_gcomb(int bits, kapt* out) {
    const int windows = (257 + bits - 1) / bits;
    std::vector<std::thread> th;
    for (int j = 0; j < windows; ++j)
        th.emplace_back([=] { k256_build_g_window_bits(bits, j, out + ((size_t)j << (bits - 1)), 1 << (bits - 1)); });
    for (auto& t : th) t.join();
}
}  // namespace sbv
