// p256_group.h — grouping a batch of generic tuples by public key, inside the step.
//
// Signers repeat: BASELINE.json's headline batch has 1024 distinct keys in 2^20 tuples; SmartBFT's
// consenters are a handful.  For a key used by several tuples of ONE batch it pays to
// build that key's comb table once and verify all its signatures without doublings instead of paying
// 256 doublings per signature.  Verdicts are identical to the generic path.  The lanes of the step, in
// the order the launcher (p256_group_kernels.hip) runs them:
//
//   group_insert   every tuple inserts its 64-byte key into an open-addressing hash table in HBM
//                  (32-bit entries = representative tuple index + 1, claimed with atomicCAS; a hash
//                  hit is only trusted after comparing all 64 key bytes, so collisions cost a probe,
//                  never a wrong group; seeded and probe-bounded since round 5: "chosen keys" below)
//                  and counts itself on its representative (exactly, since round 5; the sampled form
//                  of rounds 1-4 stays behind group_set_sampling)
//   group_assign   representatives with >= min_count users, and every key the persistent cache holds,
//                  take a group index (atomic counter; up to 65 536 per batch)
//   key_cache_*    the group's table slot: its cache slot, or a per-batch slot
//   classify / keycheck / sort_*   the grouped tuples as a list sorted by group (counting sort), the
//                  ungrouped ones as a list of their own; ungrouped tuples with a key that
//                  pointFromAffine refuses are rejected on the spot
//   table_class    which groups get rows only, which a full 8-bit table, which own a wide comb ("table
//                  classes", "hot keys" below)
//   keychain / rows / fill (p256_keytab29.h)   per cold group: validate the key, 2^(8j) * Q for j = 0..32,
//                  babies and giants of every window, then (full tables only) the other 112 entries,
//                  in chunks of windows so that table building and use are pipelined
//   gphase         u1 * G for every grouped tuple (p256_comb29.h), independent of the tables
// then the Q phase (qphase29_lane*, p256_comb29.h) runs over the grouped list in one instantiation per
// table class and the generic stage B over the ungrouped list.
//
// Shared host/device source (tests/emul runs the same functions sequentially).
#pragma once
#include "p256_core.h"
#include "p256_comb29.h"

namespace sbv {

#if defined(__HIP_DEVICE_COMPILE__)
#define SBV_ATOMIC_CAS(p, cmp, val) atomicCAS((p), (cmp), (val))
#define SBV_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#else
SBV_HD u32 host_cas(u32* p, u32 cmp, u32 val) { const u32 old = *p; if (old == cmp) *p = val; return old; }
SBV_HD u32 host_add(u32* p, u32 v) { const u32 old = *p; *p = old + v; return old; }
#define SBV_ATOMIC_CAS(p, cmp, val) host_cas((p), (cmp), (val))
#define SBV_ATOMIC_ADD(p, v) host_add((p), (v))
#endif

#define SBV_GROUP_NONE 0xFFFFFFFFu
#define SBV_GROUP_COUNTERS 12
// Hash-flooding defence of the two open-addressing tables below (in a BFT library the adversary is the
// design point, and the keys of a batch are attacker-chosen bytes that are inserted BEFORE any curve check):
//   * both hashes are keyed with a per-context random seed (GroupState::seed, KeyCache::seed): key bytes that collide under one
//     seed are unrelated under another, so collisions cannot be precomputed;
//   * a probe sequence is followed for at most SBV_GROUP_MAX_PROBES entries.  A tuple that has not found its key by then simply
//     stays UNGROUPED (it represents itself and takes the generic kernel: same verdict, no table), a cache lookup answers "not
//     cached".  At the tables' load factors (<= 0.5 and <= 0.25) a random key needs more than 64 probes with negligible
//     probability; whatever an adversary manages, insert work per tuple is bounded by 64 comparisons of 64 bytes instead of n.
#define SBV_GROUP_MAX_PROBES 64u

struct GroupState {
    u32* ht;          // hash table, ht_mask + 1 entries, zeroed before every batch
    u32 ht_mask;
    u32* rep;         // [n] representative tuple of tuple i's key
    u32* cnt;         // [n] users of a representative (zeroed before every batch)
    u32* slot_of;     // [n] table slot of a representative, or NONE
    u32* group_rep;   // [max_groups] representative tuple of slot g
    u32* counters;    // [0] groups handed out, [1] grouped tuples, [2] ungrouped tuples, [3] rejected for their key, [4] ungrouped candidates (SBV_GROUP_COUNTERS words, zeroed)
                      // P-256 table classes: [5] groups with a full table, [6] groups filled in this batch, [7] lanes of the rows-only pass, [8] groups that may take the wide pass
    u32* grp_idx;     // [n] compacted grouped tuple indices
    u32* ung_idx;     // [n] compacted ungrouped tuple indices
    u32* slots;       // [n] group of every tuple (SBV_GROUP_NONE for the ungrouped and rejected ones)
    u32* gcount;      // [max_groups] exact number of tuples of each group (key-sorted list only; zeroed before every batch)
    u32* gcursor;     // [max_groups] next free position of each group's run in grp_idx
    u32* grp_of;      // [n] group of lane L of the key-sorted list (= slots[grp_idx[L]])
    u32* ung_cand;    // [n] key-sorted step: ungrouped tuples before their keys are checked (group_classify_lane)
    u32 sorted;       // 1: grp_idx is built by the counting sort below (runs of equal keys), 0: by the split's compaction
    u32 max_groups;
    u32 min_count;    // requested threshold (users of a key in this batch)
    u32 sample_mask;  // tuples with group_sampled(i, sample_mask) are counted
    u32 min_samples;  // threshold on the sampled count
    u32 seed;         // key of the grouping hash (per context, random; 0 in the emulator unless a test sets it)
};

// Sampling rate for a threshold: thresholds from 16 on count every 8th tuple, smaller ones (tests, tiny batches) count exactly.
// The counting atomics sit on the path to the G phase, so the rate does NOT grow when the threshold drops from round 1-4's 64 (8
// samples) to round 5's default of 16 (a key's rows pay from ~4 signatures on, p256_comb29.h): a key then passes with 2 of its
// every-8th tuples — a soft edge (a 16-use key passes with probability 0.61, a 32-use key 0.91, a 64-use key 0.998, a 4-use key
// 0.08), which only decides who gets a table, never a verdict.
SBV_HD void group_set_sampling(GroupState& g, u32 min_count, u32 shift) {
    g.min_count = min_count;
    g.sample_mask = (1u << shift) - 1u;
    const u32 ms = min_count >> shift;
    g.min_samples = ms ? ms : 1u;
}
SBV_HD void group_set_threshold(GroupState& g, u32 min_count) { group_set_sampling(g, min_count, min_count >= 16 ? 3u : 0u); }
// The P-256 step's built-in default (round 5, sbv_api.hip): 8 uses, counted EXACTLY (every tuple).  Measured on the 2^20 sweep
// (profiles/r05/key_sweep_shift*_r05g.txt, one box, one session): exact / every 2nd / every 4th tuple — headline 325.0 / 324.6 / 320.8 M/s
// (the counting atomics spread over >= 1024 words and cost nothing measurable; the sampled thresholds instead admit repeated
// bit-flipped variants of the signers' keys as groups: 1024 / 1072 / 2170 groups), 65 536 keys x 16 uses 79.5 / 80.4 / 80.0 M/s,
// 262 144 keys x 4 uses 60.8 / 55.5 / 46.1 M/s (a 4-use key's rows do not pay; a soft threshold lets a quarter of them through).
#define SBV_GROUP_MIN_COUNT_DEFAULT 8u
#define SBV_GROUP_SAMPLE_SHIFT_DEFAULT 0

// Which tuples are counted: a multiplicative hash of the index, NOT its low bits — batches are often laid
// out round-robin over the signers (tuple i signed by key i mod K), and `i & mask` would then count only
// every 2^k-th KEY.  (An adversarial layout can still dodge the sample; that only costs its own batch the
// table speed-up, never a verdict.)
SBV_HD bool group_sampled(u32 i, u32 sample_mask) {
    u32 h = i * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return ((h >> 24) & sample_mask) == 0;
}

SBV_HD const u32* tuple_key_words(const uint8_t* tuples, size_t i) {
    return reinterpret_cast<const u32*>(tuples + i * 160 + 96);       // 16-byte aligned: 160 i + 96
}

// pointFromAffine's verdict on tuple idx's key, on the carry-free field: coordinates < p, y^2 = x^3 - 3x + b.
// x, y: the key in the R = 2^261 domain, tight (garbage when the verdict is false).
SBV_HD bool key29_load(const uint8_t* tuples, size_t idx, fe29& x, fe29& y) {
    const u32* k = tuple_key_words(tuples, idx);
    u256 qx, qy;
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) { qx.v[l] = bswap32(k[7 - l]); qy.v[l] = bswap32(k[8 + 7 - l]); }
    const fe p_ = fe_p();
    f29_from_plain(x, qx);
    f29_from_plain(y, qy);
    fe29 lhs, t, rhs;
    f29_sqr(lhs, y);
    f29_sqr(t, x);
    f29_mul(rhs, t, x);
    f29_sub(rhs, rhs, x);
    f29_sub(rhs, rhs, x);
    f29_sub(rhs, rhs, x);
    f29_add(rhs, rhs, f29_b());
    f29_sub(t, lhs, rhs);
    return lt256(qx, p_) && lt256(qy, p_) && f29_is_zero(t);
}
// the verdict alone (the grouping kernels' key check; round 5: on the carry-free field like everything else in the product — the
// 8 x 32 Montgomery form of round 1 left the product headers)
SBV_HD bool tuple_key_ok(const uint8_t* tuples, size_t idx) {
    fe29 x, y;
    return key29_load(tuples, idx, x, y);
}

// Key location of a tuple format: STRIDE bytes per tuple, key = WORDS dwords at byte OFF (16-byte aligned).
// P-256: 160 / 96 / 16 (Qx|Qy); Ed25519: 128 / 64 / 8 (A_enc).
// group_find_rep_t: the representative of tuple i's key (claims the hash-table entry when the key is new) -> g.rep[i]; the caller
// counts the tuple: group_insert_lane_t one atomic per tuple (the emulator; any caller without a workgroup), the kernels through
// group_count_block (group_kernels_common.h: aggregated per workgroup — a consenter batch puts 34 000 tuples on each of 16 counters).
template <int STRIDE, int OFF, int WORDS>
SBV_HD u32 group_find_rep_t(const uint8_t* tuples, size_t i, const GroupState& g) {
    const u32* k = reinterpret_cast<const u32*>(tuples + i * STRIDE + OFF);
    u32 w[WORDS];
    SBV_UNROLL
    for (int j = 0; j < WORDS; ++j) w[j] = k[j];
    // every key word goes into the hash: the batch's corrupted tuples are single-bit variants of the signers'
    // keys, and a hash that skips words sends each variant down its original's probe chain (measured: ~30
    // probes per wavefront, 1 ms per batch)
    u32 h = 0x9E3779B1u ^ g.seed;
    SBV_UNROLL
    for (int j = 0; j < WORDS; ++j) {
        h = (h ^ w[j]) * 0x85EBCA77u;
        h ^= h >> 15;
    }
    h = (h ^ (g.seed * 0x27D4EB2Fu)) * 0xC2B2AE3Du;      // the seed once more behind the last key word: the final mix is keyed too
    h ^= h >> 16;
    u32 slot = h & g.ht_mask;
    u32 mine = (u32)i;              // no group found within the probe bound: the tuple represents itself (ungrouped, generic kernel)
    for (u32 probes = 0; probes < SBV_GROUP_MAX_PROBES && probes <= g.ht_mask; ++probes) {
        u32 v = g.ht[slot];
        if (v == 0) v = SBV_ATOMIC_CAS(&g.ht[slot], 0u, (u32)i + 1u);
        if (v == 0) break;                                   // claimed: this tuple represents its key
        const u32* o = reinterpret_cast<const u32*>(tuples + (size_t)(v - 1) * STRIDE + OFF);
        u32 diff = 0;
        SBV_UNROLL
        for (int j = 0; j < WORDS; ++j) diff |= o[j] ^ w[j];
        if (diff == 0) { mine = v - 1; break; }
        slot = (slot + 1) & g.ht_mask;
    }
    g.rep[i] = mine;
    return mine;
}
template <int STRIDE, int OFF, int WORDS>
SBV_HD void group_insert_lane_t(const uint8_t* tuples, size_t i, const GroupState& g) {
    const u32 mine = group_find_rep_t<STRIDE, OFF, WORDS>(tuples, i, g);
    if (group_sampled((u32)i, g.sample_mask)) SBV_ATOMIC_ADD(&g.cnt[mine], 1u);
}
SBV_HD void group_insert_lane(const uint8_t* tuples, size_t i, const GroupState& g) { group_insert_lane_t<160, 96, 16>(tuples, i, g); }

// Ungrouped tuples whose key pointFromAffine refuses are rejected here and never reach the generic
// kernel (crypto/ecdsa returns false before any scalar multiplication, too): in a SIMT kernel an early
// exit only pays when whole wavefronts take it, so the filter has to sit in front of the compaction.
// counters[3] counts them (stats only).
SBV_HD bool group_split_lane(const uint8_t* tuples, size_t i, const GroupState& g, uint8_t* acc) {
    const u32 s = g.slot_of[g.rep[i]];
    if (s == SBV_GROUP_NONE) {
        g.slots[i] = SBV_GROUP_NONE;
        if (!tuple_key_ok(tuples, i)) {
            acc[i] = 0;
            SBV_ATOMIC_ADD(&g.counters[3], 1u);
            return false;
        }
        g.ung_idx[SBV_ATOMIC_ADD(&g.counters[2], 1u)] = (u32)i;
    } else {
        g.slots[i] = s;
        if (!g.sorted) g.grp_idx[SBV_ATOMIC_ADD(&g.counters[1], 1u)] = (u32)i;
    }
    return true;
}

// The key-sorted step splits in two passes: every tuple is classified (its group, or a candidate for the ungrouped list), then
// only the candidates — a few per cent of a batch, compacted, so whole wavefronts do the same thing — have their keys checked.
// (In one pass nearly every wavefront holds a few ungrouped lanes and pays the curve equation for all 64: 0.15 ms at 2^20, on
// the path to the G phase.)  Same lists, counters and verdicts as group_split_lane.
SBV_HD void group_classify_lane(size_t i, const GroupState& g) {
    const u32 s = g.slot_of[g.rep[i]];
    g.slots[i] = s;
    if (s == SBV_GROUP_NONE) g.ung_cand[SBV_ATOMIC_ADD(&g.counters[4], 1u)] = (u32)i;
}
SBV_HD bool group_keycheck_lane(const uint8_t* tuples, size_t L, const GroupState& g, uint8_t* acc) {
    const u32 i = g.ung_cand[L];
    if (!tuple_key_ok(tuples, i)) {
        acc[i] = 0;
        SBV_ATOMIC_ADD(&g.counters[3], 1u);
        return false;
    }
    g.ung_idx[SBV_ATOMIC_ADD(&g.counters[2], 1u)] = i;
    return true;
}

// ---- key-sorted grouped list -------------------------------------------------------------------------------------------
// The Q phase gathers one 64-byte table entry per addition from its key's comb (33 rows of 8 KB).  With the grouped list in
// tuple order the 64 lanes of a wavefront belong to 64 different keys and every gather is a fresh line from HBM (measured:
// 2.2 GB per launch for 0.05 GB of algorithmic bytes).  A counting sort by group — exact per-group counts, exclusive scan,
// scatter — makes runs of equal keys: a wavefront reads ONE key's rows, a key's ~n/keys tuples sit in neighbouring
// workgroups, and with the XCD-aware block order of the Q phase a key's rows are fetched into one L2 once.  The order inside
// a run is arbitrary (atomics) and irrelevant: verdicts are written by tuple index.
//   count    gcount[group of tuple i] += 1                      (device: LDS histogram per workgroup, then global atomics)
//   scan     gcursor[k] = sum of gcount[< k]; counters[1] = total
//   scatter  L = gcursor[group]++; grp_idx[L] = i; grp_of[L] = group
SBV_HD void group_sort_count_lane(size_t i, const GroupState& g) {
    const u32 s = g.slots[i];
    if (s != SBV_GROUP_NONE) SBV_ATOMIC_ADD(&g.gcount[s], 1u);
}
// The order of the groups' runs in the list (round 6): 0, 8, 16, ..., 1, 9, 17, ... instead of 0, 1, 2, ...  Groups are numbered by first
// appearance and cache slots by first caching, so the classes of a batch — hot keys, cold keys, rows-only tables — tend to be RANGES of
// group numbers; with the runs in numerical order a class was a range of the list, i.e. the share of a few XCDs (every XCD walks one
// contiguous eighth, so that a key's comb is fetched into one L2): a pass over half the lanes took as long as one over all of them
// (profiles/r06/timeline_ed_half_hot_r06ak.txt).  With the runs dealt out by group number modulo 8 every eighth of the list holds every
// eighth group of any range.  Position p of rows * 8 -> group (p % rows) * 8 + p / rows; a position past the last group is an empty run.
#ifndef SBV_SORT_STRIDE8
#define SBV_SORT_STRIDE8 1
#endif
SBV_HD u32 group_sort_rows(u32 groups) { return SBV_SORT_STRIDE8 ? (groups + 7u) >> 3 : groups; }
SBV_HD u32 group_sort_positions(u32 groups) { return SBV_SORT_STRIDE8 ? group_sort_rows(groups) * 8u : groups; }
SBV_HD u32 group_sort_group_at(u32 p, u32 rows) { return SBV_SORT_STRIDE8 ? (p % rows) * 8u + p / rows : p; }
SBV_HD void group_sort_scan_seq(const GroupState& g, u32 groups) {
    u32 run = 0;
    const u32 rows = group_sort_rows(groups), P = group_sort_positions(groups);
    for (u32 p = 0; p < P; ++p) {
        const u32 k = group_sort_group_at(p, rows);
        if (k < groups) { g.gcursor[k] = run; run += g.gcount[k]; }
    }
    g.counters[1] = run;
}
SBV_HD void group_sort_scatter_lane(size_t i, const GroupState& g) {
    const u32 s = g.slots[i];
    if (s == SBV_GROUP_NONE) return;
    const u32 L = SBV_ATOMIC_ADD(&g.gcursor[s], 1u);
    g.grp_idx[L] = (u32)i;
    g.grp_of[L] = s;
}

// ---- persistent key-table cache (across batches) -----------------------------------------------------------------------
// A key's comb is a pure function of its 64 bytes, and signer sets are stable for whole epochs (consenters:
// pkg/types/types.go:25-29; reconfiguration: pkg/consensus/consensus.go:185-252), so the tables a batch builds are kept:
// table slots [0, cap) of the comb pool are owned by the cache, slots [cap, cap + max_groups) are the per-batch area used
// when the cache is off or full.  After the groups of a batch are assigned, every group looks its key up (open addressing,
// entries = slot + 1, a hit only after comparing all 16 key words); a miss takes a fresh slot and the group is "cold":
// only cold groups run the table-building kernels.  Verdicts cannot depend on the cache: a slot is keyed by the exact key
// bytes and holds exactly what the batch would have built.  tslot[g] = table slot of group g, cold[g] = build it now.
struct KeyCache {
    u32* ht;          // ht_mask + 1 entries
    u32 ht_mask;
    u32* keys;        // [cap][16] key words of the cached slots
    u32* count;       // [0] slots handed out (may overshoot cap), [1] hits, [2] misses of the current batch
    u32 cap;
    u32 enabled;
    u32 seed;         // key of key_hash16 (per cache, random, fixed while the cache holds entries)
};
SBV_HD u32 key_hash16(const u32 w[16], u32 seed) {
    u32 h = 0x9E3779B1u ^ seed;
    SBV_UNROLL
    for (int j = 0; j < 16; ++j) {
        h = (h ^ w[j]) * 0x85EBCA77u;
        h ^= h >> 15;
    }
    h = (h ^ (seed * 0x27D4EB2Fu)) * 0xC2B2AE3Du;
    return h ^ (h >> 16);
}
// read-only phase: slot of a cached key, or NONE (also when the key is not among the first SBV_GROUP_MAX_PROBES entries of its
// probe sequence: the group is then built as a cold one — same verdicts)
SBV_HD u32 key_cache_lookup(const KeyCache& kc, const u32 w[16]) {
    u32 p = key_hash16(w, kc.seed) & kc.ht_mask;
    for (u32 probes = 0; probes < SBV_GROUP_MAX_PROBES && probes <= kc.ht_mask; ++probes) {
        const u32 v = kc.ht[p];
        if (v == 0) return SBV_GROUP_NONE;
        const u32* o = kc.keys + (size_t)(v - 1) * 16;
        u32 diff = 0;
        SBV_UNROLL
        for (int j = 0; j < 16; ++j) diff |= o[j] ^ w[j];
        if (diff == 0) return v - 1;
        p = (p + 1) & kc.ht_mask;
    }
    return SBV_GROUP_NONE;
}
// insert phase (keys of one batch are distinct, and none of them was found by the lookup phase): a fresh slot, or NONE
// when the cache is full
SBV_HD u32 key_cache_insert(const KeyCache& kc, const u32 w[16]) {
    if (kc.count[0] >= kc.cap) return SBV_GROUP_NONE;      // full (the counter is monotone: a stale read only delays this)
    const u32 slot = SBV_ATOMIC_ADD(&kc.count[0], 1u);
    if (slot >= kc.cap) return SBV_GROUP_NONE;
    u32* o = kc.keys + (size_t)slot * 16;
    SBV_UNROLL
    for (int j = 0; j < 16; ++j) o[j] = w[j];
    // within the probe bound, or not published at all: the slot then serves this batch only (its table is built and used through
    // tslot) and later batches rebuild the key — a cost an adversary who cannot see the seed cannot aim
    u32 p = key_hash16(w, kc.seed) & kc.ht_mask;
    for (u32 probes = 0; probes < SBV_GROUP_MAX_PROBES && probes <= kc.ht_mask; ++probes) {
        if (SBV_ATOMIC_CAS(&kc.ht[p], 0u, slot + 1u) == 0u) break;
        p = (p + 1) & kc.ht_mask;
    }
    return slot;
}
// A representative takes a table slot when its key is used often enough in THIS batch — or when the scheme's persistent cache
// already holds its comb (kc.enabled), however few of its signatures the batch carries: a warm key costs nothing to
// "build", so even a batch of a few thousand tuples then runs the comb phases instead of 256 doublings per signature.
// The lookup is read-only (everything in the cache was inserted by earlier batches) and runs for the representatives below
// the threshold only.  STRIDE / OFF / WORDS: where a tuple format keeps its key (group_insert_lane_t); the cache compares
// 16 words, a shorter key (Ed25519: 8) is padded with zeros.
template <int STRIDE, int OFF, int WORDS>
SBV_HD void key_words16_t(const uint8_t* tuples, size_t i, u32 w[16]) {
    const u32* k = reinterpret_cast<const u32*>(tuples + i * STRIDE + OFF);
    SBV_UNROLL
    for (int j = 0; j < 16; ++j) w[j] = j < WORDS ? k[j] : 0u;
}
template <int STRIDE, int OFF, int WORDS>
SBV_HD void group_assign_lane_t(const uint8_t* tuples, size_t i, const GroupState& g, const KeyCache& kc) {
    u32 s = SBV_GROUP_NONE;
    if (g.rep[i] == (u32)i) {
        bool take = g.cnt[i] >= g.min_samples;
        if (!take && kc.enabled) {
            u32 w[16];
            key_words16_t<STRIDE, OFF, WORDS>(tuples, i, w);
            take = key_cache_lookup(kc, w) != SBV_GROUP_NONE;
        }
        if (take) {
            const u32 got = SBV_ATOMIC_ADD(&g.counters[0], 1u);
            if (got < g.max_groups) { s = got; g.group_rep[got] = (u32)i; }
        }
    }
    g.slot_of[i] = s;
}
SBV_HD void group_assign_lane(const uint8_t* tuples, size_t i, const GroupState& g, const KeyCache& kc) {
    group_assign_lane_t<160, 96, 16>(tuples, i, g, kc);
}
SBV_HD void group_assign_lane(size_t i, const GroupState& g) {
    KeyCache off = {};
    group_assign_lane(nullptr, i, g, off);
}

template <int STRIDE, int OFF, int WORDS>
SBV_HD void key_cache_group_key_t(const uint8_t* tuples, const GroupState& g, u32 gidx, u32 w[16]) {
    key_words16_t<STRIDE, OFF, WORDS>(tuples, g.group_rep[gidx], w);
}
SBV_HD void key_cache_group_key(const uint8_t* tuples, const GroupState& g, u32 gidx, u32 w[16]) {
    key_cache_group_key_t<160, 96, 16>(tuples, g, gidx, w);
}
// The two phases every group of a batch goes through (device: k_key_cache_lookup / k_key_cache_insert of
// group_kernels_common.h; the emulator runs them group by group): tslot[k] = table slot of group k, cold[k] = 1 when its
// tables are built in this batch.  A miss with the cache off or full takes slot kc.cap + k of the scheme's pool (the
// per-batch area).
template <int STRIDE, int OFF, int WORDS>
SBV_HD void key_cache_phase_lookup(const uint8_t* tuples, const GroupState& g, const KeyCache& kc, u32 k, u32* tslot, uint8_t* cold) {
    u32 slot = SBV_GROUP_NONE;
    if (kc.enabled) {
        u32 w[16];
        key_cache_group_key_t<STRIDE, OFF, WORDS>(tuples, g, k, w);
        slot = key_cache_lookup(kc, w);
    }
    tslot[k] = slot;
    cold[k] = slot == SBV_GROUP_NONE ? 1 : 0;
}
template <int STRIDE, int OFF, int WORDS>
SBV_HD void key_cache_phase_insert(const uint8_t* tuples, const GroupState& g, const KeyCache& kc, u32 k, u32* tslot) {
    const bool miss = tslot[k] == SBV_GROUP_NONE;
    if (kc.enabled) SBV_ATOMIC_ADD(&kc.count[miss ? 2 : 1], 1u);
    if (!miss) return;
    u32 slot = SBV_GROUP_NONE;
    if (kc.enabled) {
        u32 w[16];
        key_cache_group_key_t<STRIDE, OFF, WORDS>(tuples, g, k, w);
        slot = key_cache_insert(kc, w);
    }
    tslot[k] = slot == SBV_GROUP_NONE ? kc.cap + k : slot;
}

// ---- table classes (round 5) ------------------------------------------------------------------------------------------------
// Every group gets its ROWS (babies and giants of every window: a comb with 4-bit windows, p256_comb29.h: qphase29_point_narrow);
// the FILL — three quarters of a table's cost — is spent only on keys that sign at least `full_min` tuples of this batch, or whose
// cached table was filled by an earlier batch.  full[k] = group k's lanes may take one addition per window; needfill[k] = run the
// fill for it in this batch (a cold key that earns it, or a cached narrow table whose key has become hot: an upgrade — its rows are
// there, only the fill runs).  kfull[slot] remembers the state of a table slot across batches (cache slots; a per-batch slot is
// always cold).  count: the group's exact size from the counting sort, or the sampled count scaled up.
#define SBV_FULL_TABLE_MIN_DEFAULT 256u
SBV_HD void group_table_class_lane(u32 k, const GroupState& g, const u32* tslot, const uint8_t* cold, const uint8_t* kfull, u32 table_slots,
                                   u32 full_min, uint8_t* full, uint8_t* needfill) {
    const u32 count = g.sorted ? g.gcount[k] : g.cnt[g.group_rep[k]] * (g.sample_mask + 1u);
    const u32 slot = tslot[k];
    const bool already = slot < table_slots && !cold[k] && kfull[slot] != 0;
    const bool wants = count >= full_min;
    full[k] = already || wants ? 1 : 0;
    needfill[k] = wants && !already ? 1 : 0;
}
// after every fill of the batch has run: what the slots hold now
SBV_HD void group_table_mark_lane(u32 k, const u32* tslot, const uint8_t* cold, const uint8_t* full, const uint8_t* needfill, u32 table_slots,
                                  uint8_t* kfull) {
    const u32 slot = tslot[k];
    if (slot >= table_slots) return;
    if (cold[k]) kfull[slot] = full[k];
    else if (needfill[k]) kfull[slot] = 1;
}

// ---- hot keys: wide combs in the generic path (round 5) -------------------------------------------------------------
// The registered-key entry gives a consenter's slot a 16- or 20-bit comb (u2 * Q in 16 / 13 additions instead of 32); keys that arrive
// INSIDE generic tuples — client keys of VerifyProposal (internal/bft/view.go:553-559), consenters of a replica that registered
// nothing — never got one.  A cache slot now counts the tuples verified against it (khits) and, once that count passes
// `promote_min` is PROMOTED: a 16-bit comb (17 windows x 32 768 entries, 35.7 MB; HBM holds 288 GB —
// 1024 hot keys are 36.5 GB) is built on the device by the registered path's builder (p256_widetab29.h), whose base points
// B_j = 2^(16 j) Q and C_j = 2^8 B_j ARE entries of the key's 8-bit table (row 2j / 2j + 1, entry 1): no host round trip, at most
// SBV_PROMOTE_MAX keys per batch, behind the batch's verdicts.  Later batches verify a promoted key's tuples from the wide comb in
// one launch that needs no table of the batch at all (p256_group_kernels.hip: the wide pass).  Verdicts cannot depend on it: same
// exact group law, and a slot's comb is a function of its key's 64 bytes.
//   kwide[slot]   wide comb of a cache slot, or SBV_WIDE_NONE           khits[slot]   tuples verified against the slot so far
//   wide[k]       this batch's group k may take the wide pass           hot[0..3]     wide combs handed out | promotions of this batch |
//                                                                                     lanes of the wide pass | (spare)
// Which pass serves a wavefront of the grouped list (p256_group_kernels.hip: k_verify_keyed_q<MODE>), from three wave-level facts.  A lane
// whose key is no point (or has no slot) is dead; a key with a wide comb keeps its 8-bit table, full OR rows only — promotion does not
// ask which —, so in a mixed wavefront it counts as what that table is.  One function for the kernels and the emulator: in round 5 the
// two copies of this rule shared a bug.
#define SBV_Q_FULL 0
#define SBV_Q_NARROW 1
#define SBV_Q_WIDE 2
#define SBV_Q_NONE 3
SBV_HD int group_wave_class(bool all_dead, bool all_dead_or_wide, bool all_dead_or_full) {
    if (all_dead) return SBV_Q_NONE;
    if (all_dead_or_wide) return SBV_Q_WIDE;
    if (all_dead_or_full) return SBV_Q_FULL;
    return SBV_Q_NARROW;
}
#define SBV_PROMOTE_MAX 64u
#define SBV_HOT_BITS 16
#define SBV_HOT_HITS_MAX 0x3FFFFFFFu
SBV_HD void group_hot_class_lane(u32 k, const GroupState& g, const u32* tslot, const uint8_t* cold, u32 cache_cap, const u32* kwide, u32* khits,
                                 uint8_t* wide) {
    const u32 slot = tslot[k];
    bool w = false;
    if (slot < cache_cap) {
        const u32 count = g.sorted ? g.gcount[k] : g.cnt[g.group_rep[k]] * (g.sample_mask + 1u);
        const u32 h = khits[slot];                              // one group per slot and batch: no atomics
        khits[slot] = h > SBV_HOT_HITS_MAX - count ? SBV_HOT_HITS_MAX : h + count;
        w = !cold[k] && kwide[slot] != 0xFFFFFFFFu;
    }
    wide[k] = w ? 1 : 0;
}
// ---- life cycle of the hot keys (round 6) --------------------------------------------------------------------
// Round 5 never forgot a count and never took a comb back: 4 096 signatures under each of 1 024 throw-away keys filled the pool for
// good, and a long-running node kept the combs of yesterday's clients (signer sets change on reconfiguration:
// pkg/consensus/consensus.go:185-252).  Now:
//   * every SBV_HOT_DECAY_EVERY-th grouped batch halves every count (a clock sweep over the cache slots): a key must keep signing
//     ~promote_min / 2 DECAY_EVERY tuples per batch to stay above the threshold, a key that stopped falls below it within
//     DECAY_EVERY * log2(count / promote_min) batches;
//   * when the pool is full, a slot that has earned a comb takes the comb of the owner with the LOWEST count — if that count is at
//     most half its own (hysteresis: two keys of similar heat never trade a 35.7 MB comb back and forth); wowner[w] = the slot
//     that owns comb w.  The victim keeps its 8-bit table and is served from it again from the next batch on.
// A slot's count grows by the tuples GROUPED under it, whatever their verdicts.  Counting accepted tuples only (ADVICE r5's other
// option) was built and measured first: it costs a pass over the grouped list per batch and buys nothing — whoever can send 4 096
// garbage signatures under a key of his own can as well send 4 096 valid ones; what bounds the damage is that combs follow the heat
// (decay + eviction), not that they are hard to earn — and it left the signers under which the synthetic batch of SURVEY 8d puts all
// its corrupted signatures (1/8 of the keys) on their 8-bit tables for ever (one PCIe caller 270 -> 197 M/s, profiles/r06).
// Counts saturate far below 2^32.  Verdicts never depend on any of this (a comb is a function of its key's 64 bytes).
#define SBV_HOT_DECAY_EVERY 16u
SBV_HD void hot_decay_lane(u32 slot, u32* khits) { khits[slot] >>= 1; }
// Eviction, one candidate at a time.  hot_evict_scan: lane `lane` of `lanes` looks at owners w = lane, lane + lanes, ... that were not
// handed out in this batch (taken: one bit per comb) and keeps the coldest (lowest count, then lowest index: deterministic across any
// number of lanes); hot_evict_better merges two lanes' findings; hot_evict_ok is the hysteresis.
SBV_HD bool hot_evict_better(u32 h, u32 w, u32 best_h, u32 best_w) { return h < best_h || (h == best_h && w < best_w); }
SBV_HD void hot_evict_scan(const u32* khits, const u32* wowner, const u32* taken, u32 wide_cap, u32 cache_cap, u32 lane, u32 lanes, u32& best_h, u32& best_w) {
    best_h = 0xFFFFFFFFu; best_w = 0xFFFFFFFFu;
    for (u32 w = lane; w < wide_cap; w += lanes) {
        if ((taken[w >> 5] >> (w & 31)) & 1u) continue;
        const u32 owner = wowner[w];
        if (owner >= cache_cap) continue;                       // never handed out (cannot happen with a full pool) or owner forgotten
        const u32 h = khits[owner];
        if (hot_evict_better(h, w, best_h, best_w)) { best_h = h; best_w = w; }
    }
}
SBV_HD bool hot_evict_ok(u32 cand_hits, u32 victim_hits) { return victim_hits <= cand_hits / 2; }
// end of the batch: group k asks for a wide comb when its slot is cached, valid (its 8-bit rows, full or not, hold the builder's base
// points), hot and has none yet.  plist[2 i] = slot, plist[2 i + 1] = wide index of promotion i.
// elist: the slots that found the pool full, candidates of this batch's evictions (hot[3] counts them; round 6)
SBV_HD void group_promote_select_lane(u32 k, const u32* tslot, const uint8_t* kvalid, u32 cache_cap, const u32* kwide,
                                      const u32* khits, u32 promote_min, u32 wide_cap, u32* hot, u32* plist, u32* elist) {
    const u32 slot = tslot[k];
    if (slot >= cache_cap || !kvalid[slot] || kwide[slot] != 0xFFFFFFFFu || khits[slot] < promote_min) return;
    if (hot[0] >= wide_cap) {                                   // the pool is full (the counter only grows: a stale read costs one atomic)
        const u32 e = SBV_ATOMIC_ADD(&hot[3], 1u);
        if (e < SBV_PROMOTE_MAX) elist[e] = slot;
        return;
    }
    const u32 i = SBV_ATOMIC_ADD(&hot[1], 1u);
    if (i >= SBV_PROMOTE_MAX) return;                           // this batch's quota: the key is asked again by the next batch
    const u32 w = SBV_ATOMIC_ADD(&hot[0], 1u);
    if (w >= wide_cap) {                                        // past the pool's end: an empty entry, and the slot becomes an eviction candidate
        const u32 e = SBV_ATOMIC_ADD(&hot[3], 1u);
        if (e < SBV_PROMOTE_MAX) elist[e] = slot;
    }
    plist[2 * i] = w < wide_cap ? slot : 0xFFFFFFFFu;
    plist[2 * i + 1] = w;
}
// The evictions of a batch, by ONE agent (a workgroup's lane 0 on the device after its lanes' scans were merged; the emulator alone):
// candidate c of `cands` gets comb best_w when hot_evict_ok; plist entries are appended behind the batch's ordinary promotions.
// Returns the new number of plist entries.
SBV_HD u32 hot_evict_commit(u32 cand_slot, u32 best_h, u32 best_w, u32* khits, u32* kwide, u32* wowner, u32* taken, u32 entries, u32* plist) {
    if (best_w == 0xFFFFFFFFu || entries >= SBV_PROMOTE_MAX || !hot_evict_ok(khits[cand_slot], best_h)) return entries;
    kwide[wowner[best_w]] = 0xFFFFFFFFu;                        // the victim is served from its 8-bit table again
    wowner[best_w] = cand_slot;
    taken[best_w >> 5] |= 1u << (best_w & 31);
    plist[2 * entries] = cand_slot;
    plist[2 * entries + 1] = best_w;
    return entries + 1;
}

// ---- per-batch key tables ----------------------------------------------------------------------------
// jbases: [groups][33] Jacobian 2^(8j) * Q with cached Z^2, Z^3 (qent layout, 40 dwords);
// valid[g] = pointFromAffine verdict.  One call produces bases j_first..j_last; a call with j_first > 0
// continues the doubling chain from base j_first - 1 left by the previous chunk.
#define SBV_JBASE_DWORDS 40

// One part of one (key, window): row[k-1] = k * base for k = part*E + 1 .. part*E + E, affine.
// Several lanes per window instead of one: the lane first reaches (part*E) * base with 7 doubling /
// conditional-addition steps, then walks its E entries; each lane normalises its own E points
// (Montgomery's trick + one inversion).  `tmp` = private scratch of E * (24 + 8) dwords.
// Every addition is exact (the second entry of part 0 is base + base: the doubling branch).
#define SBV_KEYTAB_PARTS_DEFAULT 4
#define SBV_KEYTAB_TMP_DWORDS_PER_WINDOW (SBV_GTAB_PER_WINDOW * 32)

}  // namespace sbv
