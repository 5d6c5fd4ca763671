// p256_group.h — grouping a batch of generic tuples by public key, inside the step.
//
// Signers repeat: BASELINE.json's headline batch has 1024 distinct keys in 2^20 tuples; SmartBFT's
// consenters are a handful.  For a key used by many tuples of ONE batch it pays to
// build that key's comb table once (about 25 generic verifications' worth of work) and verify all
// its signatures with the registered-key kernel (50 mixed additions, no doublings) instead of
// paying 256 doublings per signature.  Everything here happens inside every call — nothing is
// remembered from one batch to the next, and verdicts are identical to the generic path:
//
//   group_insert   every tuple inserts its 64-byte key into an open-addressing hash table in HBM
//                  (32-bit entries = representative tuple index + 1, claimed with atomicCAS; a hash
//                  hit is only trusted after comparing all 64 key bytes, so collisions cost a probe,
//                  never a wrong group) and counts itself on its representative
//   group_assign   representatives with >= min_count users take a table slot (atomic counter)
//   group_split    tuples are compacted into a "grouped" and an "ungrouped" index list
//   keytab_bases   per grouped key: validate it (pointFromAffine rules), 2^(8j) * Q for j = 0..32
//   keytab_window  per (key, window): the 128 affine multiples, Montgomery-trick normalised
// then k_p256_verify_keyed runs over the grouped list and k_p256_verify over the ungrouped one.
//
// Shared host/device source (tests/emul runs the same functions sequentially).
#pragma once
#include "p256_core.h"

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

struct GroupState {
    u32* ht;          // hash table, ht_mask + 1 entries, zeroed before every batch
    u32 ht_mask;
    u32* rep;         // [n] representative tuple of tuple i's key
    u32* cnt;         // [n] users of a representative (zeroed before every batch)
    u32* slot_of;     // [n] table slot of a representative, or NONE
    u32* group_rep;   // [max_groups] representative tuple of slot g
    u32* counters;    // [0] groups handed out, [1] grouped tuples, [2] ungrouped tuples (zeroed)
    u32* grp_idx;     // [n] compacted grouped tuple indices
    u32* ung_idx;     // [n] compacted ungrouped tuple indices
    u32* slots;       // [n] table slot per tuple (grouped ones)
    u32 max_groups;
    u32 min_count;
};

SBV_HD const u32* tuple_key_words(const uint8_t* tuples, size_t i) {
    return reinterpret_cast<const u32*>(tuples + i * 160 + 96);       // 16-byte aligned: 160 i + 96
}

SBV_HD void group_insert_lane(const uint8_t* tuples, size_t i, const GroupState& g) {
    const u32* k = tuple_key_words(tuples, i);
    u32 w[16];
    SBV_UNROLL
    for (int j = 0; j < 16; ++j) w[j] = k[j];
    u32 h = w[0] * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + w[3] * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + w[7] * 0xC2B2AE3Du;
    h = (h ^ (h >> 16)) + w[8] * 0x27D4EB2Fu;
    h = (h ^ (h >> 15)) + w[12] * 0x165667B1u;
    h = (h ^ (h >> 13)) + w[15] * 0x9E3779B1u;
    h ^= h >> 16;
    u32 slot = h & g.ht_mask;
    u32 mine = (u32)i;
    for (u32 probes = 0; probes <= g.ht_mask; ++probes) {
        u32 v = g.ht[slot];
        if (v == 0) v = SBV_ATOMIC_CAS(&g.ht[slot], 0u, (u32)i + 1u);
        if (v == 0) break;                                   // claimed: this tuple represents its key
        const u32* o = tuple_key_words(tuples, v - 1);
        u32 diff = 0;
        SBV_UNROLL
        for (int j = 0; j < 16; ++j) diff |= o[j] ^ w[j];
        if (diff == 0) { mine = v - 1; break; }
        slot = (slot + 1) & g.ht_mask;
    }
    g.rep[i] = mine;
    SBV_ATOMIC_ADD(&g.cnt[mine], 1u);
}

SBV_HD void group_assign_lane(size_t i, const GroupState& g) {
    u32 s = SBV_GROUP_NONE;
    if (g.rep[i] == (u32)i && g.cnt[i] >= g.min_count) {
        const u32 got = SBV_ATOMIC_ADD(&g.counters[0], 1u);
        if (got < g.max_groups) { s = got; g.group_rep[got] = (u32)i; }
    }
    g.slot_of[i] = s;
}

SBV_HD void group_split_lane(size_t i, const GroupState& g) {
    const u32 s = g.slot_of[g.rep[i]];
    if (s == SBV_GROUP_NONE) {
        g.ung_idx[SBV_ATOMIC_ADD(&g.counters[2], 1u)] = (u32)i;
    } else {
        g.slots[i] = s;
        g.grp_idx[SBV_ATOMIC_ADD(&g.counters[1], 1u)] = (u32)i;
    }
}

// ---- per-batch key tables ----------------------------------------------------------------------------
// bases: [groups][33] affine 2^(8j) * Q (Montgomery form); valid[g] = pointFromAffine verdict.
SBV_HD void keytab_bases_lane(const uint8_t* tuples, u32 gidx, const GroupState& g, apt* bases, uint8_t* valid) {
    const u32* k = tuple_key_words(tuples, g.group_rep[gidx]);
    u256 qx, qy;
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) { qx.v[l] = bswap32(k[7 - l]); qy.v[l] = bswap32(k[8 + 7 - l]); }
    const fe p_ = fe_p();
    apt q;
    fe_to_mont(q.x, qx);
    fe_to_mont(q.y, qy);
    const bool ok = lt256(qx, p_) && lt256(qy, p_) && pt_on_curve(q.x, q.y);
    valid[gidx] = ok ? 1 : 0;
    apt* out = bases + (size_t)gidx * SBV_GTAB_WINDOWS;
    // Jacobian chain of doublings; the 33 bases are normalised together (one inversion).  The Jacobian
    // bases are parked in the output rows' memory: X -> out[j].x, Y -> out[j].y, Z in a side array.
    jpt t;
    t.X = q.x; t.Y = q.y; t.Z = fe_one();
    fe zs[SBV_GTAB_WINDOWS];
    fe pre[SBV_GTAB_WINDOWS];
    fe acc = fe_one();
    for (int j = 0; j < SBV_GTAB_WINDOWS; ++j) {
        out[j].x = t.X; out[j].y = t.Y; zs[j] = t.Z;
        pre[j] = acc;
        fe_mul(acc, acc, t.Z);
        if (j + 1 < SBV_GTAB_WINDOWS) {
            SBV_NOUNROLL
            for (int d = 0; d < 8; ++d) pt_dbl(t, t);
        }
    }
    fe inv;
    fe_inv(inv, acc);                  // garbage in, garbage out for an invalid key (never used: valid = 0)
    for (int j = SBV_GTAB_WINDOWS - 1; j >= 0; --j) {
        fe zi, zi2, zi3;
        fe_mul(zi, inv, pre[j]);
        fe_mul(inv, inv, zs[j]);
        fe_sqr(zi2, zi);
        fe_mul(zi3, zi2, zi);
        fe_mul(out[j].x, out[j].x, zi2);
        fe_mul(out[j].y, out[j].y, zi3);
    }
}

// One quarter of one (key, window): row[k-1] = k * base for k = part*32 + 1 .. part*32 + 32, affine.
// Four lanes per window instead of one: the lane first reaches (part*32) * base with <= 7 doublings
// and <= 2 additions, then walks its 32 entries; each lane normalises its own 32 points (Montgomery's
// trick + one inversion).  `tmp` = private scratch of 32 * (24 + 8) dwords.
#define SBV_KEYTAB_PARTS 4
#define SBV_KEYTAB_PART_ENTRIES (SBV_GTAB_PER_WINDOW / SBV_KEYTAB_PARTS)
#define SBV_KEYTAB_TMP_DWORDS (SBV_KEYTAB_PART_ENTRIES * 32)
SBV_HD void keytab_window_lane(const apt& base, int part, u32* tmp, apt* row) {
    constexpr int E = SBV_KEYTAB_PART_ENTRIES;
    u32* pts = tmp;                   // E * 24 dwords
    u32* pre = tmp + E * 24;          // E * 8 dwords
    jpt t;
    pt_set_inf(t);
    const int m = part * E;           // start multiple: 0, 32, 64, 96
    SBV_NOUNROLL
    for (int bit = 6; bit >= 0; --bit) {
        pt_dbl(t, t);                                         // infinity stays infinity
        pt_add_mixed(t, base, false, ((m >> bit) & 1) == 0);  // exact: handles t = infinity
    }
    fe acc = fe_one();
    SBV_NOUNROLL
    for (int k = 0; k < E; ++k) {
        pt_add_mixed(t, base, false, false);                  // (m + k + 1) * base; never hits P == +-Q for a valid key
        fe_store16(pts + k * 24, t.X); fe_store16(pts + k * 24 + 8, t.Y); fe_store16(pts + k * 24 + 16, t.Z);
        fe_store16(pre + k * 8, acc);
        fe_mul(acc, acc, t.Z);
    }
    fe inv;
    fe_inv(inv, acc);
    SBV_NOUNROLL
    for (int k = E - 1; k >= 0; --k) {
        fe X, Y, Z, pk, zi, zi2, zi3;
        fe_load16(X, pts + k * 24); fe_load16(Y, pts + k * 24 + 8); fe_load16(Z, pts + k * 24 + 16);
        fe_load16(pk, pre + k * 8);
        fe_mul(zi, inv, pk);
        fe_mul(inv, inv, Z);
        fe_sqr(zi2, zi);
        fe_mul(zi3, zi2, zi);
        apt a;
        fe_mul(a.x, X, zi2);
        fe_mul(a.y, Y, zi3);
        fe_store16(reinterpret_cast<u32*>(row + m + k), a.x);
        fe_store16(reinterpret_cast<u32*>(row + m + k) + 8, a.y);
    }
}

}  // namespace sbv
