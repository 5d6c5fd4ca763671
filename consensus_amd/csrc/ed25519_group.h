// ed25519_group.h — the in-step key grouping of p256_group.h for the Ed25519 variant (BASELINE.json configs[4]).
//
// Same idea, simpler group law: signers repeat inside a batch, so for every public key A used often enough
// the batch builds the 32 x 128 affine-Niels comb of -A once and then every signature of that key costs
//     [S]B  (32 comb additions, the fixed table)  +  [k](-A)  (32 comb additions, the per-batch table)
// instead of 256 doublings + 96 additions.  The unified Edwards addition is complete, so nothing here has
// exceptional cases.  Grouping state, sampling, slot assignment: GroupState / group_assign_lane of
// p256_group.h; the key is the 32-byte A_enc at byte 64 of the 128-byte tuple R | S | A | k.
//
//   ed_group_split      compaction only (no key filter: see below)
//   ed_keytab_bases     per grouped key: decompress, negate, 2^(8j) * (-A), j = 0..31 (extended coordinates)
//   ed_keytab_window    per (key, window, part): the affine-Niels multiples, Montgomery-trick normalised
//   ed_gphase           [S]B for every tuple + the S < L, k < L checks
//   ed_qphase           += [k](-A) from the key's comb, windows [j0, j1); the last chunk encodes and compares
//
// Shared host/device source (tests/emul runs the same functions sequentially).
#pragma once
#include "ed25519_core.h"
#include "p256_group.h"

namespace sbv {

#define SBV_ED_KEY_WINDOWS 32              // k < L < 2^253: 32 signed 8-bit digits
#define SBV_ED_KEY_PER_WINDOW 128
#define SBV_ED_KEYTAB_ENTRIES (SBV_ED_KEY_WINDOWS * SBV_ED_KEY_PER_WINDOW)
#define SBV_ED_JBASE_DWORDS 32             // one extended point

SBV_HD const u32* ed_tuple_words(const uint8_t* tuples, size_t i) { return reinterpret_cast<const u32*>(tuples + i * 128); }

SBV_HD void ed_group_insert_lane(const uint8_t* tuples, size_t i, const GroupState& g) { group_insert_lane_t<128, 64, 8>(tuples, i, g); }

// Decompress tuple idx's public key; false = crypto/ed25519 would refuse it (Point.SetBytes fails)
SBV_HD bool ed_tuple_key_load(const uint8_t* tuples, size_t idx, ept& A) {
    const u32* w = ed_tuple_words(tuples, idx) + 16;
    u32 pk[8];
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) pk[j] = w[j];
    return ed_decompress(A, pk);
}

// Unlike the P-256 split there is no key filter here: decompressing A is a ~270-multiplication square root, and
// because nearly every wavefront holds at least one ungrouped lane the filter made the split cost as much as the
// whole [S]B phase (1.1 ms, measured) — while about half of all random 32-byte strings ARE points, so it could
// not empty the ungrouped list anyway.  Undecompressable keys are rejected by the one-lane kernel.
SBV_HD bool ed_group_split_lane(size_t i, const GroupState& g) {
    const u32 s = g.slot_of[g.rep[i]];
    if (s == SBV_GROUP_NONE) {
        g.ung_idx[SBV_ATOMIC_ADD(&g.counters[2], 1u)] = (u32)i;
    } else {
        g.slots[i] = s;
        g.grp_idx[SBV_ATOMIC_ADD(&g.counters[1], 1u)] = (u32)i;
    }
    return true;
}

SBV_HD void ept_store(u32* dst, const ept& p) {
    fe_store16(dst, p.X); fe_store16(dst + 8, p.Y); fe_store16(dst + 16, p.Z); fe_store16(dst + 24, p.T);
}
SBV_HD void ept_load(ept& p, const u32* src) {
    fe_load16(p.X, src); fe_load16(p.Y, src + 8); fe_load16(p.Z, src + 16); fe_load16(p.T, src + 24);
}

// jbases: [groups][32] extended points 2^(8j) * (-A); valid[g] = the key decompressed.  One call produces
// bases j_first..j_last; a call with j_first > 0 continues the doubling chain from base j_first - 1.
SBV_HD void ed_keytab_bases_lane(const uint8_t* tuples, u32 gidx, const GroupState& g, u32* jbases, uint8_t* valid,
                                 int j_first, int j_last) {
    u32* out = jbases + (size_t)gidx * (SBV_ED_KEY_WINDOWS * SBV_ED_JBASE_DWORDS);
    ept t;
    if (j_first == 0) {
        ept A;
        const bool ok = ed_tuple_key_load(tuples, g.group_rep[gidx], A);
        valid[gidx] = ok ? 1 : 0;         // an invalid key still gets a (garbage) table; it is never used
        t = A;
        fe25_neg(t.X, A.X);
        fe25_neg(t.T, A.T);
    } else {
        ept_load(t, out + (size_t)(j_first - 1) * SBV_ED_JBASE_DWORDS);
    }
    SBV_NOUNROLL
    for (int j = j_first; j <= j_last; ++j) {
        if (j > 0) {
            SBV_NOUNROLL
            for (int d = 0; d < 8; ++d) ed_dbl(t, t);
        }
        ept_store(out + (size_t)j * SBV_ED_JBASE_DWORDS, t);
    }
}

// One part of one (key, window): row[k-1] = k * base for k = part*E + 1 .. part*E + E as affine-Niels points.
// `tmp` = private scratch of E * (24 + 8) dwords (X, Y, Z and the running product of the Zs).
SBV_HD void ed_keytab_window_lane(const u32* jbase, int part, int parts, u32* tmp, aniels* row) {
    const int E = SBV_ED_KEY_PER_WINDOW / parts;      // parts is a power of two <= 16
    u32* pts = tmp;                   // E * 24 dwords
    u32* pre = tmp + E * 24;          // E * 8 dwords
    ept b;
    ept_load(b, jbase);
    pniels base;
    ed_to_pniels(base, b);
    ept t;
    ed_set_ident(t);
    const int m = part * E;           // start multiple
    SBV_NOUNROLL
    for (int bit = 6; bit >= 0; --bit) {
        ed_dbl(t, t);
        ed_add_pniels(t, base, false, ((m >> bit) & 1) == 0);
    }
    fe25 acc = fe25_one();
    SBV_NOUNROLL
    for (int k = 0; k < E; ++k) {
        ed_add_pniels(t, base, false, false);                 // (m + k + 1) * base
        fe_store16(pts + k * 24, t.X); fe_store16(pts + k * 24 + 8, t.Y); fe_store16(pts + k * 24 + 16, t.Z);
        fe_store16(pre + k * 8, acc);
        fe25_mul(acc, acc, t.Z);
    }
    fe25 inv;
    fe25_inv_gcd(inv, acc);           // Z is never 0 on a complete curve; an invalid key's garbage is never used
    const fe25 d2 = fe25_2d();
    SBV_NOUNROLL
    for (int k = E - 1; k >= 0; --k) {
        fe25 X, Y, Z, pk, zi, x, y;
        fe_load16(X, pts + k * 24); fe_load16(Y, pts + k * 24 + 8); fe_load16(Z, pts + k * 24 + 16);
        fe_load16(pk, pre + k * 8);
        fe25_mul(zi, inv, pk);
        fe25_mul(inv, inv, Z);
        fe25_mul(x, X, zi);
        fe25_mul(y, Y, zi);
        aniels a;
        fe25_add(a.ypx, y, x);
        fe25_sub(a.ymx, y, x);
        fe25_mul(a.xy2d, x, y);
        fe25_mul(a.xy2d, a.xy2d, d2);
        u32* dst = reinterpret_cast<u32*>(row + m + k);
        fe_store16(dst, a.ypx); fe_store16(dst + 8, a.ymx); fe_store16(dst + 16, a.xy2d);
    }
}

// gacc: 32 words per tuple (X, Y, Z, T limbs), limb-major: word w of tuple i at gacc[w * cap + i]
SBV_HD void ed_gacc_store(u32* gacc, size_t cap, size_t i, const ept& R) {
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) {
        gacc[(size_t)l * cap + i] = R.X.v[l];
        gacc[(size_t)(8 + l) * cap + i] = R.Y.v[l];
        gacc[(size_t)(16 + l) * cap + i] = R.Z.v[l];
        gacc[(size_t)(24 + l) * cap + i] = R.T.v[l];
    }
}
SBV_HD void ed_gacc_load(ept& R, const u32* gacc, size_t cap, size_t i) {
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) {
        R.X.v[l] = gacc[(size_t)l * cap + i];
        R.Y.v[l] = gacc[(size_t)(8 + l) * cap + i];
        R.Z.v[l] = gacc[(size_t)(16 + l) * cap + i];
        R.T.v[l] = gacc[(size_t)(24 + l) * cap + i];
    }
}

// [S]B (16-bit comb `btab`) for tuple i -> gacc; okb[i] = S < L and k < L (SetCanonicalBytes(S); a reduced k is always < L)
SBV_HD void ed_gphase_lane(const uint8_t* tuples, size_t i, const aniels* btab, u32* gacc, size_t cap, uint8_t* okb) {
    const u32* w = ed_tuple_words(tuples, i);
    u256 S, k;
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) { S.v[j] = w[8 + j]; k.v[j] = w[24 + j]; }
    const u256 L = ed_L();
    okb[i] = (lt256(S, L) && lt256(k, L)) ? 1 : 0;
    ept R;
    ed_set_ident(R);
    ed_add_sB(R, S, btab);
    ed_gacc_store(gacc, cap, i, R);
}

// encode(R) == R_enc (the tuple's first 8 dwords), byte for byte
SBV_HD bool ed_encoding_matches(const ept& R, const u32* renc) {
    fe25 zi, x, y;
    fe25_inv_gcd(zi, R.Z);
    fe25_mul(x, R.X, zi);
    fe25_mul(y, R.Y, zi);
    fe25_freeze(y, y);
    y.v[7] |= (fe25_is_negative(x) ? 1u : 0u) << 31;
    u32 diff = 0;
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) diff |= y.v[j] ^ renc[j];
    return diff == 0;
}

// R (from gacc) += windows [j0, j1) of [k](-A) from the key's comb.  `last` -> the verdict is returned; otherwise
// R goes back to gacc and the return value is meaningless.
SBV_HD bool ed_qphase_lane(const uint8_t* tuples, size_t i, u32 slot, u32 nkeys, const aniels* ktab, const uint8_t* kvalid,
                           u32* gacc, size_t cap, const uint8_t* okb, int j0, int j1, bool last) {
    const u32* w = ed_tuple_words(tuples, i);
    u256 k, kk;
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) k.v[j] = w[24 + j];
    bool ok = okb[i] != 0 && slot < nkeys;
    if (slot >= nkeys) slot = 0;
    ok = ok && kvalid[slot] != 0;
    const aniels* tab = ktab + (size_t)slot * SBV_ED_KEYTAB_ENTRIES;
    (void)add_const_limbs(kk, k, 0x80808080u);       // k >= L was rejected above; k < 2^253 cannot carry out
    ept R;
    ed_gacc_load(R, gacc, cap, i);
    SBV_NOUNROLL
    for (int j = j0; j < j1; ++j) {
        const int d = (int)((kk.v[j >> 2] >> ((j & 3) * 8)) & 255u) - 128;
        const int ad = d < 0 ? -d : d;
        const u32* bp = reinterpret_cast<const u32*>(tab + (size_t)j * SBV_ED_KEY_PER_WINDOW + (ad == 0 ? 0 : ad - 1));
        aniels e;
        fe_load16(e.ypx, bp); fe_load16(e.ymx, bp + 8); fe_load16(e.xy2d, bp + 16);
        ed_add_aniels(R, e, d < 0, d == 0);
    }
    if (!last) { ed_gacc_store(gacc, cap, i, R); return false; }
    u32 renc[8];
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) renc[j] = w[j];
    return ok && ed_encoding_matches(R, renc);
}

// tuple words straight from HBM (the ungrouped list is sparse: no LDS staging)
struct EdGlobalTuple {
    const u32* p;
    SBV_HD u32 operator[](int i) const { return p[i]; }
};

}  // namespace sbv
