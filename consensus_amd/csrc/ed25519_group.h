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
//   ed_qphase           += [k](-A) from the key's comb, windows [j0, j1); the last chunk marks the tuples still pending
//   ed_finish           encode(R) == R_enc for the pending tuples, one inversion per 8 tuples
//
// Shared host/device source (tests/emul runs the same functions sequentially).
#pragma once
#include "ed25519_core.h"
#include "p256_group.h"

namespace sbv {

#define SBV_ED_KEY_WINDOWS 32              // k < L < 2^253: 32 signed 8-bit digits
#define SBV_ED_KEY_PER_WINDOW 128
#define SBV_ED_KEYTAB_ENTRIES (SBV_ED_KEY_WINDOWS * SBV_ED_KEY_PER_WINDOW)
#define SBV_ED_JBASE_DWORDS SBV_ED_PT_WORDS   // one extended point, raw limbs

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

// key-sorted step, pass 1: classification only.  The ungrouped tuples become CANDIDATES (ung_cand, counters[4]); the grouped list
// is built by the counting sort of p256_group.h.
SBV_HD void ed_group_classify_lane(size_t i, const GroupState& g) {
    const u32 s = g.slot_of[g.rep[i]];
    g.slots[i] = s;
    if (s == SBV_GROUP_NONE) g.ung_cand[SBV_ATOMIC_ADD(&g.counters[4], 1u)] = (u32)i;
}
// pass 2, over the compacted candidates only (round 4; the P-256 step's k_group_keycheck on this curve): a candidate whose key does
// not decompress is rejected here, the others form the ungrouped list of the one-lane kernel.  About half of all 32-byte strings are
// no points, and an ungrouped tuple is nearly always one whose key a bit flip hit: without this pass every wavefront of the one-lane
// kernel (a ~460 000-instruction chain) ran to its end for the half of its lanes that had left after the square root.  The check
// costs that square root once more (~20 000 instructions) on 4 % of the batch's lanes; in ONE pass over all tuples it cost as much
// as the whole [S]B phase (see ed_group_split_lane), which is why it waited for the compaction.
// ungxy (round 6, may be null): X | Y of the accepted keys, 20 raw limbs per position of the list, for the quad form of the one-lane kernel
SBV_HD void ed_group_keycheck_lane(const uint8_t* tuples, u32 L, const GroupState& g, uint8_t* acc, u32* ungxy = nullptr) {
    const u32 i = g.ung_cand[L];
    ept A;
    if (ed_tuple_key_load(tuples, i, A)) {
        const u32 pos = SBV_ATOMIC_ADD(&g.counters[2], 1u);
        g.ung_idx[pos] = i;
        if (ungxy) { fe25_store_raw(ungxy + (size_t)pos * SBV_ED_UNGXY_WORDS, A.X); fe25_store_raw(ungxy + (size_t)pos * SBV_ED_UNGXY_WORDS + 10, A.Y); }
    } else {
        acc[i] = 0;
        SBV_ATOMIC_ADD(&g.counters[3], 1u);
    }
}

SBV_HD void ept_store(u32* dst, const ept& p) {
    fe25_store_raw(dst, p.X); fe25_store_raw(dst + 10, p.Y); fe25_store_raw(dst + 20, p.Z); fe25_store_raw(dst + 30, p.T);
}
SBV_HD void ept_load(ept& p, const u32* src) {
    fe25_load_raw(p.X, src); fe25_load_raw(p.Y, src + 10); fe25_load_raw(p.Z, src + 20); fe25_load_raw(p.T, src + 30);
}

// jbases: [groups][32] extended points 2^(8j) * (-A); *valid_of_slot (the byte of the group's table slot) = the key decompressed.  One call produces
// bases j_first..j_last; a call with j_first > 0 continues the doubling chain from base j_first - 1.
SBV_HD void ed_keytab_bases_lane(const uint8_t* tuples, u32 gidx, const GroupState& g, u32* jbases, uint8_t* valid_of_slot,
                                 int j_first, int j_last) {
    u32* out = jbases + (size_t)gidx * (SBV_ED_KEY_WINDOWS * SBV_ED_JBASE_DWORDS);
    ept t;
    if (j_first == 0) {
        ept A;
        const bool ok = ed_tuple_key_load(tuples, g.group_rep[gidx], A);
        *valid_of_slot = ok ? 1 : 0;      // an invalid key still gets a (garbage) table; it is never used
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

// ---- the same chain on FOUR lanes per key (round 5; VERDICT r4 #5: "quad-lane base chain as P-256 got") ------------------------
// ed_dbl is 4 squarings, then 4 products of their sums: two levels deep.  The four lanes of a quad hold the whole point; at level 1
// lane r squares X / Y / Z / X + Y, the squares travel by DPP quad_perm broadcasts (full-rate register moves, no LDS), every lane
// forms e, f, g, h, at level 2 lane r multiplies f e / h g / f g / e h = X3 / Y3 / Z3 / T3, and the products are broadcast again:
// 2 multiplications deep per doubling instead of 8 in a row.  Operation for operation what ed_dbl does on the same operands, so the
// recorded bases are BYTE FOR BYTE those of ed_keytab_bases_lane (tests/test_ed25519_cpu.py compares them); the limb bounds are
// ed_dbl's.  The exchange policy is a template parameter as in p256_keytab29.h: one lane + DPP on the device
// (ed25519_group_kernels.hip), four lanes stepped in lockstep in tests/emul.
struct edchain_quad_host {
    static const int N = 4;
    ept s[4];
    int role(int i) const { return i; }
    void bcast(fe25 out[4], const fe25 in[4], int src) const { for (int i = 0; i < 4; ++i) out[i] = in[src]; }
};
SBV_HD void edchain_l1(fe25& P, const ept& s, int role) {
    fe25 xy, in;
    fe25_add(xy, s.X, s.Y);
    fe25_select(in, role == 0, s.X, xy);
    fe25_select(in, role == 1, s.Y, in);
    fe25_select(in, role == 2, s.Z, in);
    fe25_sqr(P, in);
}
SBV_HD void edchain_l2(fe25& P, const fe25& xx, const fe25& yy, const fe25& zz, const fe25& xy2, int role) {
    fe25 zz2, e, g, h, f, a, b;
    fe25_add(zz2, zz, zz);
    fe25_add(h, yy, xx);
    fe25_sub(g, yy, xx);
    fe25_sub(e, xy2, h);
    fe25_sub(f, zz2, g);
    fe25_select(a, role == 1, h, f);           // first operand:  f | h | f | e     (ed_dbl's operand order: the bounds are per operand)
    fe25_select(a, role == 3, e, a);
    fe25_select(b, role == 0, e, g);           // second operand: e | g | g | h
    fe25_select(b, role == 3, h, b);
    fe25_mul(P, a, b);
}
template <class QX>
SBV_HD void edchain_dbl(QX& q) {
    fe25 P[QX::N], xx[QX::N], yy[QX::N], zz[QX::N], xy2[QX::N];
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) edchain_l1(P[i], q.s[i], q.role(i));
    q.bcast(xx, P, 0); q.bcast(yy, P, 1); q.bcast(zz, P, 2); q.bcast(xy2, P, 3);
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) edchain_l2(P[i], xx[i], yy[i], zz[i], xy2[i], q.role(i));
    q.bcast(xx, P, 0); q.bcast(yy, P, 1); q.bcast(zz, P, 2); q.bcast(xy2, P, 3);
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) { q.s[i].X = xx[i]; q.s[i].Y = yy[i]; q.s[i].Z = zz[i]; q.s[i].T = xy2[i]; }
}
// the whole chain of one chunk for one quad: what ed_keytab_bases_lane does, lane `role` storing coordinate `role` of every base
template <class QX>
SBV_HD void edchain_run(QX& q, const uint8_t* tuples, u32 gidx, const GroupState& g, u32* jbases, uint8_t* valid_of_slot, int j_first, int j_last) {
    u32* out = jbases + (size_t)gidx * (SBV_ED_KEY_WINDOWS * SBV_ED_JBASE_DWORDS);
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) {
        if (j_first == 0) {
            ept A;
            const bool ok = ed_tuple_key_load(tuples, g.group_rep[gidx], A);     // every lane of the quad: same key, same verdict
            if (q.role(i) == 0) *valid_of_slot = ok ? 1 : 0;
            q.s[i] = A;
            fe25_neg(q.s[i].X, A.X);
            fe25_neg(q.s[i].T, A.T);
        } else {
            ept_load(q.s[i], out + (size_t)(j_first - 1) * SBV_ED_JBASE_DWORDS);
        }
    }
    SBV_NOUNROLL
    for (int j = j_first; j <= j_last; ++j) {
        if (j > 0) {
            SBV_NOUNROLL
            for (int d = 0; d < 8; ++d) edchain_dbl(q);
        }
        SBV_UNROLL
        for (int i = 0; i < QX::N; ++i) {
            const int r = q.role(i);
            fe25 c;
            fe25_select(c, r == 0, q.s[i].X, q.s[i].T);
            fe25_select(c, r == 1, q.s[i].Y, c);
            fe25_select(c, r == 2, q.s[i].Z, c);
            fe25_store_raw(out + (size_t)j * SBV_ED_JBASE_DWORDS + 10 * r, c);
        }
    }
}

// One part of one (key, window): row[k-1] = k * base for k = part*E + 1 .. part*E + E as affine-Niels points.
// `tmp` = private scratch of E * SBV_ED_WINDOW_TMP_WORDS dwords (X, Y, Z and the running product of the Zs, raw limbs).
#define SBV_ED_WINDOW_TMP_WORDS 40
// The multiples (m0 + 1 .. m0 + E) * base as affine-Niels entries at row + (m - 1) * pitch: a double-and-add ladder of `mbits` steps to
// m0 * base, E unified additions, ONE inversion for the lane's E points (Montgomery's trick).  tmp: E * 40 words.  Shared by the 8-bit
// per-batch combs (ed_keytab_window_lane: E = 128 / parts, 7 ladder steps, 96-byte pitch) and, since round 6, the 16-bit combs of hot
// keys (ed_widetab_lane: E = 32, 15 ladder steps, 128-byte pitch).
SBV_HD void ed_comb_part_lane(const pniels& base, u32 m0, int E, int mbits, u32* tmp, uint8_t* row, u32 pitch) {
    u32* pts = tmp;                   // E * 30 dwords
    u32* pre = tmp + E * 30;          // E * 10 dwords
    ept t;
    ed_set_ident(t);
    SBV_NOUNROLL
    for (int bit = mbits - 1; bit >= 0; --bit) {
        ed_dbl(t, t);
        ed_add_pniels(t, base, false, ((m0 >> bit) & 1u) == 0);
    }
    fe25 acc = fe25_one();
    SBV_NOUNROLL
    for (int k = 0; k < E; ++k) {
        ed_add_pniels(t, base, false, false);                 // (m0 + k + 1) * base
        fe25_store_raw(pts + k * 30, t.X); fe25_store_raw(pts + k * 30 + 10, t.Y); fe25_store_raw(pts + k * 30 + 20, t.Z);
        fe25_store_raw(pre + k * 10, acc);
        fe25_mul(acc, acc, t.Z);
    }
    fe25 inv;
    fe25_inv_gcd(inv, acc);           // Z is never 0 on a complete curve; an invalid key's garbage is never used
    const fe25 d2 = fe25_2d();
    SBV_NOUNROLL
    for (int k = E - 1; k >= 0; --k) {
        fe25 X, Y, Z, pk, zi, x, y;
        fe25_load_raw(X, pts + k * 30); fe25_load_raw(Y, pts + k * 30 + 10); fe25_load_raw(Z, pts + k * 30 + 20);
        fe25_load_raw(pk, pre + k * 10);
        fe25_mul(zi, inv, pk);
        fe25_mul(inv, inv, Z);
        fe25_mul(x, X, zi);
        fe25_mul(y, Y, zi);
        aniels_r a;
        fe25_add(a.ypx, y, x);
        fe25_sub(a.ymx, y, x);
        fe25_mul(a.xy2d, x, y);
        fe25_mul(a.xy2d, a.xy2d, d2);
        aniels_store(reinterpret_cast<aniels*>(row + (size_t)(m0 + (u32)k) * pitch), a);
    }
}
SBV_HD void ed_keytab_window_lane(const u32* jbase, int part, int parts, u32* tmp, aniels* row) {
    const int E = SBV_ED_KEY_PER_WINDOW / parts;      // parts is a power of two <= 16
    ept b;
    ept_load(b, jbase);
    pniels base;
    ed_to_pniels(base, b);
    ed_comb_part_lane(base, (u32)(part * E), E, 7, tmp, reinterpret_cast<uint8_t*>(row), (u32)sizeof(aniels));
}

// ---- hot keys of this scheme (round 6; VERDICT r5 #5, Missing #5: "hot keys as P-256 has") ---------------------------------------------
// A cached key that keeps signing gets a 16-bit comb of -A in the layout of the comb of B: comb[(j << 15) + (m - 1)] = m * 2^(16 j) * (-A),
// 16 windows x 32 768 entries at a 128-byte pitch = 64 MiB (HBM holds 288 GB): [k](-A) is 16 additions instead of 32, walked by the very
// function that walks the comb of B (ed_add_sB_comb).  The bookkeeping — counts per cache slot, promotion threshold, clock sweep,
// eviction with hysteresis — is the P-256 step's (p256_group.h: group_hot_class_lane, group_promote_select_lane, hot_evict_*), on this
// scheme's own arrays.  The builder needs no doubling chain: B_j = 2^(16 j) (-A) is entry 1 of row 2 j of the slot's 8-bit comb, an
// affine-Niels point, i.e. a projective-Niels point with Z = 1; lane (j, part) walks to (32 part) B_j and emits 32 multiples.
// (the constants: ed25519_core.h, where the host side sees them too)
// key_tab: the promoted slot's 8-bit comb; comb: its wide comb (SBV_ED_HOT_COMB_BYTES)
SBV_HD void ed_widetab_lane(const aniels* key_tab, u32 j, u32 part, u32* tmp, uint8_t* comb) {
    raw_aniels e;
    raw_aniels_load(e, key_tab + (size_t)(2 * j) * SBV_ED_KEY_PER_WINDOW);
    aniels_r a;
    raw_aniels_unpack(a, e);
    pniels base;
    base.YpX = a.ypx; base.YmX = a.ymx; base.Z = fe25_one(); base.T2d = a.xy2d;
    ed_comb_part_lane(base, part * SBV_ED_HOT_LANE_ENTRIES, SBV_ED_HOT_LANE_ENTRIES, 15, tmp,
                      comb + (size_t)j * SBV_ED_HOT_PER_WINDOW * SBV_ED_HOT_PITCH, SBV_ED_HOT_PITCH);
}

// gacc: SBV_ED_GACC_WORDS words per tuple (X, Y, Z, T raw limbs), limb-major: word w of tuple i at gacc[w * cap + i]
#define SBV_ED_GACC_WORDS 40
SBV_HD void ed_gacc_store(u32* gacc, size_t cap, size_t i, const ept& R) {
    SBV_UNROLL
    for (int l = 0; l < 10; ++l) {
        gacc[(size_t)l * cap + i] = (u32)R.X.v[l];
        gacc[(size_t)(10 + l) * cap + i] = (u32)R.Y.v[l];
        gacc[(size_t)(20 + l) * cap + i] = (u32)R.Z.v[l];
        gacc[(size_t)(30 + l) * cap + i] = (u32)R.T.v[l];
    }
}
SBV_HD void ed_gacc_load(ept& R, const u32* gacc, size_t cap, size_t i) {
    SBV_UNROLL
    for (int l = 0; l < 10; ++l) {
        R.X.v[l] = (i32)gacc[(size_t)l * cap + i];
        R.Y.v[l] = (i32)gacc[(size_t)(10 + l) * cap + i];
        R.Z.v[l] = (i32)gacc[(size_t)(20 + l) * cap + i];
        R.T.v[l] = (i32)gacc[(size_t)(30 + l) * cap + i];
    }
}

// Tuple-major twin (key-sorted step): the 40 words of tuple i are contiguous (160 bytes, ten 16-byte vectors).  The G phase
// runs in tuple order before the grouping is known, the Q phase in key order: with one record per tuple either order costs
// two cache lines per visit.
struct alignas(16) ed_q4 { u32 x, y, z, w; };
SBV_HD void ed_gacc_store_tm(u32* gacc, size_t i, const ept& R) {
    u32 w[40];
    SBV_UNROLL
    for (int l = 0; l < 10; ++l) { w[l] = (u32)R.X.v[l]; w[10 + l] = (u32)R.Y.v[l]; w[20 + l] = (u32)R.Z.v[l]; w[30 + l] = (u32)R.T.v[l]; }
    ed_q4* d = reinterpret_cast<ed_q4*>(gacc + i * SBV_ED_GACC_WORDS);
    SBV_UNROLL
    for (int q = 0; q < 10; ++q) { const ed_q4 v = {w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]}; d[q] = v; }
}
SBV_HD void ed_gacc_load_tm(ept& R, const u32* gacc, size_t i) {
    u32 w[40];
    const ed_q4* s = reinterpret_cast<const ed_q4*>(gacc + i * SBV_ED_GACC_WORDS);
    SBV_UNROLL
    for (int q = 0; q < 10; ++q) { const ed_q4 v = s[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    SBV_UNROLL
    for (int l = 0; l < 10; ++l) { R.X.v[l] = (i32)w[l]; R.Y.v[l] = (i32)w[10 + l]; R.Z.v[l] = (i32)w[20 + l]; R.T.v[l] = (i32)w[30 + l]; }
}

// The grouped step's comb of B: `bits`-wide signed windows as the ECDSA steps' combs of G (p256_comb29.h: gcomb_recode / gcomb_digit),
// tab[(j << (bits-1)) + (m-1)] = m * 2^(bits j) * B.  S < L < 2^253, so windows = ceil(254 / bits) keeps S + the recoding offsets
// inside the windows: 16 additions at 16 bits (50 MB, also the one-lane kernel's table), 13 at 20 bits (654 MB; HBM holds 288 GB).
// (struct edcomb: ed25519_core.h)
// R += [S]B; the next entry is fetched one addition ahead
SBV_HD void ed_add_sB_comb(ept& R, const u256& S, const edcomb& bc) {
    u288 ss;
    gcomb_recode(ss, S, bc.bits, bc.windows);
    u32 idx; bool neg, skip;
    gcomb_digit(ss, bc.bits, 0, idx, neg, skip);
    raw_aniels cur;
    raw_aniels_load(cur, edcomb_entry(bc, idx));
    SBV_NOUNROLL
    for (int j = 0; j < bc.windows; ++j) {
        const int jn = j + 1 < bc.windows ? j + 1 : bc.windows - 1;
        u32 idxn; bool negn, skipn;
        gcomb_digit(ss, bc.bits, jn, idxn, negn, skipn);
        raw_aniels nxt;
        raw_aniels_load(nxt, edcomb_entry(bc, ((size_t)jn << (bc.bits - 1)) + idxn));
        aniels_r e;
        raw_aniels_unpack(e, cur);
        ed_add_aniels(R, e, neg, skip);
        cur = nxt; neg = negn; skip = skipn;
    }
}

// [S]B (comb `bc`) for tuple i -> gacc; okb[i] = S < L and k < L (SetCanonicalBytes(S); a reduced k is always < L)
SBV_HD void ed_gphase_lane(const uint8_t* tuples, size_t i, const edcomb& bc, u32* gacc, size_t cap, uint8_t* okb, bool tuple_major = false) {
    const u32* w = ed_tuple_words(tuples, i);
    u256 S, k;
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) { S.v[j] = w[8 + j]; k.v[j] = w[24 + j]; }
    const u256 L = ed_L();
    okb[i] = (lt256(S, L) && lt256(k, L)) ? 1 : 0;
    ept R;
    ed_set_ident(R);
    ed_add_sB_comb(R, S, bc);
    if (tuple_major) ed_gacc_store_tm(gacc, i, R);
    else ed_gacc_store(gacc, cap, i, R);
}

// R (from gacc) += windows [j0, j1) of [k](-A) from the key's comb; R goes back to gacc.  `last` -> the return value is the
// tuple's pending flag (see ed_finish_lane); otherwise it is meaningless.
SBV_HD bool ed_qphase_lane(const uint8_t* tuples, size_t i, u32 slot, u32 nkeys, const aniels* ktab, const uint8_t* kvalid,
                           u32* gacc, size_t cap, const uint8_t* okb, int j0, int j1, bool last, bool tuple_major = false) {
    const u32* w = ed_tuple_words(tuples, i);
    u256 k, kk;
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) k.v[j] = w[24 + j];
    bool ok = (!last || okb[i] != 0) && slot < nkeys;          // the range verdict of the G phase only matters to the final answer
    if (slot >= nkeys) slot = 0;
    ok = ok && kvalid[slot] != 0;
    const aniels* tab = ktab + (size_t)slot * SBV_ED_KEYTAB_ENTRIES;
    (void)add_const_limbs(kk, k, 0x80808080u);       // k >= L was rejected above; k < 2^253 cannot carry out
    int d = (int)((ed_word_at(kk, j0 >> 2) >> ((j0 & 3) * 8)) & 255u) - 128;
    raw_aniels cur;
    raw_aniels_load(cur, tab + (size_t)j0 * SBV_ED_KEY_PER_WINDOW + ((d < 0 ? -d : d) == 0 ? 0 : (d < 0 ? -d : d) - 1));
    ept R;
    if (tuple_major) ed_gacc_load_tm(R, gacc, i);
    else ed_gacc_load(R, gacc, cap, i);
    SBV_NOUNROLL
    for (int j = j0; j < j1; ++j) {
        const int jn = j + 1 < j1 ? j + 1 : j;                     // the entry of the next window is fetched while this addition runs
        const int dn = (int)((ed_word_at(kk, jn >> 2) >> ((jn & 3) * 8)) & 255u) - 128;
        const int adn = dn < 0 ? -dn : dn;
        raw_aniels nxt;
        raw_aniels_load(nxt, tab + (size_t)jn * SBV_ED_KEY_PER_WINDOW + (adn == 0 ? 0 : adn - 1));
        aniels_r e;
        raw_aniels_unpack(e, cur);
        ed_add_aniels(R, e, d < 0, d == 0);
        cur = nxt; d = dn;
    }
    // Every chunk parks R; after the last one the return value says whether the tuple is still a candidate (range checks passed,
    // its key is a point): the encoding comparison — one field inversion per tuple if done here, a third of the lane's whole
    // instruction count — is left to ed_finish_lane, which shares one inversion among SBV_ED_FINISH_T tuples.
    if (tuple_major) ed_gacc_store_tm(gacc, i, R);
    else ed_gacc_store(gacc, cap, i, R);
    return last && ok;
}

// The wide pass of a hot key's tuple: R (tuple-major gacc) += [k](-A) from the slot's 16-bit comb, all 16 windows in one go; the return
// value is the tuple's pending flag, as ed_qphase_lane's after the last chunk.
SBV_HD bool ed_qphase_wide_lane(const uint8_t* tuples, size_t i, bool slot_ok, const uint8_t* comb, u32* gacc, const uint8_t* okb) {
    const u32* w = ed_tuple_words(tuples, i);
    u256 k;
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) k.v[j] = w[24 + j];
    ept R;
    ed_gacc_load_tm(R, gacc, i);
    const edcomb kc = edcomb_make(reinterpret_cast<const aniels*>(comb), SBV_ED_HOT_BITS, SBV_ED_HOT_PITCH);
    ed_add_sB_comb(R, k, kc);         // the comb is of -A: the same walker as the comb of B
    ed_gacc_store_tm(gacc, i, R);
    return slot_ok && okb[i] != 0;
}

// ---- finish: encode(R) == R_enc for every pending tuple, ONE inversion per SBV_ED_FINISH_T tuples (Montgomery's trick) --------
// crypto/ed25519.Verify compares encodings byte for byte, and the encoding needs x = X/Z, y = Y/Z.  Done per lane at the end of
// the Q phase the inversion (division steps, ~30 000 wave instructions with the lanes' divergent step counts) ran in all 16 384
// wavefronts of a 2^20 batch; here a lane takes T consecutive tuples, multiplies the Z of the pending ones together (the running
// prefix parks in the T slot of the tuple's accumulator record, which nobody needs any more), inverts the product once and walks
// back: 5 multiplications per tuple and 1/T of an inversion.  acc[i] == SBV_ED_PENDING marks a pending tuple (written by the last Q
// chunk); it becomes the verdict.  Z of a pending tuple is never 0 (complete addition law on valid points); a zero is rejected
// on its own anyway so that it cannot poison its neighbours' product.
#define SBV_ED_FINISH_T 8
#define SBV_ED_PENDING 2
SBV_HD void ed_gacc_load_fe(fe25& r, const u32* gacc, size_t cap, size_t i, int coord, bool tuple_major) {
    SBV_UNROLL
    for (int l = 0; l < 10; ++l) r.v[l] = (i32)(tuple_major ? gacc[i * SBV_ED_GACC_WORDS + 10 * coord + l] : gacc[(size_t)(10 * coord + l) * cap + i]);
}
SBV_HD void ed_gacc_store_fe(u32* gacc, size_t cap, size_t i, int coord, bool tuple_major, const fe25& a) {
    SBV_UNROLL
    for (int l = 0; l < 10; ++l) {
        if (tuple_major) gacc[i * SBV_ED_GACC_WORDS + 10 * coord + l] = (u32)a.v[l];
        else gacc[(size_t)(10 * coord + l) * cap + i] = (u32)a.v[l];
    }
}
SBV_HD void ed_finish_lane(const uint8_t* tuples, size_t n, size_t i0, u32* gacc, size_t cap, uint8_t* acc, bool tuple_major) {
    fe25 prod = fe25_one();
    SBV_NOUNROLL
    for (int j = 0; j < SBV_ED_FINISH_T; ++j) {
        const size_t i = i0 + (size_t)j;
        if (i >= n || acc[i] != SBV_ED_PENDING) continue;
        fe25 Z;
        ed_gacc_load_fe(Z, gacc, cap, i, 2, tuple_major);
        if (fe25_is_zero(Z)) { acc[i] = 0; continue; }
        ed_gacc_store_fe(gacc, cap, i, 3, tuple_major, prod);      // product of the pending Z before this one
        fe25_mul(prod, Z, prod);
    }
    fe25 inv;
    fe25_inv_gcd(inv, prod);
    SBV_NOUNROLL
    for (int j = SBV_ED_FINISH_T - 1; j >= 0; --j) {
        const size_t i = i0 + (size_t)j;
        if (i >= n || acc[i] != SBV_ED_PENDING) continue;
        fe25 Z, pre, X, Y, zi, x, y;
        ed_gacc_load_fe(Z, gacc, cap, i, 2, tuple_major);
        ed_gacc_load_fe(pre, gacc, cap, i, 3, tuple_major);
        ed_gacc_load_fe(X, gacc, cap, i, 0, tuple_major);
        ed_gacc_load_fe(Y, gacc, cap, i, 1, tuple_major);
        fe25_mul(zi, pre, inv);                                     // 1 / Z_i
        fe25_mul(inv, Z, inv);
        fe25_mul(x, X, zi);
        fe25_mul(y, Y, zi);
        u256 yw;
        fe25_freeze(yw, y);
        yw.v[7] |= (fe25_is_negative(x) ? 1u : 0u) << 31;
        const u32* w = ed_tuple_words(tuples, i);
        u32 diff = 0;
        SBV_UNROLL
        for (int k = 0; k < 8; ++k) diff |= yw.v[k] ^ w[k];
        acc[i] = diff == 0 ? 1 : 0;
    }
}

// tuple words straight from HBM (the ungrouped list is sparse: no LDS staging)
struct EdGlobalTuple {
    const u32* p;
    SBV_HD u32 operator[](int i) const { return p[i]; }
};

// ---- the ungrouped list on FOUR lanes per tuple (round 6) ---------------------------------------------------------------------------
// Once a scheme's combs are cached the one-lane kernel over the ungrouped list — a few hundred wavefronts, each a serial chain of 256
// doublings and 80 additions per lane — is the step's critical path (1.1 ms of a 2.2 ms hot step, profiles/r06/timeline_ed_hot_r06z.txt).
// Here the four lanes of a quad hold one tuple's point and every doubling / addition is two multiplications deep instead of eight
// (edchain_dbl above; edquad_add below, the same split of ed_add_pniels: lane r multiplies (Y-X)(Y2-X2) | (Y+X)(Y2+X2) | T 2dT2 | Z Z2,
// the products travel by DPP broadcasts, lane r multiplies E F | G H | F G | E H).  An affine-Niels entry (the comb of B) is the same
// addition with Z2 = 1.  The key arrives decompressed from the key check in front (ed_group_keycheck_lane keeps X and Y of the keys it
// accepts): the square root is not taken twice.  Same group law on the same inputs as ed25519_verify_lane, so the same verdicts
// (tests/emul runs both on every ungrouped tuple).
// `op` = this lane's level-1 multiplier: role 0 -> Y2 - X2, 1 -> Y2 + X2, 2 -> 2 d T2, 3 -> Z2 (already swapped / negated for -q)
template <class QX>
SBV_HD void edquad_add(QX& q, const fe25 op[QX::N], bool skip) {
    if (skip) return;                  // the same digit in the four lanes of a quad
    fe25 P[QX::N], a[QX::N], b[QX::N], c[QX::N], d[QX::N];
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) {
        const ept& s = q.s[i];
        const int r = q.role(i);
        fe25 ymx, ypx, in;
        fe25_sub(ymx, s.Y, s.X);
        fe25_add(ypx, s.Y, s.X);
        fe25_select(in, r == 0, ymx, ypx);
        fe25_select(in, r == 2, s.T, in);
        fe25_select(in, r == 3, s.Z, in);
        fe25_mul(P[i], in, op[i]);
    }
    q.bcast(a, P, 0); q.bcast(b, P, 1); q.bcast(c, P, 2); q.bcast(d, P, 3);
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) {
        const int r = q.role(i);
        fe25 dd, e, f, g, h, lhs, rhs;
        fe25_add(dd, d[i], d[i]);
        fe25_sub(e, b[i], a[i]);          // bounds as in ed_add_pniels: a, b, c tight; dd, e, h <= 2; f, g <= 3
        fe25_sub(f, dd, c[i]);
        fe25_add(g, dd, c[i]);
        fe25_add(h, b[i], a[i]);
        fe25_select(lhs, r == 1, g, f);   // first operand:  f | g | f | e
        fe25_select(lhs, r == 3, e, lhs);
        fe25_select(rhs, r == 0, e, h);   // second operand: e | h | g | h
        fe25_select(rhs, r == 2, g, rhs);
        fe25_mul(P[i], lhs, rhs);
    }
    q.bcast(a, P, 0); q.bcast(b, P, 1); q.bcast(c, P, 2); q.bcast(d, P, 3);
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) { q.s[i].X = a[i]; q.s[i].Y = b[i]; q.s[i].Z = c[i]; q.s[i].T = d[i]; }
}
// this lane's multiplier out of a projective-Niels table entry (pn_store layout: Y+X | Y-X | Z | 2dT, ten raw limbs each)
SBV_HD void edquad_op_pniels(fe25& op, const u32* entry, int role, bool neg) {
    const int coord = role == 0 ? (neg ? 0 : 1) : role == 1 ? (neg ? 1 : 0) : role == 2 ? 3 : 2;
    fe25_load_raw(op, entry + 10 * coord);
    fe25_cneg(op, op, neg && role == 2);
}
// ... and out of a packed affine-Niels entry (y+x | y-x | 2dxy, eight words each); Z2 = 1
struct edquad_words { u32 w[8]; };
SBV_HD void edquad_fetch_aniels(edquad_words& o, const aniels* entry, int role, bool neg) {
    const int coord = role == 0 ? (neg ? 0 : 1) : role == 1 ? (neg ? 1 : 0) : 2;      // role 3 fetches 2dxy too and ignores it
    const f25_q4* s = reinterpret_cast<const f25_q4*>(entry) + 2 * coord;
    const f25_q4 lo = s[0], hi = s[1];
    o.w[0] = lo.x; o.w[1] = lo.y; o.w[2] = lo.z; o.w[3] = lo.w; o.w[4] = hi.x; o.w[5] = hi.y; o.w[6] = hi.z; o.w[7] = hi.w;
}
SBV_HD void edquad_op_aniels(fe25& op, const edquad_words& e, int role, bool neg) {
    fe25 t;
    fe25_from_words(t, e.w);
    fe25_cneg(t, t, neg && role == 2);
    fe25_select(op, role == 3, fe25_one(), t);
}
// One ungrouped tuple on a quad -> accept?  `xy` = X | Y of the decompressed key (20 raw limbs, from the key check), `qtab` = the
// tuple's 8 x 40 words of table space, `b16` = the 16-bit comb of B of the one-lane kernel.  Every lane returns the verdict.
template <class QX, typename Words>
SBV_HD void ed25519_verify_quad(QX& q, Words w, const u32* xy, u32* qtab, const aniels* b16, bool out[QX::N]) {
    u32 renc[8];
    u256 S, k;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { renc[i] = w[i]; S.v[i] = w[8 + i]; k.v[i] = w[24 + i]; }
    const u256 L = ed_L();
    const bool ok = lt256(S, L) && lt256(k, L);
    const fe25 d2 = fe25_2d();
    // -A, and the table of m (-A), m = 1 .. 8: lane r keeps coordinate r of every entry
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) {
        ept& s = q.s[i];
        fe25 x;
        fe25_load_raw(x, xy);
        fe25_load_raw(s.Y, xy + 10);
        fe25_mul(s.T, x, s.Y);
        fe25_neg(s.X, x);
        fe25_neg(s.T, s.T);
        s.Z = fe25_one();
    }
    SBV_NOUNROLL
    for (int m = 1; m <= SBV_ED_QTAB_ENTRIES; ++m) {
        if (m == 2) edchain_dbl(q);
        else if (m > 2) {
            fe25 op[QX::N];
            SBV_UNROLL
            for (int i = 0; i < QX::N; ++i) edquad_op_pniels(op[i], qtab, q.role(i), false);
            edquad_add(q, op, false);
        }
        SBV_UNROLL
        for (int i = 0; i < QX::N; ++i) {
            const ept& s = q.s[i];
            const int r = q.role(i);
            fe25 ypx, ymx, t2d, c;
            fe25_add(ypx, s.Y, s.X);
            fe25_sub(ymx, s.Y, s.X);
            fe25_mul(t2d, s.T, d2);
            fe25_select(c, r == 0, ypx, ymx);
            fe25_select(c, r == 2, s.Z, c);
            fe25_select(c, r == 3, t2d, c);
            fe25_store_raw(qtab + (m - 1) * SBV_ED_PT_WORDS + 10 * r, c);
        }
    }
    u256 kk;
    (void)add_const_limbs(kk, k, 0x88888888u);
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) ed_set_ident(q.s[i]);
    SBV_NOUNROLL
    for (int win = 63; win >= 0; --win) {
        SBV_NOUNROLL
        for (int t = 0; t < 4; ++t) edchain_dbl(q);
        const int d = (int)((ed_word_at(kk, win >> 3) >> ((win & 7) * 4)) & 15u) - 8;
        const int ad = d < 0 ? -d : d;
        fe25 op[QX::N];
        SBV_UNROLL
        for (int i = 0; i < QX::N; ++i) edquad_op_pniels(op[i], qtab + (ad == 0 ? 0 : ad - 1) * SBV_ED_PT_WORDS, q.role(i), d < 0);
        edquad_add(q, op, d == 0);
    }
    // + [S]B from the 16-bit comb, the next entry fetched one addition ahead (ed_add_sB)
    {
        u256 ss;
        (void)add_const_limbs(ss, S, 0x80008000u);
        int d = (int)(ss.v[0] & 0xFFFFu) - 32768;
        edquad_words cur[QX::N];
        SBV_UNROLL
        for (int i = 0; i < QX::N; ++i) edquad_fetch_aniels(cur[i], b16 + ((d < 0 ? -d : d) == 0 ? 0 : (d < 0 ? -d : d) - 1), q.role(i), d < 0);
        SBV_NOUNROLL
        for (int j = 0; j < SBV_ED_B16_WINDOWS; ++j) {
            const int jn = j + 1 < SBV_ED_B16_WINDOWS ? j + 1 : j;
            const int dn = (int)((ed_word_at(ss, jn >> 1) >> ((jn & 1) * 16)) & 0xFFFFu) - 32768;
            const int adn = dn < 0 ? -dn : dn;
            edquad_words nxt[QX::N];
            fe25 op[QX::N];
            SBV_UNROLL
            for (int i = 0; i < QX::N; ++i) {
                edquad_fetch_aniels(nxt[i], b16 + (size_t)jn * SBV_ED_B16_PER_WINDOW + (adn == 0 ? 0 : adn - 1), q.role(i), dn < 0);
                edquad_op_aniels(op[i], cur[i], q.role(i), d < 0);
            }
            edquad_add(q, op, d == 0);
            SBV_UNROLL
            for (int i = 0; i < QX::N; ++i) cur[i] = nxt[i];
            d = dn;
        }
    }
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) out[i] = ok && ed_encoding_matches(q.s[i], renc);
}

}  // namespace sbv
