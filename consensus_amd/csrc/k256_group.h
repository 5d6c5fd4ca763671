// k256_group.h — the grouped step for secp256k1: per-batch key combs and the comb phases over k256_fe.h (SURVEY.md §8f row 4;
// VERDICT r2 next #9).  The machinery above the field is the P-256 grouped step's (p256_group.h: hash-table grouping by the 64
// key bytes, counting sort by key, XCD-aware Q phase over the key-sorted list); what is new here is everything that touches
// the curve y^2 = x^3 + 7 over p = 2^256 - 2^32 - 977:
//
//   k256_chain_*     the doubling chain 2^(8j) Q of a fresh key on FOUR lanes (a = 0: plain Jacobian doubling is three
//                    products deep — X^2 | Y^2 | Y Z, then Y^4 | X Y^2 | (3 X^2)^2, then E (4 X Y^2 - X3) — so no fourth
//                    coordinate is needed; the exchange is the DPP quad broadcast of p256_keytab29.h)
//   k256_rows_lane   babies b B (b = 1..16) and giants 16 a B (a = 2..8) of one window from the chain's Jacobian records: the
//                    base (X : Y : Z) is the AFFINE point (X, Y) of the isomorphic curve y^2 = x^3 + 7 Z^6 (a stays 0, and
//                    neither the doubling nor the mixed addition uses b), Z joins the lane's one inversion
//   k256_fill_lane   entry 16 a + b = giant_a + baby_b, affine + affine with one inversion per lane (Montgomery's trick)
//   k256_gphase_*    u1 * G from the 16-bit comb of G (17 mixed additions) -> the accumulator parked in gacc
//   k256_qphase_*    += windows [j0, j1) of u2 * Q from the key's 8-bit comb; the last chunk checks R.x == r (mod n)
//
// Tables hold 64-byte affine entries (kapt: canonical words, plain residues) in the SAME comb pool as the P-256 per-batch
// area; the persistent key-table cache is NOT used for this curve (its slots are keyed by the 64 key bytes alone, and a byte
// string can be a point of both curves).  Every addition is exact (kpt_madd), verdicts equal k256_verify_lane's.
//
// Shared host/device source (tests/emul runs the same functions sequentially).
#pragma once
#include "k256_core.h"
#include "p256_group.h"

namespace sbv {

#define SBV_K256_REC_WORDS 27                              // one recorded chain point: X, Y, Z raw limbs
#define SBV_K256_BASES_STRIDE 72                           // words per (key, window) in GroupBuffers::bases: two records of <= 36
#define SBV_K256_STATE_WORDS 36                            // GroupBuffers::jstate stride (27 used)
#define SBV_K256_ROWS_TMP_WORDS (15 * 36)                  // per rows lane: 15 points x (X, Y, Z, prefix)
#define SBV_K256_FILL_TMP_WORDS (15 * 9)                   // per fill lane: 15 prefix products
#define SBV_K256_WINDOW_TMP (2 * SBV_K256_ROWS_TMP_WORDS)  // rows and fill share a window's strip (same stream, never concurrent)
#define SBV_K256_GACC_WORDS 27

struct kchain3 { kfe X, Y, Z; };

SBV_HD void kfe_pick(kfe& r, bool c, const kfe& a, const kfe& b) { kfe_select(r, c, a, b); }

// pointFromAffine on this curve: coordinates < p, y^2 = x^3 + 7
SBV_HD bool k256_key_load(const uint8_t* tuples, size_t idx, kfe& x, kfe& y) {
    const u32* k = tuple_key_words(tuples, idx);
    u256 qx, qy;
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) { qx.v[l] = bswap32(k[7 - l]); qy.v[l] = bswap32(k[8 + 7 - l]); }
    const u256 p_ = k256_p_words();
    kfe_from_words(x, qx);
    kfe_from_words(y, qy);
    return lt256(qx, p_) && lt256(qy, p_) && k256_on_curve(x, y);
}

// ---- chain: one doubling on a quad ------------------------------------------------------------------------------------------
// level 1: role 0 -> X X, role 1 -> Y Y, roles 2, 3 -> Y Z
SBV_HD void k256_chain_l1(kfe& P1, const kchain3& s, int role) {
    kfe a, b, t;
    kfe_pick(a, role == 0, s.X, s.Y);
    kfe_pick(t, role == 1, s.Y, s.Z);
    kfe_pick(b, role == 0, s.X, t);
    kfe_mul(P1, a, b);
}
// every lane holds A = X^2, B = Y^2, YZ.  role 0 -> B B, role 1 -> X B, roles 2, 3 -> E E with E = 3 A
SBV_HD void k256_chain_l2(kfe& P2, kfe& E, kfe& Z3, const kchain3& s, const kfe& A, const kfe& B, const kfe& YZ, int role) {
    kfe a, b, t;
    kfe_scale(E, A, 3);
    kfe_scale(Z3, YZ, 2);
    kfe_pick(t, role == 1, s.X, E);
    kfe_pick(a, role == 0, B, t);                 // B | X | E
    kfe_pick(b, role < 2, B, E);                  // B | B | E
    kfe_mul(P2, a, b);
}
// every lane holds C = Y^4, XB = X Y^2, F = E^2: X3 = F - 8 XB, Y3 = E (4 XB - X3) - 8 C — the last product in every lane
SBV_HD void k256_chain_l3(kchain3& s, const kfe& E, const kfe& Z3, const kfe& C, const kfe& XB, const kfe& F) {
    kfe X3, t;
    kfe_lin(X3, F, 1, XB, 8);
    kfe_lin(t, XB, 4, X3, 1);
    kfe_mul(t, E, t);
    kfe_lin(s.Y, t, 1, C, 8);
    s.X = X3;
    s.Z = Z3;
}
struct k256_quad_host {
    static const int N = 4;
    kchain3 s[4];
    int role(int i) const { return i; }
    void bcast(kfe out[4], const kfe in[4], int src) const { for (int i = 0; i < 4; ++i) out[i] = in[src]; }
};
template <class QX>
SBV_HD void k256_chain_dbl(QX& q) {
    kfe P[QX::N], A[QX::N], B[QX::N], YZ[QX::N], E[QX::N], Z3[QX::N];
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) k256_chain_l1(P[i], q.s[i], q.role(i));
    q.bcast(A, P, 0); q.bcast(B, P, 1); q.bcast(YZ, P, 2);
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) k256_chain_l2(P[i], E[i], Z3[i], q.s[i], A[i], B[i], YZ[i], q.role(i));
    q.bcast(A, P, 0); q.bcast(B, P, 1); q.bcast(YZ, P, 2);        // C, XB, F
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) k256_chain_l3(q.s[i], E[i], Z3[i], A[i], B[i], YZ[i]);
}
SBV_HD void kchain3_store_part(u32* dst, const kchain3& s, int role) {
    kfe a, c;
    kfe_pick(a, role == 0, s.X, s.Y);
    kfe_pick(c, role < 2, a, s.Z);
    if (role < 3) kfe_store_raw(dst + 9 * role, c);
}
SBV_HD void kchain3_load(kchain3& s, const u32* src) { kfe_load_raw(s.X, src); kfe_load_raw(s.Y, src + 9); kfe_load_raw(s.Z, src + 18); }

// the chain of one chunk of windows for one quad: records B_j and 16 B_j of windows j_first..j_last
template <class QX>
SBV_HD void k256_chain_run(QX& q, const uint8_t* tuples, u32 gidx, const GroupState& g, u32* jstate, u32* bases, uint8_t* valid,
                           int j_first, int j_last) {
    u32* st = jstate + (size_t)gidx * SBV_K256_STATE_WORDS;
    if (j_first == 0) {
        kfe x, y;
        const bool ok = k256_key_load(tuples, g.group_rep[gidx], x, y);
        SBV_UNROLL
        for (int i = 0; i < QX::N; ++i) {
            if (q.role(i) == 0) *valid = ok ? 1 : 0;
            q.s[i].X = x; q.s[i].Y = y; q.s[i].Z = kfe_one();
        }
    } else {
        SBV_UNROLL
        for (int i = 0; i < QX::N; ++i) kchain3_load(q.s[i], st);
    }
    SBV_NOUNROLL
    for (int j = j_first; j <= j_last; ++j) {
        SBV_NOUNROLL
        for (int half = 0; half < 2; ++half) {
            if (j > 0 || half > 0) {
                SBV_NOUNROLL
                for (int d = 0; d < 4; ++d) k256_chain_dbl(q);
            }
            u32* rec = bases + ((size_t)gidx * SBV_GTAB_WINDOWS + j) * SBV_K256_BASES_STRIDE + half * 36;
            SBV_UNROLL
            for (int i = 0; i < QX::N; ++i) kchain3_store_part(rec, q.s[i], q.role(i));
            if (j == SBV_GTAB_WINDOWS - 1) break;
        }
    }
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) kchain3_store_part(st, q.s[i], q.role(i));
}

// ---- rows ----------------------------------------------------------------------------------------------------------------
// which = 0: babies b B, b = 1..16 -> row[b - 1]; which = 1: giants 16 a B, a = 2..8 -> row[16 a - 1].  base2: the window's two
// chain records.  On the isomorphic curve the base is affine; a multiple (X' : Y' : Z') there is (X' : Y' : Z' Z) here.
SBV_HD void k256_rows_lane(const u32* base2, int which, bool top_window, u32* tmp, kapt* row) {
    kchain3 Bp;
    kchain3_load(Bp, base2 + which * 36);
    const int n = top_window ? 0 : (which == 0 ? 15 : 7);
    kjpt T;
    T.X = Bp.X; T.Y = Bp.Y; T.Z = kfe_one(); T.inf = false;
    kfe acc = kfe_one();
    SBV_NOUNROLL
    for (int k = 0; k < n; ++k) {
        if (k == 0) kpt_dbl(T, T);
        else kpt_madd(T, T, Bp.X, Bp.Y, false, false);
        u32* rec = tmp + k * 36;
        kfe_store_raw(rec, T.X); kfe_store_raw(rec + 9, T.Y); kfe_store_raw(rec + 18, T.Z); kfe_store_raw(rec + 27, acc);
        kfe_mul(acc, acc, T.Z);
    }
    kfe all, inv, zb;
    kfe_mul(all, acc, Bp.Z);
    kfe_inv(inv, all);
    kfe_mul(zb, inv, acc);                        // 1 / Z of the base
    kfe_mul(inv, inv, Bp.Z);                      // 1 / prod Z'
    if (which == 0) {
        kfe zi2, zi3, x, y;
        kfe_sqr(zi2, zb);
        kfe_mul(zi3, zi2, zb);
        kfe_mul(x, Bp.X, zi2);
        kfe_mul(y, Bp.Y, zi3);
        kapt_store(row, x, y);
    }
    SBV_NOUNROLL
    for (int k = n - 1; k >= 0; --k) {
        const u32* rec = tmp + k * 36;
        kfe X, Y, Z, pre, zi, zi2, zi3;
        kfe_load_raw(X, rec); kfe_load_raw(Y, rec + 9); kfe_load_raw(Z, rec + 18); kfe_load_raw(pre, rec + 27);
        kfe_mul(zi, inv, pre);                    // 1 / Z'
        kfe_mul(inv, inv, Z);
        kfe_mul(zi, zi, zb);                      // 1 / (Z' Z)
        kfe_sqr(zi2, zi);
        kfe_mul(zi3, zi2, zi);
        kfe_mul(X, X, zi2);
        kfe_mul(Y, Y, zi3);
        const int mult = which == 0 ? k + 2 : 16 * (k + 2);
        kapt_store(row + mult - 1, X, Y);
    }
}

// ---- fill ----------------------------------------------------------------------------------------------------------------
// row a (1..7): entry 16 a + b = row[16 a - 1] + row[b - 1], b = 1..15.  The summands never share an x (distinct small multiples
// of a point of prime order), so the affine addition needs no exceptional cases; a key off the curve gives garbage nobody reads.
SBV_HD void k256_fill_lane(int a, u32* tmp, kapt* row) {
    kfe gx, gy;
    kapt_load(gx, gy, row + 16 * a - 1);
    kfe acc = kfe_one();
    SBV_NOUNROLL
    for (int b = 1; b <= 15; ++b) {
        kfe sx, sy, d;
        kapt_load(sx, sy, row + b - 1);
        kfe_sub(d, sx, gx);
        kfe_store_raw(tmp + (b - 1) * 9, acc);
        kfe_mul(acc, acc, d);
    }
    kfe inv;
    kfe_inv(inv, acc);
    SBV_NOUNROLL
    for (int b = 15; b >= 1; --b) {
        kfe sx, sy, d, pre, dinv, lam, t, x3, y3;
        kapt_load(sx, sy, row + b - 1);
        kfe_sub(d, sx, gx);
        kfe_load_raw(pre, tmp + (b - 1) * 9);
        kfe_mul(dinv, inv, pre);
        kfe_mul(inv, inv, d);
        kfe_sub_nc(t, sy, gy);
        kfe_mul(lam, t, dinv);                    // (y2 - y1) / (x2 - x1)
        kfe_sqr(t, lam);
        kfe_lin3(x3, t, gx, 1, sx, 1);            // lambda^2 - x1 - x2
        kfe_sub_nc(t, gx, x3);
        kfe_mul(t, lam, t);
        kfe_sub(y3, t, gy);                       // lambda (x1 - x3) - y1
        kapt_store(row + 16 * a + b - 1, x3, y3);
    }
}

// ---- accumulator between launches: 27 raw limbs per lane of the key-sorted list, limb-major; infinity = Z all zero -----------
SBV_HD void k256_gacc_store(u32* gacc, size_t cap, size_t i, const kjpt& R) {
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) {
        gacc[(size_t)l * cap + i] = (u32)R.X.v[l];
        gacc[(size_t)(9 + l) * cap + i] = (u32)R.Y.v[l];
        gacc[(size_t)(18 + l) * cap + i] = R.inf ? 0u : (u32)R.Z.v[l];
    }
}
SBV_HD void k256_gacc_load(kjpt& R, const u32* gacc, size_t cap, size_t i) {
    i32 o = 0;
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) {
        R.X.v[l] = (i32)gacc[(size_t)l * cap + i];
        R.Y.v[l] = (i32)gacc[(size_t)(9 + l) * cap + i];
        R.Z.v[l] = (i32)gacc[(size_t)(18 + l) * cap + i];
        o |= R.Z.v[l];
    }
    R.inf = o == 0;                               // a finite point has Z != 0 (mod p): its reduced limbs are never all zero
    if (R.inf) kpt_set_inf(R);
}

// stage A's per-tuple record for the key-sorted list (same layout as P-256's: u1 | u2 | r | ok)
SBV_HD void k256_rec_store(const Scratch& s, size_t i, const u256& u1, const u256& u2, const u256& r, bool ok) {
    rec_store256(s.rec, i, SBV_REC_U1, u1.v);
    rec_store256(s.rec, i, SBV_REC_U2, u2.v);
    rec_store256(s.rec, i, SBV_REC_R, r.v);
    s.rec[i * SBV_REC_WORDS + SBV_REC_OK] = ok ? 1u : 0u;
}

// A table entry as fetched (16 words): the next one is on its way while the current addition runs (the comb of G is 35.7 MB,
// a key's comb 270 KB: a gather is an L2 / Infinity Cache / HBM round trip that one addition's worth of arithmetic hides).
struct alignas(16) kraw { u32 w[16]; };
SBV_HD void kraw_load(kraw& e, const kapt* p) {
    struct alignas(16) q4 { u32 a, b, c, d; };
    const q4* s = reinterpret_cast<const q4*>(p);
    const q4 v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
    e.w[0] = v0.a; e.w[1] = v0.b; e.w[2] = v0.c; e.w[3] = v0.d; e.w[4] = v1.a; e.w[5] = v1.b; e.w[6] = v1.c; e.w[7] = v1.d;
    e.w[8] = v2.a; e.w[9] = v2.b; e.w[10] = v2.c; e.w[11] = v2.d; e.w[12] = v3.a; e.w[13] = v3.b; e.w[14] = v3.c; e.w[15] = v3.d;
}
SBV_HD void kraw_unpack(kfe& x, kfe& y, const kraw& e) {
    u256 wx, wy;
    SBV_UNROLL
    for (int k = 0; k < 8; ++k) { wx.v[k] = e.w[k]; wy.v[k] = e.w[8 + k]; }
    kfe_from_words(x, wx);
    kfe_from_words(y, wy);
}
// The comb of G of the grouped step: `bits`-wide signed windows like the P-256 comb (p256_comb29.h: gcomb_recode / gcomb_digit):
// tab[(j << (bits - 1)) + (m - 1)] = m * 2^(bits j) * G, windows = ceil(257 / bits).  16 bits = the one-lane kernel's table
// (17 additions, 35.7 MB); 20 bits = 13 additions from 436 MB, built once per process on the host.
struct kgcomb { const kapt* tab; int bits; int windows; };
SBV_HD kgcomb kgcomb_make(const kapt* tab, int bits) { kgcomb g = {tab, bits, (257 + bits - 1) / bits}; return g; }
// R += u1 * G, next entry prefetched
SBV_HD void k256_gphase_point(kjpt& R, const u256& u1, const kgcomb& gc) {
    u288 k1;
    gcomb_recode(k1, u1, gc.bits, gc.windows);
    u32 idx; bool neg, skip;
    gcomb_digit(k1, gc.bits, 0, idx, neg, skip);
    kraw cur;
    kraw_load(cur, gc.tab + idx);
    SBV_NOUNROLL
    for (int j = 0; j < gc.windows; ++j) {
        const int jn = j + 1 < gc.windows ? j + 1 : gc.windows - 1;
        u32 idxn; bool negn, skipn;
        gcomb_digit(k1, gc.bits, jn, idxn, negn, skipn);
        kraw nxt;
        kraw_load(nxt, gc.tab + ((size_t)jn << (gc.bits - 1)) + idxn);
        kfe x, y;
        kraw_unpack(x, y, cur);
        kpt_madd(R, R, x, y, neg, skip);
        cur = nxt; neg = negn; skip = skipn;
    }
}
SBV_HD void k256_gphase_lane_sorted(const Scratch& s, size_t t, size_t L, const kgcomb& gc, u32* gacc) {
    u256 u1;
    rec_load256(u1, s.rec, t, SBV_REC_U1);
    kjpt R;
    kpt_set_inf(R);
    k256_gphase_point(R, u1, gc);
    k256_gacc_store(gacc, s.cap, L, R);
}

// R.x mod n == r, without an inversion (k256_verify_lane's final test)
SBV_HD bool k256_rx_matches(const kjpt& R, const u256& r) {
    if (R.inf) return false;
    kfe zz, c1, t;
    kfe_sqr(zz, R.Z);
    kfe_from_words(c1, r);
    kfe_mul(t, c1, zz);
    bool match = kfe_equal(t, R.X);
    u256 rn;
    const u32 carry = add256(rn, r, k256_n_words());
    const bool wraps = carry == 0 && lt256(rn, k256_p_words());
    kfe_from_words(c1, rn);
    kfe_mul(t, c1, zz);
    return match || (wraps && kfe_equal(t, R.X));
}

// R += windows [j0, j1) of u2 * Q; qtab[j * 128 + (k-1)] = k * 2^(8 j) * Q.  As on P-256 a scalar with its top bit set is
// walked as n - u2 with every digit's sign flipped, so that the carry window (32) is needed by a wavefront only rarely.
SBV_HD void k256_qphase_point(kjpt& R, const u256& u2in, const kapt* qtab, int j0, int j1) {
    const bool flip = (u2in.v[7] >> 31) != 0;
    u256 u2, nmu;
    (void)sub256(nmu, k256_n_words(), u2in);
    select256(u2, flip, nmu, u2in);
    u256 k2;
    const u32 top2 = add_const_limbs(k2, u2, 0x80808080u);
    if (j1 == SBV_GTAB_WINDOWS && !wave_any(top2 != 0)) j1 = SBV_GTAB_WINDOWS - 1;
    if (j0 >= j1) return;
    int idx; bool neg, skip;
    comb_digit(k2, top2, j0, idx, neg, skip);
    kraw cur;
    kraw_load(cur, qtab + (size_t)j0 * SBV_GTAB_PER_WINDOW + idx);
    SBV_NOUNROLL
    for (int j = j0; j < j1; ++j) {
        const int jn = j + 1 < j1 ? j + 1 : j1 - 1;
        int idxn; bool negn, skipn;
        comb_digit(k2, top2, jn, idxn, negn, skipn);
        kraw nxt;
        kraw_load(nxt, qtab + (size_t)jn * SBV_GTAB_PER_WINDOW + idxn);
        kfe x, y;
        kraw_unpack(x, y, cur);
        kpt_madd(R, R, x, y, neg != flip, skip);
        cur = nxt; neg = negn; skip = skipn;
    }
}
SBV_HD bool k256_qphase_lane_sorted(const Scratch& s, size_t t, size_t L, u32 slot, u32 nslots, const kapt* ktab, const uint8_t* kvalid,
                                    u32* gacc, int j0, int j1, bool last) {
    u256 u2;
    rec_load256(u2, s.rec, t, SBV_REC_U2);
    bool ok = slot < nslots;
    if (slot >= nslots) slot = 0;
    ok = ok && kvalid[slot] != 0;
    const kapt* qtab = ktab + (size_t)slot * (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW);
    kjpt R;
    k256_gacc_load(R, gacc, s.cap, L);
    k256_qphase_point(R, u2, qtab, j0, j1);
    if (!last) {
        size_t Ls = L;                     // an opaque copy: the store addresses are computed again instead of living through the loop (p256_comb29.h)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(Ls));
#endif
        k256_gacc_store(gacc, s.cap, Ls, R);
        return false;
    }
    u256 r;
    rec_load256(r, s.rec, t, SBV_REC_R);
    ok = ok && s.rec[t * SBV_REC_WORDS + SBV_REC_OK] != 0;
    return ok && k256_rx_matches(R, r);
}

// key check of the ungrouped candidates (the P-256 step's group_keycheck_lane on this curve)
SBV_HD bool k256_keycheck_lane(const uint8_t* tuples, size_t L, const GroupState& g, uint8_t* acc) {
    const u32 i = g.ung_cand[L];
    kfe x, y;
    if (!k256_key_load(tuples, i, x, y)) {
        acc[i] = 0;
        SBV_ATOMIC_ADD(&g.counters[3], 1u);
        return false;
    }
    g.ung_idx[SBV_ATOMIC_ADD(&g.counters[2], 1u)] = i;
    return true;
}

}  // namespace sbv
