// p256_keytab29.h — per-batch comb tables of the keys that repeat inside a batch, built on the carry-free field.
//
// A key's comb is ktab[j * 128 + (k-1)] = k * 2^(8j) * Q for j = 0..32, k = 1..128 (affine, canonical words of the
// R = 2^261 domain; window 32 only ever uses k = 1).  It is rebuilt inside every call for the keys that repeat often
// enough in that call (p256_group.h); nothing survives the call.  Three kernels, each a handful of lanes per unit of
// work, all latency-bound chains that run beside the throughput kernels (G phase / Q phase):
//
//   chain   four lanes per key:  the doubling chain 2^(8j) Q in modified Jacobian coordinates (X : Y : Z : T = a Z^4), the three
//                                product levels of a doubling spread over the quad (DPP broadcasts); records B_j = 2^(8j) Q and
//                                16 B_j per window, NOT normalised
//   rows    two lanes per (key, window):   lane 0: the "babies" b * B_j, b = 1..8; lane 1: the "giants" 16 a * B_j,
//                                a = 1..8 — chains of XYZZ mixed additions (8M + 2S) on the isomorphic curve where the
//                                recorded base is affine, normalised with one inversion per lane
//   fill    eight lanes per (key, window): lane a fills both sides of giant 16 a — entries 16 a + b and 16 a - b from babies
//                                b = 1..8 as AFFINE + AFFINE additions; +b and -b share the inverse of x_b - x_16a and the
//                                eight inverses of a lane come from ONE inversion (Montgomery's trick): 5M + 1S per entry
//                                instead of the ~17M + 4S of a Jacobian addition followed by a normalisation
//
// Exceptional cases: every sum formed here is (16 a + b) * B with 0 < 16 a + b <= 128 and B of prime order n > 2^255,
// so the two summands of an affine addition never share an x coordinate; the chains use the exact pt29_madd anyway.
// A key that pointFromAffine refuses (off the curve, coordinate >= p) gets valid = 0 and garbage tables nobody reads.
#pragma once
#include "p256_comb29.h"
#include "p256_group.h"

namespace sbv {

#define SBV_KT29_POINTS_PER_WINDOW 8                       // record slots per window: 2^d B_j, d = 0..7; the chain writes d = 0 and 4 (B_j and 16 B_j), which is what the rows step reads
#define SBV_KT29_ROWS_TMP_WORDS (15 * 45)                  // per lane of the rows kernel: up to 15 points x (X, Y, ZZ, ZZZ, prefix); 7 used
#define SBV_KT29_FILL_TMP_WORDS (15 * 4 * 9)               // scratch stride unit of the fill kernel (a lane uses 8 x 9 words at r * SBV_KT29_FILL_TMP_WORDS / 2)

SBV_HD void f29_store_raw(u32* dst, const fe29& a) {
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) dst[l] = (u32)a.v[l];
}
SBV_HD void f29_load_raw(fe29& a, const u32* src) {
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) a.v[l] = (i32)src[l];
}
SBV_HD void apt29_store_canon(apt* dst, const apt29& a) {
    u32 w[16];
    f29_store_canon(w, a.x);
    f29_store_canon(w + 8, a.y);
    struct alignas(16) q4 { u32 x, y, z, w; };
    q4* d = reinterpret_cast<q4*>(dst);
    SBV_UNROLL
    for (int k = 0; k < 4; ++k) { q4 v = {w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]}; d[k] = v; }
}

// ---- chain -------------------------------------------------------------------------------------------------------------
// The doubling chain 2^(8j) Q is the one serial dependency of a fresh key: 256 doublings, nothing to overlap inside one lane.
// FOUR lanes (one quad) share a key and split every doubling at the level of the group-law formula, in modified Jacobian
// coordinates (X : Y : Z : T = a Z^4; Cohen-Miyaji-Ono), whose 4M + 4S doubling is only THREE products deep:
//   level 1   XX = X^2            | YY = Y^2           | YZ = Y Z
//   level 2   AA = A^2, A = 2 YY  | XA = X A           | MM = M^2, M = 3 XX + T
//   level 3   M (S - X3) - U      | U T                                 with S = 2 XA, U = 2 AA, X3 = MM - 2 S
//   X3, Y3 = level 3 left, Z3 = 2 YZ, T3 = 2 U T
// Every lane holds the whole state; at each level lane r of the quad runs product r (lane 3 repeats lane 2's: the wavefront is
// SIMD, an idle lane costs the same), and the results travel by DPP quad_perm broadcasts — full-rate register moves, no LDS.
// One doubling is 3 x (81 + 59) multiply-accumulates per lane instead of the 3M + 5S = 8 products of pt29_dbl_jac in a row.
// The level functions below are plain per-lane code; keychain29_dbl<QX> strings them together through an exchange policy:
// on the device QX is one lane + DPP (p256_group_kernels.hip), in tests/emul it is four lanes stepped in lockstep.
//
// Bounds (units of p; "vr" = value-reduced as f29_red_q / f29_norm_red leave a value: within (-0.01, 1.01), limbs 0..7
// within 2^27 of [0, 2^29)): state X, Y, Z, T vr.  Level 1 products of two vr values, reduced with the 32-bit multiplier:
// within +-4.1, then f29_red_q -> vr.  A = 2 YY <= 2.02, M = 3 XX + T <= 4.04 (both re-normalised limb-wise), so
// |A||B| <= 16.4 at level 2 and AA, XA, MM lie within +-4.6 with exact 29-bit limbs 0..7.  X3 = MM - 4 XA within +-23
// -> f29_norm_red -> vr.  Level 3: M (S - X3) with |S - X3| <= 10.2: |A||B| <= 41.3 (the contract allows 64), minus
// U = 2 AA (+-9.2) in the column domain: within +-15 -> f29_red_q -> vr; U T: |A||B| <= 9.3.  tests/emul runs all of it
// under SBV_F29_CHECK.
struct kchain { fe29 X, Y, Z, T; };
#define SBV_KT29_REC_WORDS 36                              // one recorded point: X, Y, Z, T raw limbs
#define SBV_KT29_STATE_WORDS 36                            // the running point between chunks

SBV_HD void f29_pick(fe29& r, bool c, const fe29& a, const fe29& b) { f29_select(r, c, a, b); }

// level 1: role 0 -> X X, role 1 -> Y Y, roles 2, 3 -> Y Z
SBV_HD void keychain29_l1(fe29& P1, const kchain& s, int role) {
    fe29 a, b, t;
    f29_pick(a, role == 0, s.X, s.Y);
    f29_pick(t, role == 1, s.Y, s.Z);
    f29_pick(b, role == 0, s.X, t);
    f29_mulx(P1, a, b);
    f29_red_q(P1);
}
// after the exchange every lane holds XX, YY, YZ (vr).  M and Z3 are the same in every lane.
SBV_HD void keychain29_l2(fe29& P2, fe29& M, fe29& Z3, const kchain& s, const fe29& XX, const fe29& YY, const fe29& YZ, int role) {
    fe29 A, t, a, b;
    f29_add(A, YY, YY);
    f29_norm(A, A);                               // A = 2 YY
    f29_add(t, XX, XX);
    f29_add(t, t, XX);
    f29_norm(t, t);
    f29_add(t, t, s.T);
    f29_norm(M, t);                               // M = 3 XX + T
    f29_add(t, YZ, YZ);
    f29_norm_red(Z3, t);                          // Z3 = 2 Y Z
    f29_pick(t, role == 1, s.X, M);
    f29_pick(a, role == 0, A, t);                 // A | X | M
    f29_pick(b, role < 2, A, M);                  // A | A | M
    f29_mulx(P2, a, b);
}
// after the exchange every lane holds AA, XA, MM (reduce_x outputs: limbs 0..7 exact).  X3 is the same in every lane.
// P3: role 0 (and 2, 3) -> Y3 = M (S - X3) - U, role 1 -> U T (T3 = 2 U T).
SBV_HD void keychain29_l3(fe29& P3, fe29& X3, const kchain& s, const fe29& M, const fe29& AA, const fe29& XA, const fe29& MM, int role) {
    fe29 S, U, t, a, b, V;
    f29_add(S, XA, XA);                           // S = 2 XA, limbs < 2^30
    f29_sub(t, MM, S);
    f29_sub(t, t, S);                             // limbs within (-2^31, 2^29)
    f29_norm_red(X3, t);                          // X3 = MM - 2 S
    f29_add(U, AA, AA);
    f29_norm(U, U);                               // U = 2 AA
    f29_sub(t, S, X3);
    f29_norm(t, t);
    f29_pick(a, role == 1, U, M);
    f29_pick(b, role == 1, s.T, t);
    f29_pick(V, role == 1, f29_zero(), U);
    f29_cols c;
    f29_cols_zero(c);
    f29_cols_mul(c, a, b);
    f29_cols_sub_val(c, V);
    f29_reduce_x(P3, c);
    f29_red_q(P3);
}
// Y3r = level 3 of lane 0, UTr = level 3 of lane 1
SBV_HD void keychain29_finish(kchain& s, const fe29& X3, const fe29& Z3, const fe29& Y3r, const fe29& UTr) {
    fe29 t;
    s.X = X3;
    s.Y = Y3r;
    s.Z = Z3;
    f29_add(t, UTr, UTr);
    f29_norm_red(s.T, t);
}

// Exchange policy of the emulator and the unit tests: the four lanes of a quad stepped in lockstep.
struct keychain_quad_host {
    static const int N = 4;
    kchain s[4];
    int role(int i) const { return i; }
    void bcast(fe29 out[4], const fe29 in[4], int src) const { for (int i = 0; i < 4; ++i) out[i] = in[src]; }
};
// One doubling of the quad's point.  QX: N lanes in q.s[0..N), role(i) in 0..3, bcast(out, in, src) = every lane's copy of
// lane src's value.
template <class QX>
SBV_HD void keychain29_dbl(QX& q) {
    fe29 P[QX::N], XX[QX::N], YY[QX::N], YZ[QX::N], M[QX::N], Z3[QX::N], X3[QX::N];
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) keychain29_l1(P[i], q.s[i], q.role(i));
    q.bcast(XX, P, 0); q.bcast(YY, P, 1); q.bcast(YZ, P, 2);
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) keychain29_l2(P[i], M[i], Z3[i], q.s[i], XX[i], YY[i], YZ[i], q.role(i));
    q.bcast(XX, P, 0); q.bcast(YY, P, 1); q.bcast(YZ, P, 2);        // AA, XA, MM
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) keychain29_l3(P[i], X3[i], q.s[i], M[i], XX[i], YY[i], YZ[i], q.role(i));
    q.bcast(XX, P, 0); q.bcast(YY, P, 1);                             // Y3, U T
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) keychain29_finish(q.s[i], X3[i], Z3[i], XX[i], YY[i]);
}

// the chain's first point: (x : y : 1 : -3)
SBV_HD void keychain29_start(kchain& s, const fe29& x, const fe29& y) {
    fe29 t;
    s.X = x; s.Y = y; s.Z = f29_one();
    f29_add(t, s.Z, s.Z);
    f29_add(t, t, s.Z);
    f29_neg(t, t);
    f29_norm_red(s.T, t);
}
SBV_HD void kchain_store(u32* dst, const kchain& s) {
    f29_store_raw(dst, s.X); f29_store_raw(dst + 9, s.Y); f29_store_raw(dst + 18, s.Z); f29_store_raw(dst + 27, s.T);
}
SBV_HD void kchain_load(kchain& s, const u32* src) {
    f29_load_raw(s.X, src); f29_load_raw(s.Y, src + 9); f29_load_raw(s.Z, src + 18); f29_load_raw(s.T, src + 27);
}
// coordinate `role` of the state (each lane of a quad stores one of the four)
SBV_HD void kchain_store_part(u32* dst, const kchain& s, int role) {
    fe29 a, b, c;
    f29_pick(a, role == 0, s.X, s.Y);
    f29_pick(b, role == 2, s.Z, s.T);
    f29_pick(c, role < 2, a, b);
    f29_store_raw(dst + 9 * role, c);
}

// The whole chain of one chunk for one quad.  bases: [(gidx * 33 + j) * 8 + d] records of SBV_KT29_REC_WORDS words =
// 2^d B_j, B_j = 2^(8j) Q, as the chain left them (modified Jacobian, NOT normalised).  jstate[gidx]: the running point between
// chunks (B of the next window).  valid: the byte of this key's TABLE SLOT (written with the first chunk).  rec_mask: bit d =
// record 2^d B_j (the chain-of-additions rows kernel reads d = 0 and 4 only: 0x11).
template <class QX>
SBV_HD void keychain29_run(QX& q, const uint8_t* tuples, u32 gidx, const GroupState& g, u32* jstate, u32* bases, uint8_t* valid,
                           int j_first, int j_last, u32 rec_mask = 0xFFu) {
    u32* st = jstate + (size_t)gidx * SBV_KT29_STATE_WORDS;
    if (j_first == 0) {
        fe29 x, y;
        const bool ok = key29_load(tuples, g.group_rep[gidx], x, y);
        SBV_UNROLL
        for (int i = 0; i < QX::N; ++i) {
            if (q.role(i) == 0) *valid = ok ? 1 : 0;
            keychain29_start(q.s[i], x, y);
        }
        // A key pointFromAffine refuses gets NO table (round 5): nothing is recorded, the rows / fill steps and the later chunks of the
        // chain skip its slot (valid = 0), the Q phase rejects its lanes without reading a row.  Bit-flipped variants of the signers'
        // keys that repeat in a batch became groups when the threshold dropped to a handful of uses; their 256 doublings and 33 rows
        // were 0.15 ms of a cold step for verdicts that are "reject" whatever the table holds.  The quad decides as one (same key).
        if (!ok) return;
    } else {
        SBV_UNROLL
        for (int i = 0; i < QX::N; ++i) kchain_load(q.s[i], st);
    }
    // the state at the top of window j is B_j; every doubling on the way to B_(j+1) = 2^8 B_j is recorded (the rows kernels
    // build a window's babies and giants from 2^d B_j, d = 0..7, instead of walking chains of additions)
    SBV_NOUNROLL
    for (int j = j_first; j <= j_last; ++j) {
        SBV_NOUNROLL
        for (int d = 0; d < 8; ++d) {
            u32* rec = bases + (((size_t)gidx * SBV_GTAB_WINDOWS + j) * SBV_KT29_POINTS_PER_WINDOW + d) * SBV_KT29_REC_WORDS;
            if ((rec_mask >> d) & 1u) {
                SBV_UNROLL
                for (int i = 0; i < QX::N; ++i) kchain_store_part(rec, q.s[i], q.role(i));
            }
            if (j == SBV_GTAB_WINDOWS - 1) break;           // the top window has the single entry B_32
            keychain29_dbl(q);
        }
    }
    SBV_UNROLL
    for (int i = 0; i < QX::N; ++i) kchain_store_part(st, q.s[i], q.role(i));
}

// ---- rows ----------------------------------------------------------------------------------------------------------------
// which = 0: babies b * B, b = 1..8 -> row[b - 1];  which = 1: giants 16 a * B, a = 1..8 -> row[16 a - 1].
// base2 = the window's two chain records (B and 16 B, modified Jacobian); row = the window's 128 entries;
// tmp: SBV_KT29_ROWS_TMP_WORDS private words.  top_window (j == 32): only entry 1 exists (the carry digit is 0 or 1).
//
// The base point (X : Y : Z) is never normalised on its own.  (x, y) -> (x Z^2, y Z^3) maps P-256 onto the curve
// y^2 = x^3 - 3 Z^4 x + b Z^6, where the base is the AFFINE point (X, Y): the multiples are chains of mixed additions there
// (the addition formulas do not depend on the curve's coefficients; the one doubling takes a4 = T = -3 Z^4 from the chain),
// and a multiple (X' : Y' : ZZ' : ZZZ') maps back as x = X' / (ZZ' Z^2), y = Y' / (ZZZ' Z^3).  Z joins the lane's ONE
// inversion (Montgomery's trick over Z and the ZZZ' of the chain).
// crow (optional): the window's 16 entries of the key's COMPACT rows — babies 1..8 at 0..7, giants 16 a at 7 + a — a second copy
// of exactly what this lane writes into `row`.  The rows-only pass of the Q phase (p256_comb29.h: qphase29_point_narrow) gathers
// from it: 33 KB per key instead of 16 used entries strewn over 270 KB, which is what kept a 2^20 batch over 65 536 keys at a
// quarter of its issue rate (17 GB of table address space behind every gather; profiles/r05).
#define SBV_NTAB_PER_WINDOW 16
#define SBV_NTAB_ENTRIES (SBV_GTAB_WINDOWS * SBV_NTAB_PER_WINDOW)
SBV_HD void keytab29_rows_lane(const u32* base2, int which, bool top_window, u32* tmp, apt* row, apt* crow = nullptr) {
    const int babies = 8;                                                 // the symmetric fill needs babies 1..8 only
    kchain B;
    kchain_load(B, base2 + which * 4 * SBV_KT29_REC_WORDS);               // record 0 = B, record 4 = 16 B
    const int n = top_window ? 0 : (which == 0 ? babies - 1 : 7);         // points of the chain beyond its first (babies = 8: the symmetric fill below)
    apt29 step;
    step.x = B.X; step.y = B.Y;
    xyzz R;
    fe29 acc = f29_one();
    if (n > 0) {                                // 2 * base: the chain's only doubling; its ZZZ' opens the running product
        pt29_mdbl_a(R, B.X, B.Y, B.T);
        f29_store_raw(tmp, R.X); f29_store_raw(tmp + 9, R.Y); f29_store_raw(tmp + 18, R.ZZ); f29_store_raw(tmp + 27, R.ZZZ);
        f29_store_raw(tmp + 36, acc);
        acc = R.ZZZ;
    }
    SBV_NOUNROLL
    for (int k = 1; k < n; ++k) {
        pt29_madd(R, step, false);
        u32* rec = tmp + k * 45;
        f29_store_raw(rec, R.X); f29_store_raw(rec + 9, R.Y); f29_store_raw(rec + 18, R.ZZ); f29_store_raw(rec + 27, R.ZZZ);
        f29_store_raw(rec + 36, acc);
        f29_mul(acc, acc, R.ZZZ);
    }
    // the base's coordinates are fetched again where they are needed instead of living in registers through the chain and the
    // inversion (no spills at 2 waves per SIMD; compiled for 3 — 168 VGPRs, 88 spilled dwords — the step is no faster:
    // profiles/r03/ab_rows_waves_r03v.jsonl)
    const u32* brec = base2 + which * 4 * SBV_KT29_REC_WORDS;
    fe29 all, inv, zi, zi2, zi3, bz;
    f29_load_raw(bz, brec + 18);
    f29_mul(all, acc, bz);
    f29_inv(inv, all);
    f29_mul(zi, inv, acc);                      // 1 / Z
    f29_load_raw(bz, brec + 18);
    f29_mul(inv, inv, bz);                      // 1 / prod ZZZ'
    f29_sqr(zi2, zi);
    f29_mul(zi3, zi2, zi);
    {                                           // entry 1 = B itself; entry 16 = the giants' base
        apt29 a;
        fe29 bx, by;
        f29_load_raw(bx, brec); f29_load_raw(by, brec + 9);
        f29_mul(a.x, bx, zi2);
        f29_mul(a.y, by, zi3);
        apt29_store_canon(row + (which == 0 ? 0 : 15), a);
        if (crow) apt29_store_canon(crow + (which == 0 ? 0 : 8), a);
    }
    SBV_NOUNROLL
    for (int k = n - 1; k >= 0; --k) {
        const u32* rec = tmp + k * 45;
        fe29 X, Y, ZZ, ZZZ, pre, i3, w, w2;
        f29_load_raw(X, rec); f29_load_raw(Y, rec + 9); f29_load_raw(ZZ, rec + 18); f29_load_raw(ZZZ, rec + 27); f29_load_raw(pre, rec + 36);
        f29_mul(i3, inv, pre);                  // 1 / ZZZ'
        f29_mul(inv, inv, ZZZ);
        f29_mul(w, ZZ, i3);                     // ZZ' / ZZZ'
        f29_sqr(w2, w);                         // 1 / ZZ'
        f29_mul(w2, w2, zi2);                   // 1 / (ZZ' Z^2)
        f29_mul(i3, i3, zi3);                   // 1 / (ZZZ' Z^3)
        apt29 a;
        f29_mul(a.x, X, w2);
        f29_mul(a.y, Y, i3);
        const int mult = which == 0 ? k + 2 : 16 * (k + 2);               // this point is mult * B
        apt29_store_canon(row + mult - 1, a);
        if (crow) apt29_store_canon(crow + (which == 0 ? k + 1 : 9 + k), a);
    }
}

// ---- fill ----------------------------------------------------------------------------------------------------------------
// Symmetric form: lane a = 1..8 fills BOTH sides of giant 16 a from babies 1..8 — 16 a + b (b = 1..7,
// a <= 7) and 16 a - b (b = 1..8) share the inverse of x_b - x_16a, because -b B is (x_b, -y_b).  Eight denominators per lane
// instead of fifteen, and the rows step only has to build babies 2..8 (7 additions instead of 15).  Entry 8 is a baby and is
// not written again (a = 1, b = 8); entries 9..15 come from giant 16.  Lanes write disjoint entries and read only babies
// 1..8 and their own giant, which the rows step wrote.  tmp: 8 x 9 words.
// In three steps since round 6, so that the lanes of a window can share ONE inversion (keytab29_fill_group_inverses): the product
// of the lane's eight denominators | its inverse | the walk back.
SBV_HD void keytab29_fill_sym_acc(int a, u32* tmp, const apt* row, fe29& acc) {
    apt29 G;
    apt29_load(G, reinterpret_cast<const u32*>(row + 16 * a - 1));
    acc = f29_one();
    SBV_NOUNROLL
    for (int b = 1; b <= 8; ++b) {
        apt29 S;
        apt29_load(S, reinterpret_cast<const u32*>(row + b - 1));
        fe29 d;
        f29_sub(d, S.x, G.x);
        f29_store_raw(tmp + (b - 1) * 9, acc);
        f29_mul(acc, acc, d);
    }
}
// w: 16 x 9 words of one window — the eight lanes' products in w[0..8), scratch behind them.  On return w[r] = 1 / product r:
// Montgomery's trick over the lanes, one inversion per window instead of one per lane (an inversion is three quarters of a fill
// lane's instructions).  No denominator of a valid key is 0 (babies and giants are distinct multiples of a point of prime order);
// a zero would cost the whole window, not one lane, which is why the fill skips keys that are no points.
SBV_HD void keytab29_fill_group_inverses(u32* w) {
    fe29 acc = f29_one();
    SBV_NOUNROLL
    for (int r = 0; r < 8; ++r) {
        fe29 v;
        f29_load_raw(v, w + r * 9);
        f29_store_raw(w + (8 + r) * 9, acc);
        f29_mul(acc, acc, v);
    }
    fe29 inv;
    f29_inv(inv, acc);
    SBV_NOUNROLL
    for (int r = 7; r >= 0; --r) {
        fe29 v, pre, t;
        f29_load_raw(v, w + r * 9);
        f29_load_raw(pre, w + (8 + r) * 9);
        f29_mul(t, inv, pre);
        f29_mul(inv, inv, v);
        f29_store_raw(w + r * 9, t);
    }
}
SBV_HD void keytab29_fill_sym_finish(int a, u32* tmp, apt* row, fe29 inv);
SBV_HD void keytab29_fill_sym_lane(int a, u32* tmp, apt* row) {
    fe29 acc, inv;
    keytab29_fill_sym_acc(a, tmp, row, acc);
    f29_inv(inv, acc);
    keytab29_fill_sym_finish(a, tmp, row, inv);
}
SBV_HD void keytab29_fill_sym_finish(int a, u32* tmp, apt* row, fe29 inv) {
    apt29 G;
    apt29_load(G, reinterpret_cast<const u32*>(row + 16 * a - 1));
    SBV_NOUNROLL
    for (int b = 8; b >= 1; --b) {
        apt29 S, r;
        apt29_load(S, reinterpret_cast<const u32*>(row + b - 1));
        fe29 d, pre, dinv;
        f29_sub(d, S.x, G.x);
        f29_load_raw(pre, tmp + (b - 1) * 9);
        f29_mul(dinv, inv, pre);
        f29_mul(inv, inv, d);
        if (a <= 7 && b <= 7) {
            apt29_add_with_inverse(r, G, S, dinv);
            apt29_store_canon(row + 16 * a + b - 1, r);
        }
        if (!(a == 1 && b == 8)) {
            fe29 ny;
            f29_neg(ny, S.y);
            S.y = ny;
            apt29_add_with_inverse(r, G, S, dinv);
            apt29_store_canon(row + 16 * a - b - 1, r);
        }
    }
}

}  // namespace sbv
