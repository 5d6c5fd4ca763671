// k256_core.h — ECDSA verification over secp256k1, one lane per signature: the "other curves" variant of the hot path
// (the seam is curve-agnostic: api.Verifier, pkg/api/dependencies.go:54-71).  Semantics = the P-256
// path's (Go crypto/ecdsa's generic rules, SEC 1 v2.0 §4.1.4) with this curve's parameters: y^2 = x^3 + 7 over
// p = 2^256 - 2^32 - 977, prime order n, cofactor 1.
//
//   stage A  k256_prep_lane     range checks (1 <= r, s < n; Qx, Qy < p), e = hash mod n, w = s^-1 mod n (division steps),
//                               u1 = e w, u2 = r w  -> the limb-major scratch planes of p256_core.h
//   stage B  k256_verify_lane   Q on the curve; the 8 affine multiples of Q (Jacobian chain, one inversion); u2 * Q = k1 Q + k2 phi(Q)
//                               (GLV, round 6) with 32 signed 4-bit windows (4 doublings of 3M + 4S and two mixed additions of 8M + 3S
//                               each: 128 doublings instead of 256); + u1 * G from a signed 16-bit
//                               comb of G (17 mixed additions, 17 x 32768 affine entries = 35.7 MB, built once per process);
//                               accept iff R != infinity and R.x = r (mod n), tested without an inversion: X = r Z^2 or
//                               X = (r + n) Z^2 when r + n < p
// Every addition is exact: P + infinity, P + P (doubling) and P + (-P) are handled inside kpt_madd.
//
// Shared host/device source (tests/emul runs the same functions sequentially).
#pragma once
#include "k256_fe.h"
#include "k256_sc.h"
#include "p256_core.h"

namespace sbv {

struct kapt { u32 x[8], y[8]; };                 // affine table entry: canonical words, 64 bytes
struct kjpt { kfe X, Y, Z; bool inf; };          // Jacobian; inf is authoritative (Z is then arbitrary)

SBV_HD u256 k256_gx_words() { u256 r = {{0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu}}; return r; }
SBV_HD u256 k256_gy_words() { u256 r = {{0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u}}; return r; }

SBV_HD void kpt_set_inf(kjpt& p) { p.X = kfe_one(); p.Y = kfe_one(); p.Z = kfe_zero(); p.inf = true; }

// y^2 == x^3 + 7
SBV_HD bool k256_on_curve(const kfe& x, const kfe& y) {
    kfe l, t, rr;
    kfe_sqr(l, y);
    kfe_sqr(t, x);
    kfe_mul(rr, t, x);
    kfe seven = kfe_zero();
    seven.v[0] = 7;
    kfe_add(rr, rr, seven);
    return kfe_equal(l, rr);
}

// 2P, a = 0.  3M + 4S and five carries: with near-full-rate 64-bit multiply-accumulates a carry pass costs about as much as a
// squaring, so X Y^2 is one more product instead of ((X + Y^2)^2 - X^2 - Y^4) / 2 with its three carried additions.
// No point of order two exists (the group order is prime), so Y = 0 never happens on the curve.
SBV_HD void kpt_dbl(kjpt& r, const kjpt& p) {
    kfe A, B, C, XB, E, F, Z3, t;
    kfe_sqr(A, p.X);
    kfe_sqr(B, p.Y);
    kfe_sqr(C, B);
    kfe_mul(XB, p.X, B);                          // X Y^2
    kfe_scale(E, A, 3);
    kfe_sqr(F, E);
    kfe_mul(Z3, p.Y, p.Z);
    kfe_scale(Z3, Z3, 2);
    kfe_lin(r.X, F, 1, XB, 8);                    // X3 = E^2 - 8 X Y^2
    kfe_lin(t, XB, 4, r.X, 1);                    // 4 X Y^2 - X3
    kfe_mul(t, E, t);
    kfe_lin(r.Y, t, 1, C, 8);                     // Y3 = E (4 X Y^2 - X3) - 8 Y^4
    r.Z = Z3;
    r.inf = p.inf;
}

// r = p + (x2, y2) (affine, reduced, not infinity; `neg` adds (x2, -y2); `skip` adds nothing).  8M + 3S and two carries: the
// differences H, R and V - X3 go into their products uncarried.  Exact in every case: P + P and P + (-P) are detected on the
// uncarried H (a 3-in-2^29 filter on its low limb, then the exact test) and resolved after the fact, so the common path stays
// branch-free.
SBV_HD void kpt_madd(kjpt& r, const kjpt& p, const kfe& x2, const kfe& y2in, bool neg, bool skip) {
    kfe y2;
    kfe_cneg_nc(y2, y2in, neg);
    kfe Z1Z1, U2, S2, H, Rr, HH, HHH, V, t;
    kfe_sqr(Z1Z1, p.Z);
    kfe_mul(U2, x2, Z1Z1);
    kfe_mul(S2, p.Z, Z1Z1);
    kfe_mul(S2, y2, S2);
    kfe_sub_nc(H, U2, p.X);
    kfe_sub_nc(Rr, S2, p.Y);
    kfe_sqr(HH, H);
    kfe_mul(HHH, H, HH);
    kfe_mul(V, p.X, HH);
    kjpt s;
    kfe_sqr(t, Rr);
    kfe_lin3(s.X, t, HHH, 1, V, 2);               // X3 = R^2 - H^3 - 2 V
    kfe_sub_nc(t, V, s.X);
    kfe_mul(t, Rr, t);
    kfe m;
    kfe_mul(m, p.Y, HHH);
    kfe_sub(s.Y, t, m);                           // Y3 = R (V - X3) - Y1 H^3
    kfe_mul(s.Z, p.Z, H);
    s.inf = false;
    if (kfe_diff_maybe_zero(H) && !p.inf) {       // rare: decide exactly
        kfe hc, rc;
        kfe_carry32(hc, H);
        if (kfe_is_zero(hc)) {
            kfe_carry32(rc, Rr);
            if (kfe_is_zero(rc)) {                // p == (x2, y2): the result is 2 (x2, y2)
                kjpt a;
                a.X = x2; kfe_carry32(a.Y, y2); a.Z = kfe_one(); a.inf = false;
                kpt_dbl(s, a);
            } else {                              // p == -(x2, y2)
                kpt_set_inf(s);
            }
        }
    }
    if (p.inf) { s.X = x2; kfe_carry32(s.Y, y2); s.Z = kfe_one(); s.inf = false; }
    if (skip) s = p;
    r = s;
}

SBV_HD void kapt_load(kfe& x, kfe& y, const kapt* e) {
    struct alignas(16) q4 { u32 a, b, c, d; };
    const q4* s = reinterpret_cast<const q4*>(e);
    const q4 v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
    u256 wx = {{v0.a, v0.b, v0.c, v0.d, v1.a, v1.b, v1.c, v1.d}}, wy = {{v2.a, v2.b, v2.c, v2.d, v3.a, v3.b, v3.c, v3.d}};
    kfe_from_words(x, wx);
    kfe_from_words(y, wy);
}
SBV_HD void kapt_store(kapt* e, const kfe& x, const kfe& y) {
    u256 wx, wy;
    kfe_to_words(wx, x);
    kfe_to_words(wy, y);
    SBV_UNROLL
    for (int k = 0; k < 8; ++k) { e->x[k] = wx.v[k]; e->y[k] = wy.v[k]; }
}

// ---- the comb of G: window j (0..16) row m - 1 (m = 1..32768) holds m * 2^(16 j) * G ------------------------------------------
#define SBV_K256_G_WINDOWS 17
#define SBV_K256_G_PER_WINDOW 32768
#define SBV_K256_G_ENTRIES ((size_t)SBV_K256_G_WINDOWS * SBV_K256_G_PER_WINDOW)

// R += u1 * G: k = u1 + sum_j 2^(16 j + 15); 16-bit word j of k, minus 32768, is the signed digit of window j and the carry out
// of the top word is the digit (0 or 1) of window 16
SBV_HD void k256_add_u1G(kjpt& R, const u256& u1, const kapt* gtab) {
    u256 k;
    const u32 top = add_const_limbs(k, u1, 0x80008000u);
    SBV_NOUNROLL
    for (int j = 0; j < SBV_K256_G_WINDOWS; ++j) {
        int idx; bool neg, skip;
        comb16_digit(k, top, j, idx, neg, skip);
        kfe x, y;
        kapt_load(x, y, gtab + (size_t)j * SBV_K256_G_PER_WINDOW + idx);
        kpt_madd(R, R, x, y, neg, skip);
    }
}

// ---- stage A -----------------------------------------------------------------------------------------------------------
// tuple = r | s | hash | Qx | Qy, 5 x 32 big-endian bytes (include/sbv.h); w = its 40 dwords
template <typename WordPtr>
SBV_HD void k256_prep_lane(WordPtr w, size_t i, const Scratch& sc_) {
    u256 r, s, e, qx, qy;
    tuple_field(r, w, 0);
    tuple_field(s, w, 1);
    tuple_field(e, w, 2);
    tuple_field(qx, w, 3);
    tuple_field(qy, w, 4);
    const u256 n_ = k256_n_words(), p_ = k256_p_words();
    const bool ok = !is_zero256(r) && lt256(r, n_) && !is_zero256(s) && lt256(s, n_) && lt256(qx, p_) && lt256(qy, p_);
    ksc_cond_sub_n(e, e);                       // hashToNat: e < 2^256 < 2 n
    u256 one = {{1, 0, 0, 0, 0, 0, 0, 0}}, sv, wv, u1, u2;
    select256(sv, ok, s, one);                  // s >= n is rejected above; keep the inversion's input in range
    ksc_inv(wv, sv);
    ksc_mul(u1, e, wv);
    ksc_mul(u2, r, wv);
    soa_store(sc_.r, sc_.cap, i, r);
    soa_store(sc_.u1, sc_.cap, i, u1);
    soa_store(sc_.u2, sc_.cap, i, u2);
    soa_store(sc_.qx, sc_.cap, i, qx);
    soa_store(sc_.qy, sc_.cap, i, qy);
    sc_.ok[i] = ok ? 1 : 0;
    if (sc_.rec) {                              // grouped step (k256_group.h): the key-sorted list reads one record per tuple
        rec_store256(sc_.rec, i, SBV_REC_U1, u1.v);
        rec_store256(sc_.rec, i, SBV_REC_U2, u2.v);
        rec_store256(sc_.rec, i, SBV_REC_R, r.v);
        sc_.rec[i * SBV_REC_WORDS + SBV_REC_OK] = ok ? 1u : 0u;
    }
}

// Stage A of T tuples in one lane with ONE inversion (Montgomery's trick, as the P-256 slab kernel does): tuples
// first + k * step, k = 0..T-1.  Same results in the same places as k256_prep_lane.  Between the two passes the exclusive
// prefix product, s and e of a tuple are parked where its u1, its record's OK slot / the sm plane and its u2 will be.
// words(idx) = the tuple's 40 dwords.  Four multiplications mod n and 1 / T of a division chain per signature instead of
// two and a whole one.  (The grouped step runs T = 8: measured in round 4, 4.97 -> 4.68 ms per 2^20 step.)
template <typename WordsFn>
SBV_HD void k256_prep_chunk(WordsFn words, size_t n, const Scratch& sc_, size_t first, size_t step, int T) {
    const u256 n_ = k256_n_words(), p_ = k256_p_words();
    u256 acc = {{1, 0, 0, 0, 0, 0, 0, 0}};
    for (int k = 0; k < T; ++k) {
        const size_t idx = first + (size_t)k * step;
        if (idx >= n) continue;
        auto w = words(idx);
        u256 r, s, e, qx, qy;
        tuple_field(r, w, 0);
        tuple_field(s, w, 1);
        tuple_field(e, w, 2);
        tuple_field(qx, w, 3);
        tuple_field(qy, w, 4);
        const bool ok = !is_zero256(r) && lt256(r, n_) && !is_zero256(s) && lt256(s, n_) && lt256(qx, p_) && lt256(qy, p_);
        ksc_cond_sub_n(e, e);                   // hashToNat: e < 2^256 < 2 n
        u256 one = {{1, 0, 0, 0, 0, 0, 0, 0}}, sv;
        select256(sv, ok, s, one);              // keep the product chain invertible
        soa_store(sc_.u1, sc_.cap, idx, acc);   // exclusive prefix product
        soa_store(sc_.sm, sc_.cap, idx, sv);
        soa_store(sc_.u2, sc_.cap, idx, e);
        soa_store(sc_.r, sc_.cap, idx, r);
        soa_store(sc_.qx, sc_.cap, idx, qx);
        soa_store(sc_.qy, sc_.cap, idx, qy);
        sc_.ok[idx] = ok ? 1 : 0;
        u256 t;
        ksc_mul(t, acc, sv);
        acc = t;
    }
    u256 inv;
    ksc_inv(inv, acc);
    for (int k = T - 1; k >= 0; --k) {
        const size_t idx = first + (size_t)k * step;
        if (idx >= n) continue;
        u256 pre, sv, e, r, wv, u1, u2, t;
        soa_load(pre, sc_.u1, sc_.cap, idx);
        soa_load(sv, sc_.sm, sc_.cap, idx);
        soa_load(e, sc_.u2, sc_.cap, idx);
        soa_load(r, sc_.r, sc_.cap, idx);
        ksc_mul(wv, inv, pre);                  // s_k^-1
        ksc_mul(t, inv, sv);                    // drop s_k from the running inverse
        inv = t;
        ksc_mul(u1, e, wv);
        ksc_mul(u2, r, wv);
        soa_store(sc_.u1, sc_.cap, idx, u1);
        soa_store(sc_.u2, sc_.cap, idx, u2);
        if (sc_.rec) {
            rec_store256(sc_.rec, idx, SBV_REC_U1, u1.v);
            rec_store256(sc_.rec, idx, SBV_REC_U2, u2.v);
            rec_store256(sc_.rec, idx, SBV_REC_R, r.v);
            sc_.rec[idx * SBV_REC_WORDS + SBV_REC_OK] = sc_.ok[idx];
        }
    }
}

// ---- stage B -----------------------------------------------------------------------------------------------------------
// qtab: this lane's strip of SBV_K256_QTAB_WORDS dwords (16-byte aligned): 8 affine entries, then 7 raw chain records
#define SBV_K256_QTAB_WORDS (8 * 16 + 7 * 36 + 4)
SBV_HD void kfe_store_raw(u32* dst, const kfe& a) {
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) dst[l] = (u32)a.v[l];
}
SBV_HD void kfe_load_raw(kfe& a, const u32* src) {
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) a.v[l] = (i32)src[l];
}

SBV_HD bool k256_verify_lane(const Scratch& s, size_t i, u32* qtab, const kapt* gtab) {
    u256 u2, qxw, qyw;          // r and u1 are fetched where they are used, behind the doubling loop: 16 registers less across it
    soa_load(u2, s.u2, s.cap, i);
    soa_load(qxw, s.qx, s.cap, i);
    soa_load(qyw, s.qy, s.cap, i);
    bool ok = s.ok[i] != 0;
    kfe qx, qy;
    kfe_from_words(qx, qxw);
    kfe_from_words(qy, qyw);
    ok = ok && k256_on_curve(qx, qy);           // (0, 0) is not on the curve: 0 != 7
    // table k * Q, k = 1..8: Jacobian chain parked raw behind the table, normalised with ONE inversion (Montgomery's trick)
    kapt* tab = reinterpret_cast<kapt*>(qtab);
    u32* raw = qtab + 8 * 16;                   // record k - 2 (k = 2..8): X, Y, Z, prefix product of the Zs before it
    kapt_store(tab, qx, qy);
    {
        kjpt T;
        T.X = qx; T.Y = qy; T.Z = kfe_one(); T.inf = false;
        kfe acc = kfe_one();
        SBV_NOUNROLL
        for (int k = 2; k <= 8; ++k) {
            kpt_madd(T, T, qx, qy, false, false);            // k Q; k = 2 takes the doubling branch; never infinity (prime order > 8)
            u32* rec = raw + (k - 2) * 36;
            kfe_store_raw(rec, T.X); kfe_store_raw(rec + 9, T.Y); kfe_store_raw(rec + 18, T.Z); kfe_store_raw(rec + 27, acc);
            kfe_mul(acc, acc, T.Z);
        }
        kfe inv;
        kfe_inv(inv, acc);                      // a key off the curve gives garbage here; ok is already false
        SBV_NOUNROLL
        for (int k = 8; k >= 2; --k) {
            const u32* rec = raw + (k - 2) * 36;
            kfe X, Y, Z, pre, zi, zi2, zi3;
            kfe_load_raw(X, rec); kfe_load_raw(Y, rec + 9); kfe_load_raw(Z, rec + 18); kfe_load_raw(pre, rec + 27);
            kfe_mul(zi, inv, pre);
            kfe_mul(inv, inv, Z);
            kfe_sqr(zi2, zi);
            kfe_mul(zi3, zi2, zi);
            kfe_mul(X, X, zi2);
            kfe_mul(Y, Y, zi3);
            kapt_store(tab + (k - 1), X, Y);
        }
    }
    // u2 * Q = k1 * (+-Q) + k2 * (+-phi(Q)), phi(x, y) = (beta x, y): the GLV decomposition (k256_sc.h: ksc_split_lambda) halves the
    // doublings — 128 instead of 256, with two table additions per 4-bit window instead of one (round 6).  Both halves are below 2^128:
    // k_i + sum_j 8 * 16^j over 32 nibbles; nibble j, minus 8, is the signed digit of window j and the carry into bit 128 is window 32.
    // phi of a table entry is one multiplication by beta, done at use (33 per signature: three additions' worth).
    u256 k1, k2;
    bool n1, n2;
    ksc_split_lambda(k1, n1, k2, n2, u2);
    const u256 eights = {{0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u, 0u, 0u, 0u, 0u}};
    u256 kk1, kk2;
    (void)add256(kk1, k1, eights);
    (void)add256(kk2, k2, eights);
    // beta as nine 29-bit limbs (k256_sc.h: k256_beta_words): compile-time constants live in scalar registers, not in nine VGPRs across the loop
    const kfe beta = {{0x119501EE, 0x09CB6143, 0x1D626570, 0x0092EA25, 0x034E99CF, 0x03CF561A, 0x1C41B991, 0x056CAF80, 0x007AE96A}};
    kjpt R;
    kpt_set_inf(R);
    {
        kfe bx;
        kfe_mul(bx, qx, beta);
        kpt_madd(R, R, qx, qy, n1, kk1.v[4] == 0);
        kpt_madd(R, R, bx, qy, n2, kk2.v[4] == 0);
    }
    SBV_NOUNROLL
    for (int j = 31; j >= 0; --j) {
        SBV_NOUNROLL
        for (int d = 0; d < 4; ++d) kpt_dbl(R, R);
        u32 w1 = 0, w2 = 0;
        SBV_UNROLL
        for (int w = 0; w < 4; ++w) { w1 = (j >> 3) == w ? kk1.v[w] : w1; w2 = (j >> 3) == w ? kk2.v[w] : w2; }
        const int d1 = (int)((w1 >> ((j & 7) * 4)) & 15u) - 8;
        const int d2 = (int)((w2 >> ((j & 7) * 4)) & 15u) - 8;
        const int a1 = d1 < 0 ? -d1 : d1, a2 = d2 < 0 ? -d2 : d2;
        kfe x, y;
        kapt_load(x, y, tab + (a1 == 0 ? 0 : a1 - 1));
        kpt_madd(R, R, x, y, (d1 < 0) != n1, d1 == 0);
        kapt_load(x, y, tab + (a2 == 0 ? 0 : a2 - 1));
        kfe_mul(x, x, beta);
        kpt_madd(R, R, x, y, (d2 < 0) != n2, d2 == 0);
    }
    {
        u256 u1;
        soa_load(u1, s.u1, s.cap, i);
        k256_add_u1G(R, u1, gtab);
    }
    if (R.inf) return false;
    u256 r;
    soa_load(r, s.r, s.cap, i);
    // R.x mod n == r  <=>  X == r Z^2, or X == (r + n) Z^2 when r + n < p  (R.x in [n, p) wraps once: p < 2 n)
    kfe zz, c1, t;
    kfe_sqr(zz, R.Z);
    kfe_from_words(c1, r);
    kfe_mul(t, c1, zz);
    bool match = kfe_equal(t, R.X);
    u256 rn, d;
    const u32 carry = add256(rn, r, k256_n_words());
    const bool wraps = carry == 0 && lt256(rn, k256_p_words());
    kfe_from_words(c1, rn);
    kfe_mul(t, c1, zz);
    match = match || (wraps && kfe_equal(t, R.X));
    (void)d;
    return ok && match;
}

// ---- the comb of G, built on the host (or by the emulator): one call per window ------------------------------------------------
// row[m - 1] = m * 2^(16 j) * G, m = 1..count: a Jacobian chain normalised in chunks with Montgomery's trick
// (bits = window width: 16 for the device comb; the host signer uses an 8-bit comb of the same shape)
SBV_HD void k256_build_g_window_bits(int bits, int j, kapt* row, int count) {
    kjpt B;
    kfe_from_words(B.X, k256_gx_words());
    kfe_from_words(B.Y, k256_gy_words());
    B.Z = kfe_one();
    B.inf = false;
    for (int d = 0; d < bits * j; ++d) kpt_dbl(B, B);
    kfe bx, by;
    {   // base -> affine
        kfe zi, zi2, zi3;
        kfe_inv(zi, B.Z);
        kfe_sqr(zi2, zi);
        kfe_mul(zi3, zi2, zi);
        kfe_mul(bx, B.X, zi2);
        kfe_mul(by, B.Y, zi3);
    }
    const int CH = 64;
    kjpt T;
    kpt_set_inf(T);
    for (int m0 = 0; m0 < count; m0 += CH) {
        const int m1 = m0 + CH < count ? m0 + CH : count;
        kjpt pts[CH];
        kfe pre[CH];
        kfe acc = kfe_one();
        for (int m = m0; m < m1; ++m) {
            kpt_madd(T, T, bx, by, false, false);            // (m + 1) * base
            pts[m - m0] = T;
            pre[m - m0] = acc;
            kfe_mul(acc, acc, T.Z);
        }
        kfe inv;
        kfe_inv(inv, acc);
        for (int m = m1 - 1; m >= m0; --m) {
            kfe zi, zi2, zi3, X, Y;
            kfe_mul(zi, inv, pre[m - m0]);
            kfe_mul(inv, inv, pts[m - m0].Z);
            kfe_sqr(zi2, zi);
            kfe_mul(zi3, zi2, zi);
            kfe_mul(X, pts[m - m0].X, zi2);
            kfe_mul(Y, pts[m - m0].Y, zi3);
            kapt_store(row + m, X, Y);
        }
    }
}
SBV_HD void k256_build_g_window(int j, kapt* row, int count) { k256_build_g_window_bits(16, j, row, count); }

}  // namespace sbv
