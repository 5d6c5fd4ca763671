// p256_pt29.h — P-256 group law over the carry-free field of p256_fe29.h.
//
// Accumulators of the comb phases (u1*G from the 16-bit comb of G, u2*Q from a key's 8-bit comb) are held
// in XYZZ coordinates (X, Y, ZZ, ZZZ) with x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2 [Explicit-Formulas Database,
// "xyzz" for short Weierstrass curves]: adding an affine table point is 8M + 2S (madd-2008-s) against 8M + 3S
// for Jacobian coordinates, the two leading products (U2, S2) do not wait for a Z^2 -> Z^3 chain, and the
// final check R.x == r needs ZZ only.  Infinity is an explicit per-lane flag: coordinates are residues that
// are not kept canonical, so "ZZ == 0" would be a comparison mod p, not a register test.
//
// Every routine is EXACT for every input, like p256_pt.h: crypto/ecdsa's verdict on u1*G == +-u2*Q
//  depends on it.  The exceptional cases of an addition (P == Q -> doubling, P == -Q ->
// infinity) are detected by f29_maybe_zero(U2 - X1), three instructions, and resolved in a branch that
// honest data never takes.
#pragma once
#include "p256_fe29.h"
#include "p256_pt.h"
#include "p256_sc.h"

namespace sbv {

struct xyzz { fe29 X, Y, ZZ, ZZZ; bool inf; };
struct apt29 { fe29 x, y; };           // affine; limbs tight (unpacked from canonical storage)

// f29_norm plus a value reduction by the multiple of p the top limb indicates: any |value| < 16 p with limbs
// |v[i]| < 2^31  ->  value in (-2^229, 2^256 + 2^229), limbs 0..7 within (-2^25 - 8, 2^29 + 2^25 + 8).
// p = 2^256 - 2^224 + 2^192 + 2^96 - 1 touches five limbs: bits 256 / 224 / 192 / 96 / 0 are bit 24 of limb 8,
// 21 of limb 7, 18 of limb 6, 9 of limb 3, 0 of limb 0.
SBV_HD void f29_norm_red(fe29& r, const fe29& a) {
    f29_norm(r, a);
    const i32 q = r.v[8] >> 24;                 // floor(value / 2^256), |q| <= 16
    r.v[8] -= f29_shl(q, 24);
    r.v[7] += f29_shl(q, 21);
    r.v[6] -= f29_shl(q, 18);
    r.v[3] -= f29_shl(q, 9);
    r.v[0] += q;
}

SBV_HD void pt29_set_inf(xyzz& R) {
    R.X = f29_zero(); R.Y = f29_zero(); R.ZZ = f29_zero(); R.ZZZ = f29_zero();
    R.inf = true;
}

// 2 * (x, y) for an affine point (never infinity: finite points of a prime-order curve have y != 0)
SBV_HD void pt29_mdbl(xyzz& R, const fe29& x, const fe29& y) {
    fe29 U, V, W, S, M, t, X3, Y3;
    f29_add(U, y, y);
    f29_norm(U, U);
    f29_sqr(V, U);
    f29_mul(W, U, V);
    f29_mul(S, x, V);
    f29_sqr(t, x);
    const fe29 one = f29_one();
    f29_sub(t, t, one);                         // x^2 - 1   (a = -3: M = 3 x^2 + a)
    f29_add(M, t, t);
    f29_add(M, M, t);
    f29_norm(M, M);
    f29_sqr(t, M);
    f29_sub(t, t, S);
    f29_sub(X3, t, S);                          // X3 = M^2 - 2 S
    f29_norm_red(R.X, X3);
    f29_sub(t, S, R.X);
    f29_mul(t, M, t);
    f29_mul(Y3, W, y);
    f29_sub(Y3, t, Y3);                         // Y3 = M (S - X3) - W y
    f29_norm_red(R.Y, Y3);
    R.ZZ = V;
    R.ZZZ = W;
    R.inf = false;
}

// 2 * (x, y) on the curve y^2 = x^3 + a4 x + b' for ANY a4 (mdbl-2008-s-1: the formulas never use b').  The table builder
// doubles a Jacobian point (X : Y : Z) of P-256 as the affine point (X, Y) of the isomorphic curve with a4 = -3 Z^4
// (p256_keytab29.h).  x, y, a4 value-reduced (f29_norm_red / f29_red_q); outputs as pt29_mdbl's.
SBV_HD void pt29_mdbl_a(xyzz& R, const fe29& x, const fe29& y, const fe29& a4) {
    fe29 U, V, W, S, M, t, X3, Y3;
    f29_add(U, y, y);
    f29_norm(U, U);
    f29_sqr(V, U);
    f29_mul(W, U, V);
    f29_mul(S, x, V);
    f29_sqr(t, x);
    f29_add(M, t, t);
    f29_add(M, M, t);
    f29_norm(M, M);                             // 3 x^2 alone reaches 3 * 2^29 per limb: one more loose limb would wrap the i32
    f29_add(M, M, a4);                          // M = 3 x^2 + a4, value within +-5.6 p
    f29_norm(M, M);
    f29_sqr(t, M);
    f29_sub(t, t, S);
    f29_sub(X3, t, S);                          // X3 = M^2 - 2 S
    f29_norm_red(R.X, X3);
    f29_sub(t, S, R.X);
    f29_norm(t, t);
    f29_mul(t, M, t);
    f29_mul(Y3, W, y);
    f29_sub(Y3, t, Y3);                         // Y3 = M (S - X3) - W y
    f29_norm_red(R.Y, Y3);
    R.ZZ = V;
    R.ZZZ = W;
    R.inf = false;
}

// R += (q.x, neg ? -q.y : q.y).  R.X, R.Y as left by this function (or pt29_mdbl / a load of stored
// coordinates): value-reduced; ZZ, ZZZ tight with |value| < 5 p.
SBV_HD void pt29_madd(xyzz& R, const apt29& q, bool neg) {
    if (R.inf) {
        R.X = q.x;
        f29_cneg(R.Y, q.y, neg);
        R.ZZ = f29_one();
        R.ZZZ = f29_one();
        R.inf = false;
        return;
    }
    // Hot path: the 32-bit-multiplier reduction (f29_reduce_x) and two fused reductions.  Bounds (units of p):
    // X1, Y1 in (-0.01, 1.01) after f29_red_q; every product |A||B| <= 28, so every reduced value lies within +-4.9;
    // P, Rr within +-5.3 (the zero filter covers +-16); X3 before f29_red_q within +-19, Y3 within +-5.
    fe29 U2, S2, P, Rr, PP, PPP, Q, t, V;
    f29_mulx(U2, q.x, R.ZZ);
    f29_mulx(S2, q.y, R.ZZZ);
    f29_sub(P, U2, R.X);
    f29_cneg(S2, S2, neg);
    f29_sub(Rr, S2, R.Y);
    f29_norm(Rr, Rr);                           // -S2 - Y1 reaches -2^30 per limb: too loose for the squaring below
    if (f29_maybe_zero(P)) {                    // random data: probability 2^-24 per lane
        if (f29_is_zero_slow(P)) {
            if (f29_is_zero(Rr)) {              // P == Q
                fe29 qy;
                f29_cneg(qy, q.y, neg);
                pt29_mdbl(R, q.x, qy);
            } else {
                pt29_set_inf(R);                // P == -Q
            }
            return;
        }
    }
    f29_sqrx(PP, P);
    f29_mulx(PPP, P, PP);
    f29_mulx(Q, R.X, PP);
    f29_cols c;
    f29_cols_zero(c);
    f29_cols_sqr(c, Rr);
    f29_add(V, PPP, Q);
    f29_add(V, V, Q);
    f29_cols_sub_val(c, V);
    fe29 X3;
    f29_reduce_x(X3, c);                        // X3 = Rr^2 - PPP - 2 Q, one reduction
    f29_red_q(X3);
    f29_sub(t, Q, X3);
    f29_neg(V, R.Y);
    f29_cols_zero(c);
    f29_cols_mul(c, Rr, t);
    f29_cols_mul(c, V, PPP);
    f29_reduce_x(R.Y, c);                       // Y3 = Rr (Q - X3) - Y1 PPP, one reduction
    f29_red_q(R.Y);
    R.X = X3;
    f29_mulx(R.ZZ, R.ZZ, PP);
    f29_mulx(R.ZZZ, R.ZZZ, PPP);
}

// ---- general XYZZ doubling and addition (the lanes-per-signature butterfly of the latency kernel) ------------------------
// dbl-2008-s-1 with a = -3: M = 3 (X - ZZ)(X + ZZ).  R finite or infinity (infinity stays infinity).
SBV_HD void pt29_dbl(xyzz& R) {
    if (R.inf) return;
    fe29 U, V, W, S, M, t1, t2, X3, Y3;
    f29_add(U, R.Y, R.Y);
    f29_norm(U, U);
    f29_sqr(V, U);
    f29_mul(W, U, V);
    f29_mul(S, R.X, V);
    f29_sub(t1, R.X, R.ZZ);
    f29_add(t2, R.X, R.ZZ);
    f29_norm(t2, t2);
    f29_mul(M, t1, t2);
    f29_add(t1, M, M);
    f29_add(M, M, t1);
    f29_norm(M, M);
    f29_sqr(t1, M);
    f29_sub(t1, t1, S);
    f29_sub(t1, t1, S);
    f29_norm_red(X3, t1);                       // X3 = M^2 - 2 S
    f29_sub(t1, S, X3);
    f29_mul(t1, M, t1);
    f29_mul(Y3, W, R.Y);
    f29_sub(Y3, t1, Y3);
    f29_norm_red(R.Y, Y3);                      // Y3 = M (S - X3) - W Y1
    R.X = X3;
    f29_mul(R.ZZ, V, R.ZZ);
    f29_mul(R.ZZZ, W, R.ZZZ);
}
// R += Q, both XYZZ (add-2008-s, 12M + 2S), exact for every input: either operand may be infinity, R == Q doubles,
// R == -Q gives infinity.  Coordinates as the comb phases leave them (X, Y value-reduced; ZZ, ZZZ within +-5 p).
SBV_HD void pt29_add(xyzz& R, const xyzz& Q) {
    if (Q.inf) return;
    if (R.inf) { R = Q; return; }
    fe29 U1, U2, S1, S2, P, Rr, PP, PPP, Qv, t, V;
    f29_mulx(U1, R.X, Q.ZZ);
    f29_mulx(U2, Q.X, R.ZZ);
    f29_mulx(S1, R.Y, Q.ZZZ);
    f29_mulx(S2, Q.Y, R.ZZZ);
    f29_sub(P, U2, U1);
    f29_norm_red(P, P);                         // +-8.4 p -> [0, p): keeps |A||B| of the products below in range
    f29_sub(Rr, S2, S1);
    f29_norm_red(Rr, Rr);
    if (f29_maybe_zero(P)) {
        if (f29_is_zero_slow(P)) {
            if (f29_is_zero(Rr)) pt29_dbl(R);
            else pt29_set_inf(R);
            return;
        }
    }
    f29_sqrx(PP, P);
    f29_mulx(PPP, P, PP);
    f29_mulx(Qv, U1, PP);
    f29_cols c;
    f29_cols_zero(c);
    f29_cols_sqr(c, Rr);
    f29_add(V, PPP, Qv);
    f29_add(V, V, Qv);
    f29_cols_sub_val(c, V);
    fe29 X3;
    f29_reduce_x(X3, c);
    f29_red_q(X3);
    f29_sub(t, Qv, X3);
    f29_neg(V, S1);
    f29_cols_zero(c);
    f29_cols_mul(c, Rr, t);
    f29_cols_mul(c, V, PPP);
    f29_reduce_x(R.Y, c);
    f29_red_q(R.Y);
    R.X = X3;
    f29_mulx(t, R.ZZ, Q.ZZ);
    f29_mulx(R.ZZ, t, PP);
    f29_mulx(t, R.ZZZ, Q.ZZZ);
    f29_mulx(R.ZZZ, t, PPP);
}

// R.x mod N == r  <=>  R != infinity and (X == r ZZ  or  (r + N < p and X == (r + N) ZZ))  (mod p); r < N plain.
SBV_HD bool pt29_rx_matches(const xyzz& R, const u256& r) {
    if (R.inf) return false;
    fe29 rM, t;
    f29_from_plain(rM, r);
    f29_mul(t, rM, R.ZZ);
    f29_sub(t, t, R.X);
    bool match = f29_is_zero(t);
    const sc n_ = sc_n();
    const fe p_ = fe_p();
    u256 rn;
    const u32 carry = add256(rn, r, n_);
    if (carry == 0 && lt256(rn, p_)) {          // only for r < p - N ~ 2^128: essentially never
        f29_from_plain(rM, rn);
        f29_mul(t, rM, R.ZZ);
        f29_sub(t, t, R.X);
        match = match || f29_is_zero(t);
    }
    return match;
}

// ---- Jacobian doubling (the 256-doubling chain that turns a fresh public key into its comb bases) --------------------
// dbl-2001-b for a = -3: 3M + 5S.  X, Y, Z in / out: value-reduced (f29_norm_red).  Z = 0 (mod p) stays 0 (mod p).
struct jpt29 { fe29 X, Y, Z; };
SBV_HD void pt29_dbl_jac(jpt29& R) {
    fe29 delta, gamma, beta, alpha, t1, t2, g2;
    f29_sqr(delta, R.Z);
    f29_sqr(gamma, R.Y);
    f29_mul(beta, R.X, gamma);
    f29_sub(t1, R.X, delta);
    f29_add(t2, R.X, delta);
    f29_norm(t2, t2);
    f29_mul(alpha, t1, t2);
    f29_add(t1, alpha, alpha);
    f29_add(alpha, alpha, t1);                  // alpha = 3 (X - delta)(X + delta): limbs < 3 * 2^29
    f29_norm(alpha, alpha);
    f29_add(t1, R.Y, R.Z);
    f29_norm(t1, t1);
    f29_sqr(t1, t1);
    f29_sub(t1, t1, gamma);
    f29_sub(t1, t1, delta);
    f29_norm_red(R.Z, t1);                      // Z3 = (Y + Z)^2 - gamma - delta
    f29_add(t1, beta, beta);
    f29_add(t1, t1, t1);
    f29_norm(beta, t1);                         // 4 beta
    f29_sqr(t1, alpha);
    f29_sub(t1, t1, beta);
    f29_sub(t1, t1, beta);
    f29_norm_red(R.X, t1);                      // X3 = alpha^2 - 8 beta
    f29_sub(t1, beta, R.X);
    f29_mul(t1, alpha, t1);
    f29_sqr(g2, gamma);
    f29_add(t2, g2, g2);
    f29_add(t2, t2, t2);
    f29_norm(t2, t2);                           // 4 gamma^2
    f29_sub(t1, t1, t2);
    f29_sub(t1, t1, t2);
    f29_norm_red(R.Y, t1);                      // Y3 = alpha (4 beta - X3) - 8 gamma^2
}

// The same doubling for the per-signature chain of the all-distinct-keys kernel (256 of them per signature): the
// 32-bit-multiplier reduction and fused reductions for X3 and Y3 (p256_fe29.h, column-level interface).  `inf` stays.
// Bounds (units of p): X, Y value-reduced, Z within +-5; alpha within +-13 before its product is formed, every
// |A||B| <= 64.
struct jpt29f { fe29 X, Y, Z; bool inf; };
SBV_HD void pt29_dbl_jacx(jpt29f& R) {
    fe29 delta, gamma, beta, alpha, t1, t2, g2, V;
    f29_sqrx(delta, R.Z);
    f29_sqrx(gamma, R.Y);
    f29_mulx(beta, R.X, gamma);
    f29_sub(t1, R.X, delta);
    f29_add(t2, R.X, delta);
    f29_norm(t2, t2);
    f29_mulx(alpha, t1, t2);
    f29_red_q(alpha);
    f29_add(t1, alpha, alpha);
    f29_add(alpha, alpha, t1);                  // alpha = 3 (X - delta)(X + delta), value within (0, 3.1)
    f29_norm(alpha, alpha);
    f29_add(t1, R.Y, R.Z);
    f29_norm(t1, t1);
    f29_cols c;
    f29_cols_zero(c);
    f29_cols_sqr(c, t1);
    f29_add(V, gamma, delta);
    f29_cols_sub_val(c, V);
    f29_reduce_x(R.Z, c);                       // Z3 = (Y + Z)^2 - gamma - delta
    f29_red_q(R.Z);
    f29_add(t1, beta, beta);
    f29_add(t1, t1, t1);
    f29_norm(beta, t1);
    f29_red_q(beta);                            // 4 beta, value-reduced
    f29_cols_zero(c);
    f29_cols_sqr(c, alpha);
    f29_add(V, beta, beta);
    f29_cols_sub_val(c, V);
    f29_reduce_x(R.X, c);                       // X3 = alpha^2 - 8 beta
    f29_red_q(R.X);
    f29_sub(t1, beta, R.X);
    f29_sqrx(g2, gamma);
    f29_red_q(g2);
    f29_add(t2, g2, g2);
    f29_norm(t2, t2);
    f29_add(V, t2, t2);                         // 4 gamma^2 (limbs < 2^30 + 16)
    f29_cols_zero(c);
    f29_cols_mul(c, alpha, t1);
    f29_cols_sub_val(c, V);
    f29_cols_sub_val(c, V);
    f29_reduce_x(R.Y, c);                       // Y3 = alpha (4 beta - X3) - 8 gamma^2
    f29_red_q(R.Y);
}

// R += (q.x, +-q.y) for a Jacobian accumulator (madd-2007-bl shape: 8M + 3S), exact like pt29_madd.
SBV_HD void pt29_madd_jacx(jpt29f& R, const apt29& q, bool neg) {
    if (R.inf) {
        R.X = q.x;
        f29_cneg(R.Y, q.y, neg);
        R.Z = f29_one();
        R.inf = false;
        return;
    }
    fe29 zz, U2, S2, H, Rr, HH, HHH, Vv, t, W;
    f29_sqrx(zz, R.Z);
    f29_mulx(U2, q.x, zz);
    f29_mulx(t, R.Z, zz);
    f29_mulx(S2, q.y, t);
    f29_sub(H, U2, R.X);
    f29_cneg(S2, S2, neg);
    f29_sub(Rr, S2, R.Y);
    f29_norm(Rr, Rr);
    if (f29_maybe_zero(H)) {
        if (f29_is_zero_slow(H)) {
            if (f29_is_zero(Rr)) {              // P == Q: the doubling of the accumulator itself
                pt29_dbl_jacx(R);
            } else {
                R.X = f29_zero(); R.Y = f29_zero(); R.Z = f29_zero(); R.inf = true;
            }
            return;
        }
    }
    f29_sqrx(HH, H);
    f29_mulx(HHH, H, HH);
    f29_mulx(Vv, R.X, HH);
    f29_cols c;
    f29_cols_zero(c);
    f29_cols_sqr(c, Rr);
    f29_add(W, HHH, Vv);
    f29_add(W, W, Vv);
    f29_cols_sub_val(c, W);
    fe29 X3;
    f29_reduce_x(X3, c);
    f29_red_q(X3);
    f29_sub(t, Vv, X3);
    f29_neg(W, R.Y);
    f29_cols_zero(c);
    f29_cols_mul(c, Rr, t);
    f29_cols_mul(c, W, HHH);
    f29_reduce_x(R.Y, c);
    f29_red_q(R.Y);
    R.X = X3;
    f29_mulx(R.Z, R.Z, H);
}

// affine + affine -> affine given the inverse of (x2 - x1):  lambda = (y2 - y1) / (x2 - x1),
// x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1.  2M + 1S.  Only for x1 != x2 (distinct small multiples of a
// point of prime order — the table builder's case).  Inputs tight / canonical; outputs value-reduced.
SBV_HD void apt29_add_with_inverse(apt29& r, const apt29& a, const apt29& b, const fe29& dinv) {
    fe29 t, lam, x3;
    f29_sub(t, b.y, a.y);
    f29_mul(lam, t, dinv);
    f29_sqr(t, lam);
    f29_sub(t, t, a.x);
    f29_sub(t, t, b.x);
    f29_norm_red(x3, t);
    f29_sub(t, a.x, x3);
    f29_mul(t, lam, t);
    f29_sub(t, t, a.y);
    f29_norm_red(r.y, t);
    r.x = x3;
}

// table entry (64 bytes: x | y, each the canonical 8-word residue of this domain) -> affine point
SBV_HD void apt29_load(apt29& q, const u32* src) {
    struct alignas(16) q4 { u32 x, y, z, w; };
    const q4* s = reinterpret_cast<const q4*>(src);
    const q4 a = s[0], b = s[1], c = s[2], d = s[3];
    const u32 wx[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const u32 wy[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    f29_unpack(q.x, wx);
    f29_unpack(q.y, wy);
}

}  // namespace sbv
