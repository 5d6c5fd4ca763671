// p256_pt.h — P-256 group law on Jacobian coordinates (X:Y:Z), x = X/Z^2, y = Y/Z^3, Z = 0 is
// the point at infinity.  Every routine is EXACT for every input (infinity, P == Q, P == -Q):
// crypto/ecdsa's verdict on adversarial signatures (u1*G == +-u2*Q) depends on
// it.  The P == Q case is the only one the generic addition formulas get wrong silently, so it
// is detected (H == 0 && r == 0) and routed through a doubling; that branch is wavefront-
// divergent only for the lanes that hit it and costs nothing otherwise (s_cbranch_execz).
//
// Formulas: dbl-2001-b (a = -3; 3M + 5S) and add-2007-bl-style Jacobian addition with cached
// Z2^2, Z2^3 (11M + 3S), mixed addition with Z2 = 1 (8M + 3S).  [Explicit-Formulas Database]
#pragma once
#include "p256_fe.h"

namespace sbv {

struct jpt { fe X, Y, Z; };
struct apt { fe x, y; };                 // affine, Montgomery form; 64 bytes
struct qent { fe X, Y, Z, ZZ, ZZZ; };    // per-signature table entry; 160 bytes

SBV_HD void pt_set_inf(jpt& p) { p.X = fe_zero(); p.Y = fe_zero(); p.Z = fe_zero(); }
SBV_HD bool pt_is_inf(const jpt& p) { return fe_is_zero(p.Z); }

// r = 2p.  Z = 0 stays Z = 0 (Z3 = (Y+Z)^2 - Y^2 - Z^2).  Finite points have Y != 0 (prime order).
template <bool FAST = false>
SBV_HD void pt_dbl(jpt& r, const jpt& p, u32* st = nullptr) {
    fe delta, gamma, beta, alpha, t1, t2;
    fe_sqr<FAST>(delta, p.Z, st);
    fe_sqr<FAST>(gamma, p.Y, st);
    fe_mul<FAST>(beta, p.X, gamma, st);
    fe_sub(t1, p.X, delta);
    fe_add<FAST>(t2, p.X, delta, st);
    fe_mul<FAST>(alpha, t1, t2, st);
    fe_dbl<FAST>(t1, alpha, st);
    fe_add<FAST>(alpha, alpha, t1, st);            // alpha = 3 (X - delta)(X + delta)
    fe_add<FAST>(t1, p.Y, p.Z, st);
    fe_sqr<FAST>(t1, t1, st);
    fe_sub(t1, t1, gamma);
    fe_sub(r.Z, t1, delta);              // Z3 = (Y + Z)^2 - gamma - delta
    fe_dbl<FAST>(beta, beta, st);
    fe_dbl<FAST>(beta, beta, st);                  // 4 beta
    fe_sqr<FAST>(t1, alpha, st);
    fe_sub(t1, t1, beta);
    fe_sub(r.X, t1, beta);               // X3 = alpha^2 - 8 beta
    fe_sub(t1, beta, r.X);
    fe_mul<FAST>(t1, alpha, t1, st);
    fe_sqr<FAST>(t2, gamma, st);
    fe_dbl<FAST>(t2, t2, st);
    fe_dbl<FAST>(t2, t2, st);
    fe_dbl<FAST>(t2, t2, st);                      // 8 gamma^2
    fe_sub(r.Y, t1, t2);                 // Y3 = alpha (4 beta - X3) - 8 gamma^2
}

// The P == Q fallback.  With SBV_NOINLINE_FALLBACK the (never taken on honest data) doubling is an
// out-of-line call instead of an inlined copy inside every addition: smaller code, lower register
// pressure in the additions (experiment knob; see profiles/).
#if defined(SBV_NOINLINE_FALLBACK) && defined(__HIP_DEVICE_COMPILE__)
template <bool FAST>
__device__ __noinline__ void pt_dbl_cold(jpt& r, const jpt& p, u32* st) { pt_dbl<FAST>(r, p, st); }
#define SBV_PT_DBL_COLD(F, r, p, st) pt_dbl_cold<F>(r, p, st)
#else
#define SBV_PT_DBL_COLD(F, r, p, st) pt_dbl<F>(r, p, st)
#endif

// R += (q.x, q.y) with y negated when `neg`; no-op when `skip` (window digit 0).
template <bool FAST = false>
SBV_HD void pt_add_mixed(jpt& R, const apt& q, bool neg, bool skip, u32* st = nullptr) {
    fe qy, z1z1, u2, s2, h, rr, hh, hhh, v, t;
    fe_cneg(qy, q.y, neg);
    const bool p_inf = fe_is_zero(R.Z);
    fe_sqr<FAST>(z1z1, R.Z, st);
    fe_mul<FAST>(u2, q.x, z1z1, st);
    fe_mul<FAST>(t, R.Z, z1z1, st);
    fe_mul<FAST>(s2, qy, t, st);
    fe_sub(h, u2, R.X);
    fe_sub(rr, s2, R.Y);
    const bool same = !p_inf && fe_is_zero(h) && fe_is_zero(rr);   // P == Q
    jpt g;
    fe_sqr<FAST>(hh, h, st);
    fe_mul<FAST>(hhh, h, hh, st);
    fe_mul<FAST>(v, R.X, hh, st);
    fe_sqr<FAST>(t, rr, st);
    fe_sub(t, t, hhh);
    fe_sub(t, t, v);
    fe_sub(g.X, t, v);                   // X3 = r^2 - H^3 - 2 X1 H^2
    fe_sub(t, v, g.X);
    fe_mul<FAST>(t, rr, t, st);
    fe_mul<FAST>(v, R.Y, hhh, st);
    fe_sub(g.Y, t, v);                   // Y3 = r (X1 H^2 - X3) - Y1 H^3
    fe_mul<FAST>(g.Z, R.Z, h, st);                 // Z3 = Z1 H   (0 when P == -Q: infinity, as it must be)
    if (same) SBV_PT_DBL_COLD(FAST, g, R, st);
    const fe one = fe_one();
    const bool take_q = p_inf;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) {
        const u32 nx = take_q ? q.x.v[i] : g.X.v[i];
        const u32 ny = take_q ? qy.v[i] : g.Y.v[i];
        const u32 nz = take_q ? one.v[i] : g.Z.v[i];
        R.X.v[i] = skip ? R.X.v[i] : nx;
        R.Y.v[i] = skip ? R.Y.v[i] : ny;
        R.Z.v[i] = skip ? R.Z.v[i] : nz;
    }
}

// R += (q.X : q.Y : q.Z) (finite, with cached ZZ = Z^2, ZZZ = Z^3), y negated when `neg`;
// no-op when `skip`.
template <bool FAST = false>
SBV_HD void pt_add_qent(jpt& R, const qent& q, bool neg, bool skip, u32* st = nullptr) {
    fe qy, z1z1, u1, u2, s1, s2, h, rr, hh, hhh, v, t;
    fe_cneg(qy, q.Y, neg);
    const bool p_inf = fe_is_zero(R.Z);
    fe_sqr<FAST>(z1z1, R.Z, st);
    fe_mul<FAST>(u1, R.X, q.ZZ, st);
    fe_mul<FAST>(u2, q.X, z1z1, st);
    fe_mul<FAST>(s1, R.Y, q.ZZZ, st);
    fe_mul<FAST>(t, R.Z, z1z1, st);
    fe_mul<FAST>(s2, qy, t, st);
    fe_sub(h, u2, u1);
    fe_sub(rr, s2, s1);
    const bool same = !p_inf && fe_is_zero(h) && fe_is_zero(rr);
    jpt g;
    fe_sqr<FAST>(hh, h, st);
    fe_mul<FAST>(hhh, h, hh, st);
    fe_mul<FAST>(v, u1, hh, st);
    fe_sqr<FAST>(t, rr, st);
    fe_sub(t, t, hhh);
    fe_sub(t, t, v);
    fe_sub(g.X, t, v);
    fe_sub(t, v, g.X);
    fe_mul<FAST>(t, rr, t, st);
    fe_mul<FAST>(v, s1, hhh, st);
    fe_sub(g.Y, t, v);
    fe_mul<FAST>(t, R.Z, q.Z, st);
    fe_mul<FAST>(g.Z, t, h, st);
    if (same) SBV_PT_DBL_COLD(FAST, g, R, st);
    const bool take_q = p_inf;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) {
        const u32 nx = take_q ? q.X.v[i] : g.X.v[i];
        const u32 ny = take_q ? qy.v[i] : g.Y.v[i];
        const u32 nz = take_q ? q.Z.v[i] : g.Z.v[i];
        R.X.v[i] = skip ? R.X.v[i] : nx;
        R.Y.v[i] = skip ? R.Y.v[i] : ny;
        R.Z.v[i] = skip ? R.Z.v[i] : nz;
    }
}

// y^2 == x^3 - 3x + b  (Montgomery-form inputs)
template <bool FAST = false>
SBV_HD bool pt_on_curve(const fe& x, const fe& y, u32* st = nullptr) {
    fe lhs, rhs, t;
    fe_sqr<FAST>(lhs, y, st);
    fe_sqr<FAST>(t, x, st);
    fe_mul<FAST>(rhs, t, x, st);
    fe_sub(rhs, rhs, x);
    fe_sub(rhs, rhs, x);
    fe_sub(rhs, rhs, x);
    const fe b = fe_b_mont();
    fe_add<FAST>(rhs, rhs, b, st);
    return fe_eq(lhs, rhs);
}

}  // namespace sbv
