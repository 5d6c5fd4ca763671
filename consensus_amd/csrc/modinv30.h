// modinv30.h — modular inversion by Bernstein–Yang "safegcd" division steps, 30 bits at a time.
//
// Why: Fermat inversion is a chain of ~265 (mod p) / ~350 (mod N) dependent multiplications, i.e.
// 50–140 k VALU instructions on the critical path of stage A (one inversion per lane), of the
// per-batch key-table windows and of every latency-bound micro-batch.  Division steps need 20 rounds of
// (30 branch-free steps on the low words + two small matrix-vector products on 9 x 30-bit signed
// limbs): ~15 k instructions, all of them 32-bit adds/shifts and v_mad_i64_i32.
//
// Algorithm: D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation and modular inversion"
// (TCHES 2019), in the half-delta variant (zeta = -(delta + 1/2), 590 steps suffice for 256-bit inputs;
// 20 x 30 = 600 are run).  Constant time is not the point here (everything verified is public) —
// uniform control flow across a wavefront is: every lane executes the same 600 steps.
// Written from the paper's description; the limb layout (signed 30-bit limbs in 32-bit words, 64-bit
// accumulators) is the natural one for v_mad_i64_i32.  Prior art with the same shape — signed-30 limbs, a 2x2
// transition matrix per 30 steps, divsteps / update_de / update_fg / normalize, the half-delta counter and the
// 590 -> 600 step bound — is libsecp256k1's modinv32 (src/modinv32_impl.h, Pieter Wuille et al.), which also
// documents the bound; this file owes its structure to that reading of the paper.
//
// Contract: modinv30(out, x, mi): x in [0, m), m odd, m < 2^256  ->  out = x^-1 mod m in [0, m);
// x = 0 gives 0 (like x^(m-2)).  Plain integers, no Montgomery form; see fe_inv_gcd / sc_inv_gcd for
// the Montgomery-domain wrappers.
#pragma once
#include "sbv_common.h"

namespace sbv {

struct s30 { int32_t v[9]; };                       // value = sum v[i] * 2^(30 i)
struct modinfo30 { s30 m; u32 m_inv30; };           // modulus and modulus^-1 mod 2^30
struct trans30 { int32_t u, v, q, r; };             // 2x2 transition matrix of 30 division steps, scaled by 2^30

#define SBV_M30 0x3FFFFFFF

SBV_HD void s30_from_u256(s30& r, const u256& a) {
    const u32* w = a.v;
    r.v[0] = (int32_t)(w[0] & SBV_M30);
    r.v[1] = (int32_t)(((w[0] >> 30) | (w[1] << 2)) & SBV_M30);
    r.v[2] = (int32_t)(((w[1] >> 28) | (w[2] << 4)) & SBV_M30);
    r.v[3] = (int32_t)(((w[2] >> 26) | (w[3] << 6)) & SBV_M30);
    r.v[4] = (int32_t)(((w[3] >> 24) | (w[4] << 8)) & SBV_M30);
    r.v[5] = (int32_t)(((w[4] >> 22) | (w[5] << 10)) & SBV_M30);
    r.v[6] = (int32_t)(((w[5] >> 20) | (w[6] << 12)) & SBV_M30);
    r.v[7] = (int32_t)(((w[6] >> 18) | (w[7] << 14)) & SBV_M30);
    r.v[8] = (int32_t)(w[7] >> 16);
}
// a normalised: limbs 0..7 in [0, 2^30), value in [0, 2^256)
SBV_HD void s30_to_u256(u256& r, const s30& a) {
    const u32 v0 = (u32)a.v[0], v1 = (u32)a.v[1], v2 = (u32)a.v[2], v3 = (u32)a.v[3], v4 = (u32)a.v[4], v5 = (u32)a.v[5],
              v6 = (u32)a.v[6], v7 = (u32)a.v[7], v8 = (u32)a.v[8];
    r.v[0] = v0 | (v1 << 30);
    r.v[1] = (v1 >> 2) | (v2 << 28);
    r.v[2] = (v2 >> 4) | (v3 << 26);
    r.v[3] = (v3 >> 6) | (v4 << 24);
    r.v[4] = (v4 >> 8) | (v5 << 22);
    r.v[5] = (v5 >> 10) | (v6 << 20);
    r.v[6] = (v6 >> 12) | (v7 << 18);
    r.v[7] = (v7 >> 14) | (v8 << 16);
}

// 30 division steps on the low words of f (odd) and g.  Returns the new zeta; t = the matrix such that
// [f', g'] = t * [f, g] / 2^30.  Branch-free: the two conditions become masks.
SBV_HD int32_t divsteps30(int32_t zeta, u32 f0, u32 g0, trans30& t) {
    u32 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    SBV_UNROLL
    for (int i = 0; i < 30; ++i) {
        u32 c1 = (u32)(zeta >> 31);                  // all ones when zeta < 0 (delta > 0)
        const u32 c2 = 0u - (g & 1u);                // all ones when g is odd
        const u32 x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // (f, u, v) negated when zeta < 0
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;                                    // swap case: zeta < 0 and g odd
        zeta = (int32_t)((u32)zeta ^ c1) - 1;        // zeta -> -zeta - 2, or zeta - 1
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}

// The same 30 division steps, variable time: a run of k even values of g is k steps that only halve g and double (u, v), so the
// run is taken in ONE iteration (count trailing zeros); what is left are the ~15 steps with g odd.  Step for step the same
// sequence as divsteps30 — same matrix, same zeta — in about half the instructions; lanes of a wavefront take different
// numbers of iterations (SIMT divergence: the wavefront runs the longest lane's count, ~17-19 instead of 30).  Everything
// this library inverts while VERIFYING is public; the signing kernel keeps the constant-time form.
SBV_HD int32_t divsteps30_var(int32_t zeta, u32 f0, u32 g0, trans30& t) {
    u32 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    int i = 30;
    for (;;) {
        const int zeros = __builtin_ctz(g | (1u << i));      // at most i: bit i is set
        g >>= zeros; u <<= zeros; v <<= zeros;
        zeta -= zeros;
        i -= zeros;
        if (i == 0) break;
        // g is odd
        if (zeta < 0) {                                      // delta > 0: (f, g) <- (g, g - f), rows of the matrix likewise
            const u32 nf = g, nu = q, nv = r;
            g -= f; q -= u; r -= v;
            f = nf; u = nu; v = nv;
            zeta = -zeta - 2;
        } else {
            g += f; q += u; r += v;
            zeta -= 1;
        }
        g >>= 1; u <<= 1; v <<= 1;
        --i;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}

// [d, e] <- t * [d, e] / 2^30 (mod m), both kept in (-2m, m): a multiple of m is added so that the low
// 30 bits vanish before the exact shift.
SBV_HD void update_de30(s30& d, s30& e, const trans30& t, const modinfo30& mi) {
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (t.u & sd) + (t.v & se);
    int32_t me = (t.q & sd) + (t.r & se);
    int64_t cd = u * d.v[0] + v * e.v[0];
    int64_t ce = q * d.v[0] + r * e.v[0];
    md -= (int32_t)((mi.m_inv30 * (u32)cd + (u32)md) & SBV_M30);
    me -= (int32_t)((mi.m_inv30 * (u32)ce + (u32)me) & SBV_M30);
    cd += (int64_t)mi.m.v[0] * md;
    ce += (int64_t)mi.m.v[0] * me;
    cd >>= 30;
    ce >>= 30;
    SBV_UNROLL
    for (int i = 1; i < 9; ++i) {
        cd += u * d.v[i] + v * e.v[i];
        ce += q * d.v[i] + r * e.v[i];
        cd += (int64_t)mi.m.v[i] * md;
        ce += (int64_t)mi.m.v[i] * me;
        d.v[i - 1] = (int32_t)cd & SBV_M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & SBV_M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}

// [f, g] <- t * [f, g] / 2^30 (exact)
SBV_HD void update_fg30(s30& f, s30& g, const trans30& t) {
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    int64_t cf = u * f.v[0] + v * g.v[0];
    int64_t cg = q * f.v[0] + r * g.v[0];
    cf >>= 30;
    cg >>= 30;
    SBV_UNROLL
    for (int i = 1; i < 9; ++i) {
        cf += u * f.v[i] + v * g.v[i];
        cg += q * f.v[i] + r * g.v[i];
        f.v[i - 1] = (int32_t)cf & SBV_M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & SBV_M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}

// r in (-2m, m) -> [0, m), negated first when sign < 0
SBV_HD void normalize30(s30& r, int32_t sign, const modinfo30& mi) {
    int32_t c = r.v[8] >> 31;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] += mi.m.v[i] & c;
    c = sign >> 31;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = (r.v[i] ^ c) - c;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= SBV_M30; }
    c = r.v[8] >> 31;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] += mi.m.v[i] & c;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= SBV_M30; }
}

// CT = true: 20 x 30 branch-free steps whatever the input (the signing kernel's nonce inversion).  CT = false (default): the
// variable-time steps above, and the loop ends as soon as g is 0 — once it is, further rounds leave f and d as they are
// (their matrix is [[2^30, 0], [0, 1]]) — typically after 17-18 of the 20 rounds.  Same result bit for bit.
template <bool CT>
SBV_HD void modinv30_t(u256& out, const u256& x, const modinfo30& mi) {
    s30 d, e, f, g;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) { d.v[i] = 0; e.v[i] = 0; f.v[i] = mi.m.v[i]; }
    e.v[0] = 1;
    s30_from_u256(g, x);
    int32_t zeta = -1;
    SBV_NOUNROLL
    for (int it = 0; it < 20; ++it) {
        trans30 t;
        zeta = CT ? divsteps30(zeta, (u32)f.v[0], (u32)g.v[0], t) : divsteps30_var(zeta, (u32)f.v[0], (u32)g.v[0], t);
        update_de30(d, e, t, mi);
        update_fg30(f, g, t);
        if (!CT) {
            int32_t o = 0;
            SBV_UNROLL
            for (int i = 0; i < 9; ++i) o |= g.v[i];
            if (o == 0) break;
        }
    }
    // g = 0 and f = +-gcd(m, x) = +-1 now (x != 0); d = +-x^-1
    normalize30(d, f.v[8], mi);
    s30_to_u256(out, d);
}
SBV_HD void modinv30(u256& out, const u256& x, const modinfo30& mi) { modinv30_t<false>(out, x, mi); }
SBV_HD void modinv30_ct(u256& out, const u256& x, const modinfo30& mi) { modinv30_t<true>(out, x, mi); }

// ---- the three moduli of this library ---------------------------------------------------------------
SBV_HD modinfo30 modinfo30_p256() {       // p = 2^256 - 2^224 + 2^192 + 2^96 - 1
    modinfo30 r = {{{0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3F, 0x0, 0x0, 0x1000, 0x3FFFC000, 0xFFFF}}, 0x3FFFFFFFu};
    return r;
}
SBV_HD modinfo30 modinfo30_p256_order() { // N, the order of the P-256 base point
    modinfo30 r = {{{0x3C632551, 0x0EE72B0B, 0x3179E84F, 0x39BEAB69, 0x3FFFFFBC, 0x3FFFFFFF, 0xFFF, 0x3FFFC000, 0xFFFF}}, 0x11FF43B1u};
    return r;
}
SBV_HD modinfo30 modinfo30_25519() {      // 2^255 - 19
    modinfo30 r = {{{0x3FFFFFED, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x7FFF}}, 0x179435E5u};
    return r;
}

}  // namespace sbv
