// ed25519_fe.h — arithmetic in GF(2^255 - 19) for the Ed25519 verifier variant
// (BASELINE.json configs[4]; Go: crypto/internal/edwards25519/field).
//
// Representation: 8 x 32-bit little-endian limbs, *weakly reduced*: any value in [0, 2^256) stands
// for its residue mod p (2^256 = 38 mod p, so a carry out of 256 bits folds back as +38).  Only
// fe25_freeze produces the canonical representative, and only comparisons / encoding need it.
// Products are the same row-wise v_mad_u64_u32 chains as GF(p256) (p256_fe.h: mul_wide / sqr_wide);
// the reduction is 8 more multiply-adds by the constant 38.
#pragma once
#include "p256_fe.h"   // mul_wide, sqr_wide, addc/subb helpers

namespace sbv {

typedef u256 fe25;

SBV_HD fe25 fe25_zero() { fe25 r = {{0, 0, 0, 0, 0, 0, 0, 0}}; return r; }
SBV_HD fe25 fe25_one() { fe25 r = {{1, 0, 0, 0, 0, 0, 0, 0}}; return r; }
// d = -121665/121666, 2d, sqrt(-1)
SBV_HD fe25 fe25_d() { fe25 r = {{0x135978A3u, 0x75EB4DCAu, 0x4141D8ABu, 0x00700A4Du, 0x7779E898u, 0x8CC74079u, 0x2B6FFE73u, 0x52036CEEu}}; return r; }
SBV_HD fe25 fe25_2d() { fe25 r = {{0x26B2F159u, 0xEBD69B94u, 0x8283B156u, 0x00E0149Au, 0xEEF3D130u, 0x198E80F2u, 0x56DFFCE7u, 0x2406D9DCu}}; return r; }
SBV_HD fe25 fe25_sqrtm1() { fe25 r = {{0x4A0EA0B0u, 0xC4EE1B27u, 0xAD2FE478u, 0x2F431806u, 0x3DFBD7A7u, 0x2B4D0099u, 0x4FC1DF0Bu, 0x2B832480u}}; return r; }

// fold a carry word c (value c * 2^256 = c * 38 mod p) into t; result < 2^256
SBV_HD void fe25_fold(fe25& r, const fe25& t, u32 c) {
    // c <= 2^32 / 38 is required so that c * 38 fits one limb; callers guarantee c < 2^26
    u32 cy = 0;
    r.v[0] = addc(t.v[0], c * 38u, cy);
    SBV_UNROLL
    for (int i = 1; i < 8; ++i) r.v[i] = addc(t.v[i], 0u, cy);
    // a second carry can only happen when the sum wrapped, leaving a value < c*38: no third fold
    r.v[0] += cy * 38u;
}

SBV_HD void fe25_add(fe25& r, const fe25& a, const fe25& b) {
    fe25 t;
    const u32 c = add256(t, a, b);
    fe25_fold(r, t, c);
}
SBV_HD void fe25_sub(fe25& r, const fe25& a, const fe25& b) {
    fe25 t;
    const u32 bw = sub256(t, a, b);          // t = a - b + bw * 2^256, and 2^256 = 38: subtract 38 * bw
    u32 b2 = 0;
    r.v[0] = subb(t.v[0], bw * 38u, b2);
    SBV_UNROLL
    for (int i = 1; i < 8; ++i) r.v[i] = subb(t.v[i], 0u, b2);
    r.v[0] -= b2 * 38u;                       // wrapped once more: value is >= 2^256 - 38, no third step
}
SBV_HD void fe25_neg(fe25& r, const fe25& a) { const fe25 z = fe25_zero(); fe25_sub(r, z, a); }
SBV_HD void fe25_cneg(fe25& r, const fe25& a, bool neg) { fe25 n; fe25_neg(n, a); select256(r, neg, n, a); }

// T (512 bits) -> T_lo + 38 * T_hi, weakly reduced
SBV_HD void fe25_reduce_wide(fe25& r, const u32 t[16]) {
    fe25 lo;
    u64 q = 0;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) {
        q = (u64)t[8 + i] * 38u + t[i] + (q >> 32);     // <= 38*(2^32-1) + 2*(2^32-1) < 2^64
        lo.v[i] = (u32)q;
    }
    fe25_fold(r, lo, (u32)(q >> 32));                    // carry <= 39
}
SBV_HD void fe25_mul(fe25& r, const fe25& a, const fe25& b) {
    u32 t[16];
    mul_wide(t, a.v, b.v);
    fe25_reduce_wide(r, t);
}
SBV_HD void fe25_sqr(fe25& r, const fe25& a) {
    u32 t[16];
    sqr_wide(t, a.v);
    fe25_reduce_wide(r, t);
}

// canonical representative in [0, p)
SBV_HD void fe25_freeze(fe25& r, const fe25& a) {
    fe25 t = a;
    SBV_UNROLL
    for (int k = 0; k < 2; ++k) {            // t = (t mod 2^255) + 19 * (t >> 255), twice
        const u32 top = t.v[7] >> 31;
        t.v[7] &= 0x7FFFFFFFu;
        u32 c = 0;
        t.v[0] = addc(t.v[0], top * 19u, c);
        SBV_UNROLL
        for (int i = 1; i < 8; ++i) t.v[i] = addc(t.v[i], 0u, c);
    }
    // t <= 2^255 - 1 + 19 < 2p: subtract p = 2^255 - 19 once if t >= p
    fe25 d;
    u32 bw = 0;
    d.v[0] = subb(t.v[0], 0xFFFFFFEDu, bw);
    SBV_UNROLL
    for (int i = 1; i < 7; ++i) d.v[i] = subb(t.v[i], 0xFFFFFFFFu, bw);
    d.v[7] = subb(t.v[7], 0x7FFFFFFFu, bw);
    select256(r, bw == 0, d, t);
}
SBV_HD bool fe25_eq(const fe25& a, const fe25& b) {
    fe25 x, y;
    fe25_freeze(x, a);
    fe25_freeze(y, b);
    return eq256(x, y);
}
SBV_HD bool fe25_is_zero(const fe25& a) { fe25 x; fe25_freeze(x, a); return is_zero256(x); }
SBV_HD bool fe25_is_negative(const fe25& a) { fe25 x; fe25_freeze(x, a); return (x.v[0] & 1u) != 0; }

// z^(2^252 - 3) = z^((p-5)/8)   (the classic 2^k-1 ladder: 251 squarings, 11 multiplications)
SBV_HD void fe25_pow22523(fe25& out, const fe25& z) {
    fe25 t0, t1, t2;
    fe25_sqr(t0, z);                                              // 2
    fe25_sqr(t1, t0); fe25_sqr(t1, t1);                           // 8
    fe25_mul(t1, z, t1);                                          // 9
    fe25_mul(t0, t0, t1);                                         // 11
    fe25_sqr(t0, t0);                                             // 22
    fe25_mul(t0, t1, t0);                                         // 31 = 2^5 - 1
    fe25_sqr(t1, t0); for (int i = 1; i < 5; ++i) fe25_sqr(t1, t1);
    fe25_mul(t0, t1, t0);                                         // 2^10 - 1
    fe25_sqr(t1, t0); for (int i = 1; i < 10; ++i) fe25_sqr(t1, t1);
    fe25_mul(t1, t1, t0);                                         // 2^20 - 1
    fe25_sqr(t2, t1); for (int i = 1; i < 20; ++i) fe25_sqr(t2, t2);
    fe25_mul(t1, t2, t1);                                         // 2^40 - 1
    fe25_sqr(t1, t1); for (int i = 1; i < 10; ++i) fe25_sqr(t1, t1);
    fe25_mul(t0, t1, t0);                                         // 2^50 - 1
    fe25_sqr(t1, t0); for (int i = 1; i < 50; ++i) fe25_sqr(t1, t1);
    fe25_mul(t1, t1, t0);                                         // 2^100 - 1
    fe25_sqr(t2, t1); for (int i = 1; i < 100; ++i) fe25_sqr(t2, t2);
    fe25_mul(t1, t2, t1);                                         // 2^200 - 1
    fe25_sqr(t1, t1); for (int i = 1; i < 50; ++i) fe25_sqr(t1, t1);
    fe25_mul(t0, t1, t0);                                         // 2^250 - 1
    fe25_sqr(t0, t0); fe25_sqr(t0, t0);                           // 2^252 - 4
    fe25_mul(out, t0, z);                                         // 2^252 - 3
}
// z^(p-2) = z^(2^255 - 21): (z^(2^252-3))^8 * z^3 = z^(2^255 - 24 + 3)
SBV_HD void fe25_inv(fe25& out, const fe25& z) {
    fe25 t, z2, z3;
    fe25_pow22523(t, z);
    fe25_sqr(t, t); fe25_sqr(t, t); fe25_sqr(t, t);               // 2^255 - 24
    fe25_sqr(z2, z);
    fe25_mul(z3, z2, z);
    fe25_mul(out, t, z3);
}

// 32 little-endian bytes given as 8 little-endian dwords -> field element (bit 255 cleared by the caller)
// The same inverse by division steps (modinv30.h): the input is frozen to [0, p) first; plain integers, so
// no domain conversion.  0 -> 0, like z^(p-2).
SBV_HD void fe25_inv_gcd(fe25& out, const fe25& z) {
    fe25 t;
    fe25_freeze(t, z);
    modinv30(out, t, modinfo30_25519());
}

SBV_HD void fe25_from_words(fe25& r, const u32 w[8]) {
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) r.v[i] = w[i];
}

}  // namespace sbv
