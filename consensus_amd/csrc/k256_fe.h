// k256_fe.h — GF(p), p = 2^256 - 2^32 - 977 (secp256k1), for the "other curves" variant of the hot path (SURVEY.md §8f row 4:
// api.Signer / api.Verifier are curve-agnostic, pkg/api/dependencies.go:46-71).
//
// Same idea as p256_fe29.h — nine signed 29-bit limbs, products as 81 independent 64-bit multiply-accumulates
// (v_mad_i64_i32), no carry chains inside a product — but this prime is a pseudo-Mersenne number, so values stay in the
// plain domain (no Montgomery factor) and the reduction is a fold:
//     2^261 = 32 * 2^256 = 32 * (2^32 + 977) = 2^37 + 31264  (mod p)  ->  limb k + 9 adds 31264 to limb k and 256 to limb k + 1
//     2^256 = 2^32 + 977                                      (mod p)  ->  bit 24 of limb 8 upwards adds 977 to limb 0 and 8 to limb 1
//
// Contract.  "Reduced": limbs 0..7 in [0, 2^29), limb 8 in [-2^20, 2^24 + 2^20]; the value is then in (-2^253, 2^256 + 2^253),
// so it is congruent to 0 iff it IS 0 or p (kfe_is_zero tests exactly that).  Everything that carries (kfe_mul, kfe_sqr,
// kfe_add, kfe_sub, kfe_lin*, kfe_scale, kfe_carry*) returns reduced values.  The *_nc forms (no carry) return the limb-wise
// sum or difference: signed limbs, not reduced, good as operands of ONE multiplication and nothing else.  A product's 17
// columns are sums of nine 64-bit terms, so kfe_mul / kfe_sqr require 9 * max|a_i| * max|b_j| < 2^62.9: reduced x reduced,
// difference x difference (|limb| < 2^29), or (3 x reduced) x reduced.  SBV_K256_CHECK (emulator builds) turns both rules
// into assertions.  On gfx950 a 64-bit multiply-accumulate issues almost as fast as a 32-bit add, so what the point formulas
// save is carries, not products (k256_core.h).
//
// Shared host/device source (tests/emul compiles it with g++).
#pragma once
#include "modinv30.h"
#include "sbv_common.h"

#if defined(SBV_K256_CHECK)
#include <cstdio>
#include <cstdlib>
#endif

namespace sbv {

typedef int32_t i32;
typedef int64_t i64;

struct kfe { i32 v[9]; };

#define SBV_KM29 0x1FFFFFFF
#define SBV_K_P0 0x1FFFFC2F          // p mod 2^29

SBV_HD kfe kfe_zero() { kfe r = {{0, 0, 0, 0, 0, 0, 0, 0, 0}}; return r; }
SBV_HD kfe kfe_one() { kfe r = {{1, 0, 0, 0, 0, 0, 0, 0, 0}}; return r; }
SBV_HD kfe kfe_p() { kfe r = {{0x1FFFFC2F, 0x1FFFFFF7, 0x1FFFFFFF, 0x1FFFFFFF, 0x1FFFFFFF, 0x1FFFFFFF, 0x1FFFFFFF, 0x1FFFFFFF, 0x00FFFFFF}}; return r; }
SBV_HD u256 k256_p_words() { u256 r = {{0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}}; return r; }
SBV_HD modinfo30 modinfo30_k256_p() {
    modinfo30 r = {{{0x3FFFFC2F, 0x3FFFFFFB, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0xFFFF}}, 0x2DDACACFu};
    return r;
}

SBV_HD void kfe_check(const kfe& a, const char* where) {
#if defined(SBV_K256_CHECK)
    bool bad = a.v[8] < -(1 << 20) || a.v[8] > (1 << 24) + (1 << 20);
    for (int i = 0; i < 8; ++i) bad = bad || a.v[i] < 0 || a.v[i] > SBV_KM29;
    if (bad) { fprintf(stderr, "k256_fe contract violated in %s\n", where); abort(); }
#else
    (void)a; (void)where;
#endif
}

// Limbs as 64-bit values (|t[i]| < 2^63 - 2^35) -> reduced.  Two carry passes around the fold of everything from bit 256 up.
SBV_HD void kfe_carry64(kfe& r, i64 t[9]) {
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= SBV_KM29; }
    const i64 top = t[8] >> 24;                 // multiples of 2^256 (may be negative)
    t[8] &= 0xFFFFFF;
    t[0] += 977 * top;
    t[1] += 8 * top;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= SBV_KM29; }
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = (i32)t[i];
    kfe_check(r, "kfe_carry64");
}

// the same on 32-bit limbs (|t[i]| < 2^31 - 2^26): for sums and differences of a few reduced values
SBV_HD void kfe_carry32(kfe& r, const kfe& a) {
    i32 t[9];
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) t[i] = a.v[i];
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= SBV_KM29; }
    const i32 top = t[8] >> 24;
    t[8] &= 0xFFFFFF;
    t[0] += 977 * top;
    t[1] += 8 * top;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= SBV_KM29; }
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = t[i];
    kfe_check(r, "kfe_carry32");
}
SBV_HD void kfe_add_nc(kfe& r, const kfe& a, const kfe& b) {
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
}
SBV_HD void kfe_sub_nc(kfe& r, const kfe& a, const kfe& b) {
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] - b.v[i];
}
SBV_HD void kfe_cneg_nc(kfe& r, const kfe& a, bool neg) {
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = neg ? -a.v[i] : a.v[i];
}
SBV_HD void kfe_add(kfe& r, const kfe& a, const kfe& b) {
    kfe t;
    kfe_add_nc(t, a, b);
    kfe_carry32(r, t);
}
SBV_HD void kfe_sub(kfe& r, const kfe& a, const kfe& b) {
    kfe t;
    kfe_sub_nc(t, a, b);
    kfe_carry32(r, t);
}
// r = a * k for k = 2, 3 (32-bit)
SBV_HD void kfe_scale(kfe& r, const kfe& a, int k) {
    kfe t;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) t.v[i] = a.v[i] * k;
    kfe_carry32(r, t);
}
// r = a - b * kb - c * kc for small non-negative constants (64-bit)
SBV_HD void kfe_lin3(kfe& r, const kfe& a, const kfe& b, int kb, const kfe& c, int kc) {
    i64 t[9];
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) t[i] = (i64)a.v[i] - (i64)b.v[i] * kb - (i64)c.v[i] * kc;
    kfe_carry64(r, t);
}
// r = a * k - b * m for small non-negative constants (k, m <= 16)
SBV_HD void kfe_lin(kfe& r, const kfe& a, int k, const kfe& b, int m) {
    i64 t[9];
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) t[i] = (i64)a.v[i] * k - (i64)b.v[i] * m;
    kfe_carry64(r, t);
}
SBV_HD void kfe_mul_small(kfe& r, const kfe& a, int k) {
    i64 t[9];
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) t[i] = (i64)a.v[i] * k;
    kfe_carry64(r, t);
}
SBV_HD void kfe_cneg(kfe& r, const kfe& a, bool neg) {
    i64 t[9];
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) t[i] = neg ? -(i64)a.v[i] : (i64)a.v[i];
    kfe_carry64(r, t);
}
SBV_HD void kfe_select(kfe& r, bool c, const kfe& a, const kfe& b) {
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = c ? a.v[i] : b.v[i];
}

// 17 product columns -> reduced.  |c[k]| < 2^62.9 (kfe_check_product).  The high columns are NOT carried through one another
// (a 16-step chain of 64-bit shifts and adds): each is split on its own into its low 29 bits and the rest, position 9 + k
// then holds lo(c[9+k]) + hi(c[8+k]) < 2^35, and that folds into positions k and k + 1.  One carry pass pair (kfe_carry64)
// at the end does all the propagation.
SBV_HD void kfe_reduce(kfe& r, i64 c[17]) {
    i64 h[9];                                   // positions 9..17
    i64 prev_hi = 0;
    SBV_UNROLL
    for (int k = 0; k < 8; ++k) {
        const i64 hi = c[9 + k] >> 29;
        h[k] = (c[9 + k] & SBV_KM29) + prev_hi;
        prev_hi = hi;
    }
    h[8] = prev_hi;                             // |h[8]| < 2^34
    i64 t[9];
    SBV_UNROLL
    for (int k = 0; k < 9; ++k) t[k] = c[k] + 31264 * h[k] + (k == 0 ? 0 : 256 * h[k - 1]);
    const i64 t9 = 256 * h[8];                  // position 9 once more; |t9| < 2^42
    t[0] += 31264 * t9;
    t[1] += 256 * t9;
    kfe_carry64(r, t);                          // |t[k]| < 2^62.9 + 2^57: the carries added on the way stay below 2^63
}

SBV_HD void kfe_check_product(const kfe& a, const kfe& b, const char* where) {
#if defined(SBV_K256_CHECK)
    i64 ma = 0, mb = 0;
    for (int i = 0; i < 9; ++i) {
        const i64 x = a.v[i] < 0 ? -(i64)a.v[i] : (i64)a.v[i], y = b.v[i] < 0 ? -(i64)b.v[i] : (i64)b.v[i];
        if (x > ma) ma = x;
        if (y > mb) mb = y;
    }
    const long double bound = 9.0L * (long double)ma * (long double)mb;
    if (bound >= 8.6e18L) { fprintf(stderr, "k256_fe product bound violated in %s: 9 * %lld * %lld\n", where, (long long)ma, (long long)mb); abort(); }
#else
    (void)a; (void)b; (void)where;
#endif
}
SBV_HD void kfe_mul(kfe& r, const kfe& a, const kfe& b) {
    kfe_check_product(a, b, "kfe_mul");
    i64 c[17];
    SBV_UNROLL
    for (int k = 0; k < 17; ++k) c[k] = 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        SBV_UNROLL
        for (int j = 0; j < 9; ++j) c[i + j] += (i64)a.v[i] * b.v[j];
    }
    kfe_reduce(r, c);
}
SBV_HD void kfe_sqr(kfe& r, const kfe& a) {
    kfe_check_product(a, a, "kfe_sqr");
    i64 c[17];
    SBV_UNROLL
    for (int k = 0; k < 17; ++k) c[k] = 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        c[2 * i] += (i64)a.v[i] * a.v[i];
        SBV_UNROLL
        for (int j = i + 1; j < 9; ++j) c[i + j] += 2 * ((i64)a.v[i] * a.v[j]);
    }
    kfe_reduce(r, c);
}

// canonical 256-bit words of a reduced value (the value in [0, p))
SBV_HD void kfe_to_words(u256& w, const kfe& a) {
    kfe_check(a, "kfe_to_words");
    // value in (-2^253, 2^256 + 2^253): add p once so that it is positive, then subtract p up to three times
    i64 t[9];
    const kfe p = kfe_p();
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) t[i] = (i64)a.v[i] + p.v[i];
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= SBV_KM29; }
    // now limbs 0..7 in [0, 2^29), t[8] >= 0 holds the rest (< 2^26): pack into 9 32-bit words (value < 2^258)
    u32 x[9];
    SBV_UNROLL
    for (int k = 0; k < 9; ++k) x[k] = 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, wd = bit >> 5, sh = bit & 31;
        const u64 v = (u64)t[i] << sh;
        x[wd] |= (u32)v;
        if (wd + 1 < 9) x[wd + 1] |= (u32)(v >> 32);
    }
    const u256 pw = k256_p_words();
    SBV_UNROLL
    for (int rep = 0; rep < 3; ++rep) {
        u32 d[9], bw = 0;
        SBV_UNROLL
        for (int k = 0; k < 8; ++k) d[k] = subb(x[k], pw.v[k], bw);
        d[8] = subb(x[8], 0u, bw);
        const bool ge = bw == 0;
        SBV_UNROLL
        for (int k = 0; k < 9; ++k) x[k] = ge ? d[k] : x[k];
    }
    SBV_UNROLL
    for (int k = 0; k < 8; ++k) w.v[k] = x[k];
}
// any 256-bit integer (not necessarily < p) -> reduced limbs of the same residue: the nine bit fields ARE reduced limbs
SBV_HD void kfe_from_words(kfe& r, const u256& w) {
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, wd = bit >> 5, sh = bit & 31;
        u32 v = w.v[wd] >> sh;
        if (sh > 3 && wd + 1 < 8) v |= w.v[wd + 1] << (32 - sh);
        r.v[i] = (i32)(v & (i == 8 ? 0xFFFFFFu : (u32)SBV_KM29));
    }
    kfe_check(r, "kfe_from_words");
}

// exact zero test of a reduced value: the value is 0 or p
SBV_HD bool kfe_maybe_zero(const kfe& a) { return a.v[0] == 0 || a.v[0] == SBV_K_P0; }
SBV_HD bool kfe_is_zero_slow(const kfe& a) {
    const kfe p = kfe_p();
    i32 z = 0, e = 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) { z |= a.v[i]; e |= a.v[i] ^ p.v[i]; }
    return z == 0 || e == 0;
}
SBV_HD bool kfe_is_zero(const kfe& a) {
    kfe_check(a, "kfe_is_zero");
    return kfe_maybe_zero(a) && kfe_is_zero_slow(a);
}
// d = a - b (kfe_sub_nc of two reduced values): false means d is certainly not 0 mod p.  |d| < 1.25 * 2^256, so d = 0 mod p
// iff d is 0, p or -p, whose low 29 bits are 0, p mod 2^29 and 2^29 - p mod 2^29 = 977.
SBV_HD bool kfe_diff_maybe_zero(const kfe& d) {
    const i32 lo = d.v[0] & SBV_KM29;
    return lo == 0 || lo == SBV_K_P0 || lo == 977;
}
SBV_HD bool kfe_equal(const kfe& a, const kfe& b) {
    kfe d;
    kfe_sub(d, a, b);
    return kfe_is_zero(d);
}

// a^-1 (0 -> 0): division steps on the canonical words (modinv30.h works on plain integers, which is what this field holds)
SBV_HD void kfe_inv(kfe& r, const kfe& a) {
    u256 w, iw;
    kfe_to_words(w, a);
    modinv30(iw, w, modinfo30_k256_p());
    kfe_from_words(r, iw);
}

}  // namespace sbv
