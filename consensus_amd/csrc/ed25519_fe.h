// ed25519_fe.h — arithmetic in GF(2^255 - 19) for the Ed25519 verifier variant
// (BASELINE.json configs[4]; Go: crypto/internal/edwards25519/field), carry-free form.
//
// Why this form: on gfx950 only plain 32-bit VALU operations issue at full rate; v_add_co / v_addc chains issue at
// roughly half rate AND serialise on the carry (profiles/r01/microbench.jsonl).  The round-1 field (8 x 32-bit words,
// weakly reduced below 2^256) spent most of an addition's ~14.5 k cycles per wavefront in such chains: every field
// add/sub was a 17-step dependent carry chain, every product a row-wise carry chain (profiles/r02: the comb phases ran
// at 97 us per addition per 2^20 lanes, against 65 us for the wider P-256 mixed addition on the carry-free P-256 field).
//
// Representation: 10 signed limbs, limb i at bit ceil(25.5 i) (26, 25, 26, 25, ... bits wide); the value is
// sum v[i] * 2^ceil(25.5 i) and is NOT canonical.  2^255 = 19 (mod p) folds the upper half of a product straight into
// the lower half, inside the multiply-accumulates: a product is 100 independent v_mad_i64_i32 into ten 64-bit columns
// (a square 55), followed by ONE carry pass.  Additions and subtractions are 10 independent 32-bit operations.
//
// Contracts (checked at run time in the emulator build, -DSBV_F25_CHECK):
//   "tight"  |v[i]| <= 1.01 * 2^25 (i even) / 2^24 (i odd): what fe25_mul / fe25_sqr / fe25_carry return
//   fe25_mul(a, b): a up to 8x tight, b up to 3x tight (b's limbs are multiplied by 19 in 32 bits)
//   fe25_sqr(a):    a up to 3x tight
//   fe25_add / fe25_sub / fe25_neg: no carry; bounds add up
// Only fe25_freeze produces the canonical representative; comparisons and encodings go through it.
#pragma once
#include "modinv30.h"
#include "p256_fe.h"   // u256, select / compare helpers
#ifdef SBV_F25_CHECK
#include <stdio.h>
#include <stdlib.h>
#endif

namespace sbv {

typedef int32_t i32;
typedef int64_t i64;

struct fe25 { i32 v[10]; };

#ifdef SBV_F25_CHECK
static inline void f25_check(const fe25& a, int mult, const char* what) {
    for (int i = 0; i < 10; ++i) {
        const i64 lim = (i64)mult * (((i & 1) ? (1 << 24) : (1 << 25)) + (1 << 19));
        if (a.v[i] > lim || a.v[i] < -lim) { fprintf(stderr, "fe25 contract: %s limb %d = %d exceeds %d x tight\n", what, i, a.v[i], mult); abort(); }
    }
}
#define SBV_F25_CHECK_OP(a, mult, what) f25_check(a, mult, what)
#else
#define SBV_F25_CHECK_OP(a, mult, what) ((void)0)
#endif

SBV_HD fe25 fe25_zero() { fe25 r = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}; return r; }
SBV_HD fe25 fe25_one() { fe25 r = {{1, 0, 0, 0, 0, 0, 0, 0, 0, 0}}; return r; }
// d = -121665/121666, 2d, sqrt(-1): canonical values cut into limbs (tests/test_emul_fe25.py recomputes them)
SBV_HD fe25 fe25_d() { fe25 r = {{0x35978A3, 0x0D37284, 0x3156EBD, 0x06A0A0E, 0x001C029, 0x179E898, 0x3A03CBB, 0x1CE7198, 0x2E2B6FF, 0x1480DB3}}; return r; }
SBV_HD fe25 fe25_2d() { fe25 r = {{0x2B2F159, 0x1A6E509, 0x22ADD7A, 0x0D4141D, 0x0038052, 0x0F3D130, 0x3407977, 0x19CE331, 0x1C56DFF, 0x0901B67}}; return r; }
SBV_HD fe25 fe25_sqrtm1() { fe25 r = {{0x20EA0B0, 0x186C9D2, 0x08F189D, 0x035697F, 0x0BD0C60, 0x1FBD7A7, 0x2804C9E, 0x1E16569, 0x004FC1D, 0x0AE0C92}}; return r; }

SBV_HD void fe25_add(fe25& r, const fe25& a, const fe25& b) {
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) r.v[i] = a.v[i] + b.v[i];
}
SBV_HD void fe25_sub(fe25& r, const fe25& a, const fe25& b) {
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) r.v[i] = a.v[i] - b.v[i];
}
SBV_HD void fe25_neg(fe25& r, const fe25& a) {
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) r.v[i] = -a.v[i];
}
SBV_HD void fe25_select(fe25& r, bool c, const fe25& a, const fe25& b) {      // r = c ? a : b
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) r.v[i] = c ? a.v[i] : b.v[i];
}
SBV_HD void fe25_cneg(fe25& r, const fe25& a, bool neg) {
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) r.v[i] = neg ? -a.v[i] : a.v[i];
}

SBV_HD i64 f25_mad(i32 a, i32 b, i64 c) { return (i64)a * (i64)b + c; }
SBV_HD i64 f25_madu(u32 a, u32 b, i64 c) { return (i64)((u64)c + (u64)a * (u64)b); }
SBV_HD i64 f25_shl(i64 x, int k) { return (i64)((u64)x << k); }
#if defined(__HIP_DEVICE_COMPILE__)
// A constant the optimiser cannot see through: keeps "x * 64" a v_mad_i64_i32 with the constant in an SGPR instead of a
// 64-bit shift-and-add sequence (three half-rate instructions and a carry chain).
__device__ __forceinline__ i32 f25_opaque(i32 k) { asm("" : "+s"(k)); return k; }
#else
SBV_HD i32 f25_opaque(i32 k) { return k; }
#endif

SBV_HD int f25_width(int i) { return (i & 1) ? 25 : 26; }
SBV_HD int f25_pos(int i) { return (51 * i + 1) / 2; }              // ceil(25.5 i)
// Rounding offset of column i: with 2^(w-1) added to a column the floor carry below becomes a rounding one and the limb is
// (low w bits) - 2^(w-1), balanced around zero.
SBV_HD i64 f25_round(int i) { return (i64)1 << (f25_width(i) - 1); }

// One carry pass over ten 64-bit columns: 0 -> 1 -> ... -> 9 -> (x 19) -> 0 -> 1.
// A column h = hi * 2^32 + lo sends floor(h / 2^w) = hi * 2^(32-w) + (lo >> w) up: two multiply-accumulates by constants
// (v_mad_i64_i32 / v_mad_u64_u32 issue faster than the add-with-carry pairs of a 64-bit addition and need no 64-bit
// shift), and keeps lo mod 2^w.  The rounding offset of column k+1 rides on the carry out of column k:
// 2^(w[k+1]-1) = 2^18 * 2^(32-w[k]) for every k, so it is one 32-bit addition to hi.  Every limb ends tight.
SBV_HD void f25_carry_cols(fe25& r, i64 h[10]) {
    const i32 k64 = f25_opaque(64), k128 = f25_opaque(128), k19x128 = f25_opaque(19 * 128);
    const u32 k1 = (u32)f25_opaque(1), k19 = (u32)f25_opaque(19);
    u32 m[10];
    h[0] += f25_round(0);
    SBV_UNROLL
    for (int k = 0; k < 9; ++k) {
        const u32 lo = (u32)h[k];
        const i32 hi = (i32)(h[k] >> 32);
        const int w = f25_width(k);
        h[k + 1] = f25_mad(hi + (1 << 18), w == 26 ? k64 : k128, h[k + 1]);
        h[k + 1] = f25_madu(lo >> w, k1, h[k + 1]);
        m[k] = lo & ((1u << w) - 1u);
    }
    i64 h0 = (i64)m[0];                                   // still carries its rounding offset
    {
        const u32 lo = (u32)h[9];
        const i32 hi = (i32)(h[9] >> 32);
        h0 = f25_mad(hi, k19x128, h0);                    // 2^255 = 19
        h0 = f25_madu(lo >> 25, k19, h0);
        m[9] = lo & ((1u << 25) - 1u);
    }
    const u32 lo0 = (u32)h0;
    const i32 hi0 = (i32)(h0 >> 32);                      // |19 * carry| < 2^42: the second carry out of limb 0 is a 32-bit number
    const i32 c0 = hi0 * 64 + (i32)(lo0 >> 26);
    r.v[0] = (i32)(lo0 & ((1u << 26) - 1u)) - (1 << 25);
    r.v[1] = (i32)m[1] - (1 << 24) + c0;
    SBV_UNROLL
    for (int i = 2; i < 10; ++i) r.v[i] = (i32)m[i] - (i32)f25_round(i);
}
// any limbs within the 32-bit range -> tight
SBV_HD void fe25_carry(fe25& r, const fe25& a) {
    i64 h[10];
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) h[i] = (i64)a.v[i];
    f25_carry_cols(r, h);
}

SBV_HD void fe25_mul(fe25& r, const fe25& a, const fe25& b) {
    SBV_F25_CHECK_OP(a, 8, "mul a");
    SBV_F25_CHECK_OP(b, 3, "mul b");
    i32 b19[10], a2[10];
    SBV_UNROLL
    for (int j = 0; j < 10; ++j) { b19[j] = 19 * b.v[j]; a2[j] = 2 * a.v[j]; }
    i64 h[10];
    SBV_UNROLL
    for (int k = 0; k < 10; ++k) h[k] = 0;
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) {
        SBV_UNROLL
        for (int j = 0; j < 10; ++j) {
            const i32 x = ((i & 1) && (j & 1)) ? a2[i] : a.v[i];      // two odd limbs meet one bit above their column
            const i32 y = (i + j >= 10) ? b19[j] : b.v[j];            // 2^255 = 19
            const int k = (i + j) % 10;
            h[k] = f25_mad(x, y, h[k]);
        }
    }
    f25_carry_cols(r, h);
}
SBV_HD void fe25_sqr(fe25& r, const fe25& a) {
    SBV_F25_CHECK_OP(a, 3, "sqr a");
    i32 a2[10], a19[10], a38[10];
    SBV_UNROLL
    for (int j = 0; j < 10; ++j) { a2[j] = 2 * a.v[j]; a19[j] = 19 * a.v[j]; a38[j] = (j & 1) ? 38 * a.v[j] : 0; }    // 38 x only fits for the 25-bit limbs
    i64 h[10];
    SBV_UNROLL
    for (int k = 0; k < 10; ++k) h[k] = 0;
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) {
        SBV_UNROLL
        for (int j = i; j < 10; ++j) {
            // coefficient = (i == j ? 1 : 2) * (both odd ? 2 : 1) * (wrap ? 19 : 1), split over the two factors
            const bool odd2 = (i & 1) && (j & 1), wrap = i + j >= 10;
            const i32 x = (i == j) ? a.v[i] : a2[i];
            const i32 y = wrap ? (odd2 ? a38[j] : a19[j]) : (odd2 ? a2[j] : a.v[j]);
            const int k = (i + j) % 10;
            h[k] = f25_mad(x, y, h[k]);
        }
    }
    f25_carry_cols(r, h);
}

// ---- canonical form -------------------------------------------------------------------------------------------------------

// tight (or up to ~8x tight) limbs -> the canonical residue in [0, p) as exact non-negative limbs
SBV_HD void fe25_canon_limbs(i32 out[10], const fe25& a) {
    i64 h[10];
    // + 8p keeps every limb positive for |v[i]| < 4 * 2^w: p = (2^26 - 19, 2^25 - 1, 2^26 - 1, ...)
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) h[i] = (i64)a.v[i] + 8 * (((i64)1 << f25_width(i)) - 1);
    h[0] -= 8 * 18;
    SBV_NOUNROLL
    for (int pass = 0; pass < 2; ++pass) {                           // floor carries; the top carry re-enters as 19
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) {
            const i64 cr = h[i] >> f25_width(i);
            h[i] -= f25_shl(cr, f25_width(i));
            h[i + 1] += cr;
        }
        const i64 top = h[9] >> 25;
        h[9] -= f25_shl(top, 25);
        h[0] += 19 * top;
    }
    // Pass 1 leaves value < 2^255 + 19 * 16; if pass 2 still carries out of bit 255 the rest is < 2^10, so its + 19 cannot
    // carry again: all limbs are exact now and 0 <= value < 2^255.  value >= p  <=>  value + 19 >= 2^255.
    i64 t[10];
    t[0] = h[0] + 19;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        const i64 cr = t[i] >> f25_width(i);
        t[i] -= f25_shl(cr, f25_width(i));
        t[i + 1] = h[i + 1] + cr;
    }
    const bool ge = (t[9] >> 25) != 0;
    t[9] &= ((i64)1 << 25) - 1;
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) out[i] = (i32)(ge ? t[i] : h[i]);
}
// canonical residue as 8 little-endian 32-bit words
SBV_HD void fe25_freeze(u256& w, const fe25& a) {
    i32 c[10];
    fe25_canon_limbs(c, a);
    SBV_UNROLL
    for (int k = 0; k < 8; ++k) w.v[k] = 0;
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) {
        const int pos = f25_pos(i), word = pos >> 5, sh = pos & 31;
        const u64 piece = (u64)(u32)c[i] << sh;
        w.v[word] |= (u32)piece;
        if (word + 1 < 8) w.v[word + 1] |= (u32)(piece >> 32);
    }
}
// 8 little-endian words, bit 255 ignored -> limbs in [0, 2^w): 2x tight, a valid `b` operand of fe25_mul
SBV_HD void fe25_from_words(fe25& r, const u32 w[8]) {
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) {
        const int pos = f25_pos(i), word = pos >> 5, sh = pos & 31, wid = f25_width(i);
        u32 bits = w[word] >> sh;                                    // 32-bit pieces only: no 64-bit shifts, no paired loads
        if (sh + wid > 32 && word + 1 < 8) bits |= w[word + 1] << (32 - sh);
        r.v[i] = (i32)(bits & ((1u << wid) - 1u));
    }
}
SBV_HD bool fe25_eq(const fe25& a, const fe25& b) {
    u256 x, y;
    fe25_freeze(x, a);
    fe25_freeze(y, b);
    return eq256(x, y);
}
SBV_HD bool fe25_is_zero(const fe25& a) { u256 x; fe25_freeze(x, a); return is_zero256(x); }
SBV_HD bool fe25_is_negative(const fe25& a) { u256 x; fe25_freeze(x, a); return (x.v[0] & 1u) != 0; }

// ---- storage ----------------------------------------------------------------------------------------------------------------
// tables: the canonical residue, 32 bytes (two 16-byte vectors); scratch between kernels: the ten raw limbs
struct alignas(16) f25_q4 { u32 x, y, z, w; };
SBV_HD void fe25_store_packed(u32* dst, const fe25& a) {
    u256 w;
    fe25_freeze(w, a);
    f25_q4* d = reinterpret_cast<f25_q4*>(dst);
    const f25_q4 lo = {w.v[0], w.v[1], w.v[2], w.v[3]}, hi = {w.v[4], w.v[5], w.v[6], w.v[7]};
    d[0] = lo;
    d[1] = hi;
}
SBV_HD void fe25_load_packed(fe25& a, const u32* src) {
    const f25_q4* s = reinterpret_cast<const f25_q4*>(src);
    const f25_q4 lo = s[0], hi = s[1];
    const u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    fe25_from_words(a, w);
}
#define SBV_F25_RAW_WORDS 10
SBV_HD void fe25_store_raw(u32* dst, const fe25& a) {
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) dst[i] = (u32)a.v[i];
}
SBV_HD void fe25_load_raw(fe25& a, const u32* src) {
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) a.v[i] = (i32)src[i];
}

// z^(2^252 - 3) = z^((p-5)/8)   (the classic 2^k-1 ladder: 251 squarings, 11 multiplications); z tight
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
// z^(p-2) = z^(2^255 - 21): (z^(2^252-3))^8 * z^3
SBV_HD void fe25_inv(fe25& out, const fe25& z) {
    fe25 t, z2, z3;
    fe25_pow22523(t, z);
    fe25_sqr(t, t); fe25_sqr(t, t); fe25_sqr(t, t);               // 2^255 - 24
    fe25_sqr(z2, z);
    fe25_mul(z3, z2, z);
    fe25_mul(out, t, z3);
}
// The same inverse by division steps (modinv30.h) on the canonical integer; 0 -> 0, like z^(p-2).
SBV_HD void fe25_inv_gcd(fe25& out, const fe25& z) {
    u256 t, y;
    fe25_freeze(t, z);
    modinv30(y, t, modinfo30_25519());
    fe25_from_words(out, y.v);
}

}  // namespace sbv
