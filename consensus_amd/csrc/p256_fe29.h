// p256_fe29.h — GF(p) arithmetic for NIST P-256 in a carry-free representation built for gfx950's VALU.
//
// Why a second representation (p256_fe.h keeps 8 x 32-bit limbs): measured on MI355X (profiles/r01/
// microbench.jsonl) only plain 32-bit VOP1/VOP2 instructions issue at full rate (~2.35 cycles per wave
// instruction per SIMD); everything that writes or reads a carry (v_add_co / v_addc_co), every 64-bit or
// three-source instruction (v_mad_u64_u32, v_lshl_add_u64, v_add3, v_alignbit) issues at HALF rate, and a
// VALU that writes VCC must be two wait states ahead of the VALU that reads it.  In the 8 x 32 form a field
// multiplication is 64 multiplies but ~110 carry instructions and ~60 register moves (≈ 430 issue slots);
// here it is 81 + 36 multiply-accumulates that add straight into 64-bit column accumulators, no carry flag
// anywhere (≈ 330 slots), and additions / subtractions are 9 full-rate instructions with no reduction.
//
// Representation: 9 SIGNED 32-bit limbs, radix 2^29: value(x) = sum v[i] * 2^(29 i).  Elements live in the
// Montgomery domain with R = 2^261 and are NOT kept canonical:
//   "tight"   limbs 0..7 in [0, 2^29), limb 8 small and signed           (what f29_mul / f29_sqr return)
//   "loose"   any limbs with |v[i]| < 2^31 after a few additions / subtractions of tight values
// Contracts (checked by tests/emul against big integers, including the worst cases the bounds allow):
//   f29_mul(r, a, b): needs  sum_i |a.v[i]| * |b.v[k-i]| < 3 * 2^60 for every column k  and  |A| * |B| <= 64 p^2;
//                     returns tight r with value in (A*B/R, A*B/R + p): within (-p/2, 3p/2) for |A||B| <= 16 p^2 (the
//                     comb phases), within (-2p, 3p) in general (R = 32 * 2^256).
//   f29_sqr(r, a):    the same, and |a.v[i]| < 2^30 (the cross terms use 2 * a.v[j] as a 32-bit operand).
//   In practice: a tight value, or the sum / difference of two tight values, may be multiplied by another
//   such value directly; anything looser goes through f29_norm() first.
// Comparisons mod p go through f29_is_zero() (a three-instruction filter on the low limb, then the exact
// test in a branch that random data takes with probability 2^-24).
//
// Semantics served: the same as p256_fe.h — the field arithmetic under crypto/ecdsa.VerifyASN1 for P-256,
// which is what an implementation of the reference's api.Verifier runs per signature
// (pkg/api/dependencies.go:54-71).
#pragma once
#include "p256_fe.h"

namespace sbv {

typedef int32_t i32;
typedef int64_t i64;

struct fe29 { i32 v[9]; };

#define SBV_M29 0x1FFFFFFFu
// x * 2^n for a possibly negative x (a left shift of a negative value is undefined before C++20; the unsigned detour is not)
SBV_HD i32 f29_shl(i32 x, int n) { return (i32)((u32)x << n); }

// p, R mod p, R^2 mod p, b*R mod p, 2^266 mod p (8x32 Montgomery form -> this domain), 2^256 mod p (back)
SBV_HD fe29 f29_p() { fe29 r = {{0x1FFFFFFF, 0x1FFFFFFF, 0x1FFFFFFF, 0x000001FF, 0x00000000, 0x00000000, 0x00040000, 0x1FE00000, 0x00FFFFFF}}; return r; }
SBV_HD fe29 f29_one() { fe29 r = {{0x00000020, 0x00000000, 0x00000000, 0x1FFFC000, 0x1FFFFFFF, 0x1FFFFFFF, 0x1F7FFFFF, 0x03FFFFFF, 0x00000000}}; return r; }
SBV_HD fe29 f29_r2() { fe29 r = {{0x00000C00, 0x00000000, 0x1FFF0000, 0x1FDFFFFF, 0x1FBFFFFF, 0x1FFFFFFF, 0x1FFFFFFF, 0x1FFFFFFE, 0x00000013}}; return r; }
SBV_HD fe29 f29_b() { fe29 r = {{0x1897BBFB, 0x1CDF6229, 0x018486C4, 0x01732821, 0x1DAD59E0, 0x0ABF7212, 0x1A06D110, 0x17721D20, 0x008600C3}}; return r; }
SBV_HD fe29 f29_c266() { fe29 r = {{0x00000400, 0x00000000, 0x00000000, 0x1FF80000, 0x1FFFFFFF, 0x1FFFFFFF, 0x0FFFFFFF, 0x1FFFFFFF, 0x00000003}}; return r; }
SBV_HD fe29 f29_c256() { fe29 r = {{0x00000001, 0x00000000, 0x00000000, 0x1FFFFE00, 0x1FFFFFFF, 0x1FFFFFFF, 0x1FFBFFFF, 0x001FFFFF, 0x00000000}}; return r; }
SBV_HD fe29 f29_r3() { fe29 r = {{0x00050000, 0x1FF40000, 0x1EFFFFFF, 0x0DFFFFFF, 0x07FFFFFF, 0x1FFFFFFF, 0x0000000B, 0x00000010, 0x00000C00}}; return r; }   // R^3 mod p
SBV_HD fe29 f29_zero() { fe29 r = {{0, 0, 0, 0, 0, 0, 0, 0, 0}}; return r; }

// ---- the multiply-accumulate primitive ---------------------------------------------------------------
// c + a * b with a 64-bit accumulator: v_mad_i64_i32 / v_mad_u64_u32, destination = addend register pair.
// The reduction multiplies by powers of two; hipcc would strength-reduce those to 64-bit shifts plus
// v_add_co / v_addc_co pairs (carry flag, hazards), so the constants are made opaque on the device.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ i32 f29_opaque(i32 k) { asm("" : "+s"(k)); return k; }
#else
SBV_HD i32 f29_opaque(i32 k) { return k; }
#endif
SBV_HD i64 f29_mad(i32 a, i32 b, i64 c) { return c + (i64)a * (i64)b; }
SBV_HD i64 f29_madu(u32 a, u32 b, i64 c) { return (i64)((u64)c + (u64)a * (u64)b); }

// 2^(29 k) * p = 2^(29 k) * (2^256 - 2^224 + 2^192 + 2^96 - 1):  256 = 8*29 + 24, 224 = 7*29 + 21,
// 192 = 6*29 + 18, 96 = 3*29 + 9.
struct f29_consts { i32 k8, k9, k18, k21n, k24; u32 k1; };
SBV_HD f29_consts f29_load_consts() {
    f29_consts k;
    k.k8 = f29_opaque(8); k.k9 = f29_opaque(1 << 9); k.k18 = f29_opaque(1 << 18);
    k.k21n = f29_opaque(-(1 << 21)); k.k24 = f29_opaque(1 << 24); k.k1 = (u32)f29_opaque(1);
    return k;
}

// c[0..16] -> r = sum c[k] 2^(29 k) / 2^261 mod p, tight.
// Word-by-word Montgomery reduction in radix 2^29: -p^-1 = 1 (mod 2^29), so the multiplier of step k is the low
// 29 bits m of column k; adding m * p * 2^(29 k) clears them, the rest of the column moves up as the carry
// (hi * 8 + (lo >> 29): no 64-bit shift), and the four other terms of p are multiply-accumulates into columns
// k+3, k+6, k+7, k+8.
SBV_HD void f29_reduce(fe29& r, i64 c[17]) {
    const f29_consts K = f29_load_consts();
    SBV_UNROLL
    for (int k = 0; k < 9; ++k) {
        const u32 lo = (u32)c[k];
        const i32 hi = (i32)(c[k] >> 32);
        const u32 m = lo & SBV_M29;
        c[k + 1] = f29_mad(hi, K.k8, c[k + 1]);
        c[k + 1] = f29_madu(lo >> 29, K.k1, c[k + 1]);
        c[k + 3] = f29_mad((i32)m, K.k9, c[k + 3]);
        c[k + 6] = f29_mad((i32)m, K.k18, c[k + 6]);
        c[k + 7] = f29_mad((i32)m, K.k21n, c[k + 7]);
        c[k + 8] = f29_mad((i32)m, K.k24, c[k + 8]);
    }
    SBV_UNROLL
    for (int j = 9; j < 16; ++j) {
        const u32 lo = (u32)c[j];
        const i32 hi = (i32)(c[j] >> 32);
        r.v[j - 9] = (i32)(lo & SBV_M29);
        c[j + 1] = f29_mad(hi, K.k8, c[j + 1]);
        c[j + 1] = f29_madu(lo >> 29, K.k1, c[j + 1]);
    }
    r.v[7] = (i32)((u32)c[16] & SBV_M29);
    r.v[8] = (i32)(c[16] >> 29);           // |value| < 2p: the top limb fits easily
}

// Contract checks for the CPU test tier (tests/emul is compiled with -DSBV_F29_CHECK): every multiplication any
// emulated code path executes asserts the operand bounds it relies on, so a formula that feeds it looser limbs than
// analysed fails loudly in the container instead of wrapping silently on the GPU.
#if defined(SBV_F29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
}  // namespace sbv
#include <stdio.h>
#include <stdlib.h>
namespace sbv {
inline void f29_check_operands(const fe29& a, const fe29& b, bool squaring) {
    for (int k = 0; k < 17; ++k) {
        unsigned __int128 col = 0;
        for (int i = 0; i < 9; ++i) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            const i64 x = a.v[i], y = b.v[j];
            col += (unsigned __int128)(u64)(x < 0 ? -x : x) * (u64)(y < 0 ? -y : y);
        }
        if (col >= ((unsigned __int128)3 << 60)) { fprintf(stderr, "f29: column %d would exceed 3 * 2^60 (two such products may share one set of i64 columns)\n", k); abort(); }
    }
    const i64 ta = a.v[8] < 0 ? -(i64)a.v[8] : a.v[8], tb = b.v[8] < 0 ? -(i64)b.v[8] : b.v[8];
    if ((unsigned __int128)(ta + 1) * (unsigned __int128)(tb + 1) > ((unsigned __int128)1 << 54)) { fprintf(stderr, "f29: |A||B| may exceed 64 p^2\n"); abort(); }
    if (squaring)
        for (int i = 0; i < 9; ++i)
            if (a.v[i] >= (1 << 30) || a.v[i] <= -(1 << 30)) { fprintf(stderr, "f29_sqr: limb %d needs |v| < 2^30 (it is doubled)\n", i); abort(); }
}
#define SBV_F29_CHECK_OPERANDS(a, b, sq) f29_check_operands(a, b, sq)
#else
#define SBV_F29_CHECK_OPERANDS(a, b, sq) ((void)0)
#endif

SBV_HD void f29_mul(fe29& r, const fe29& a, const fe29& b) {
    SBV_F29_CHECK_OPERANDS(a, b, false);
    i64 c[17];
    SBV_UNROLL
    for (int k = 0; k < 17; ++k) c[k] = 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        SBV_UNROLL
        for (int j = 0; j < 9; ++j) c[i + j] = f29_mad(a.v[i], b.v[j], c[i + j]);
    }
    f29_reduce(r, c);
}

// 45 multiplies: the cross terms use 2 * a[j] (one full-rate shift each)
SBV_HD void f29_sqr(fe29& r, const fe29& a) {
    SBV_F29_CHECK_OPERANDS(a, a, true);
    i64 c[17];
    i32 d[9];
    SBV_UNROLL
    for (int k = 0; k < 17; ++k) c[k] = 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) d[i] = a.v[i] * 2;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        c[2 * i] = f29_mad(a.v[i], a.v[i], c[2 * i]);
        SBV_UNROLL
        for (int j = i + 1; j < 9; ++j) c[i + j] = f29_mad(a.v[i], d[j], c[i + j]);
    }
    f29_reduce(r, c);
}

// ---- column-level interface for the hot comb path: several products into ONE reduction ------------------------------
// The reduction is almost half of a multiplication (59 - 68 of its ~145 multiply-accumulates), so the mixed addition
// forms  X3 = Rr^2 - (PPP + 2 Q)  and  Y3 = Rr (Q - X3) - Y1 PPP  in the column domain and reduces each once:
// a second product accumulates into the same 17 columns (negated through its first operand), and an already-reduced
// value V enters as V * 2^261, i.e. limb i into column 9 + i.
struct f29_cols { i64 c[17]; };
SBV_HD void f29_cols_zero(f29_cols& t) {
    SBV_UNROLL
    for (int k = 0; k < 17; ++k) t.c[k] = 0;
}
SBV_HD void f29_cols_mul(f29_cols& t, const fe29& a, const fe29& b) {
    SBV_F29_CHECK_OPERANDS(a, b, false);
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        SBV_UNROLL
        for (int j = 0; j < 9; ++j) t.c[i + j] = f29_mad(a.v[i], b.v[j], t.c[i + j]);
    }
}
SBV_HD void f29_cols_sqr(f29_cols& t, const fe29& a) {
    SBV_F29_CHECK_OPERANDS(a, a, true);
    i32 d[9];
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) d[i] = a.v[i] * 2;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        t.c[2 * i] = f29_mad(a.v[i], a.v[i], t.c[2 * i]);
        SBV_UNROLL
        for (int j = i + 1; j < 9; ++j) t.c[i + j] = f29_mad(a.v[i], d[j], t.c[i + j]);
    }
}
// t -= v * 2^261   (|v.v[i]| < 2^31)
SBV_HD void f29_cols_sub_val(f29_cols& t, const fe29& v) {
    const i32 km1 = f29_opaque(-1);
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) t.c[9 + i] = f29_mad(v.v[i], km1, t.c[9 + i]);
    t.c[16] = f29_mad(v.v[8], f29_opaque(-(1 << 29)), t.c[16]);           // limb 8 sits one limb above column 16
}
// The reduction with 32-bit multipliers: step k clears the low 32 bits of column k (m = the signed low word), so the
// carry is just (hi + sign) * 8 — one multiply-accumulate instead of two and no masking.  The price is a wider result:
// sum m_k 2^(29 k) reaches +-2^263, so r lies within +-(4.01 p + |T| / R) instead of (T/R, T/R + p); callers that keep
// values across iterations follow it with f29_red_q().  Limbs 0..7 of r are exact 29-bit limbs as in f29_reduce.
SBV_HD void f29_reduce_x(fe29& r, f29_cols& t) {
    const f29_consts K = f29_load_consts();
    i64* c = t.c;
    SBV_UNROLL
    for (int k = 0; k < 9; ++k) {
        const u32 lo = (u32)c[k];
        const i32 hi = (i32)(c[k] >> 32);
        const i32 m = (i32)lo;
        const i32 cy = hi + (i32)(lo >> 31);
        c[k + 1] = f29_mad(cy, K.k8, c[k + 1]);
        c[k + 3] = f29_mad(m, K.k9, c[k + 3]);
        c[k + 6] = f29_mad(m, K.k18, c[k + 6]);
        c[k + 7] = f29_mad(m, K.k21n, c[k + 7]);
        c[k + 8] = f29_mad(m, K.k24, c[k + 8]);
    }
    SBV_UNROLL
    for (int j = 9; j < 16; ++j) {
        const u32 lo = (u32)c[j];
        const i32 hi = (i32)(c[j] >> 32);
        r.v[j - 9] = (i32)(lo & SBV_M29);
        c[j + 1] = f29_mad(hi, K.k8, c[j + 1]);
        c[j + 1] = f29_madu(lo >> 29, K.k1, c[j + 1]);
    }
    r.v[7] = (i32)((u32)c[16] & SBV_M29);
    r.v[8] = (i32)(c[16] >> 29);
}
SBV_HD void f29_mulx(fe29& r, const fe29& a, const fe29& b) {
    f29_cols t;
    f29_cols_zero(t);
    f29_cols_mul(t, a, b);
    f29_reduce_x(r, t);
}
SBV_HD void f29_sqrx(fe29& r, const fe29& a) {
    f29_cols t;
    f29_cols_zero(t);
    f29_cols_sqr(t, a);
    f29_reduce_x(r, t);
}
// value reduction of a tight element by the multiple of p its top limb indicates: |value| < 64 p  ->  value in
// (-2^231, 2^256 + 2^231), limbs 0..7 within 2^27 of [0, 2^29)
SBV_HD void f29_red_q(fe29& r) {
    const i32 q = r.v[8] >> 24;
    r.v[8] -= f29_shl(q, 24);
    r.v[7] += f29_shl(q, 21);
    r.v[6] -= f29_shl(q, 18);
    r.v[3] -= f29_shl(q, 9);
    r.v[0] += q;
}

// ---- additions: limb-wise, no carries, no reduction ------------------------------------------------------
// (CPU test tier: a limb that leaves the i32 range aborts — the sums are formed without any carry, so a formula that stacks
// one loose value too many wraps silently on the GPU; tests/emul is how that is caught.)
#if defined(SBV_F29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
inline void f29_check_limb(i64 v, const char* what) {
    if (v > 2147483647LL || v < -2147483647LL - 1) { fprintf(stderr, "f29: a limb of %s leaves the 32-bit range\n", what); abort(); }
}
#define SBV_F29_CHECK_LIMB(v, what) f29_check_limb(v, what)
#else
#define SBV_F29_CHECK_LIMB(v, what) ((void)0)
#endif
SBV_HD void f29_add(fe29& r, const fe29& a, const fe29& b) {
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) { SBV_F29_CHECK_LIMB((i64)a.v[i] + (i64)b.v[i], "a sum"); r.v[i] = a.v[i] + b.v[i]; }
}
SBV_HD void f29_sub(fe29& r, const fe29& a, const fe29& b) {
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) { SBV_F29_CHECK_LIMB((i64)a.v[i] - (i64)b.v[i], "a difference"); r.v[i] = a.v[i] - b.v[i]; }
}
SBV_HD void f29_neg(fe29& r, const fe29& a) {
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = -a.v[i];
}
// r = neg ? -a : a        ((x ^ m) - m with m = 0 / -1)
SBV_HD void f29_cneg(fe29& r, const fe29& a, bool neg) {
    const i32 m = neg ? -1 : 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = (a.v[i] ^ m) - m;
}
SBV_HD void f29_select(fe29& r, bool c, const fe29& a, const fe29& b) {
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = c ? a.v[i] : b.v[i];
}
// One parallel carry step: limbs 0..7 back into [0, 2^29 + 4) (for inputs with |v[i]| < 2^31), the value
// unchanged.  Three full-rate instructions per limb, no dependency chain.
SBV_HD void f29_norm(fe29& r, const fe29& a) {
    i32 c[8];
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) c[i] = a.v[i] >> 29;
    r.v[0] = a.v[0] & (i32)SBV_M29;
    SBV_UNROLL
    for (int i = 1; i < 8; ++i) r.v[i] = (a.v[i] & (i32)SBV_M29) + c[i - 1];
    r.v[8] = a.v[8] + c[7];
}

// ---- canonical form and comparisons (rare: once per signature, or behind the filter below) ---------------------
// any value in (-16p, 16p), limbs |v[i]| < 2^31  ->  the representative in [0, p), exact 29-bit limbs
SBV_HD void f29_canon(fe29& r, const fe29& a) {
    const fe29 P = f29_p();
    i64 v[9];                                        // 64-bit working limbs: a 2^31 - 1 limb plus a carry must not wrap
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) v[i] = a.v[i];
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { v[i + 1] += v[i] >> 29; v[i] &= (i64)SBV_M29; }
    // q = floor(value / 2^256) is within one of floor(value / p): subtract q * p, then fix up by at most one p each way
    const i64 q = v[8] >> 24;                         // |q| <= 16; p = 2^256 - 2^224 + 2^192 + 2^96 - 1 limb by limb
    v[8] -= q * (1 << 24); v[7] += q * (1 << 21); v[6] -= q * (1 << 18); v[3] -= q * (1 << 9); v[0] += q;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { v[i + 1] += v[i] >> 29; v[i] &= (i64)SBV_M29; }
    SBV_NOUNROLL
    for (int pass = 0; pass < 2; ++pass) {
        const i64 neg = v[8] >> 63;                   // -1 when negative: add p
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) v[i] += neg & (i64)P.v[i];
        SBV_UNROLL
        for (int i = 0; i < 8; ++i) { v[i + 1] += v[i] >> 29; v[i] &= (i64)SBV_M29; }
    }
    SBV_NOUNROLL
    for (int pass = 0; pass < 2; ++pass) {
        i64 t[9];
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) t[i] = v[i] - P.v[i];
        SBV_UNROLL
        for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= (i64)SBV_M29; }
        const bool ge = t[8] >= 0;                      // value >= p
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) v[i] = ge ? t[i] : v[i];
    }
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = (i32)v[i];
}
// value == 0 (mod p) for |value| < 16 p.  A multiple k*p with |k| <= 16 has low limb -k mod 2^29 (limb 0 receives
// no carry, so it IS the value mod 2^29): everything else is rejected by three instructions.
SBV_HD bool f29_maybe_zero(const fe29& a) { return (((u32)a.v[0] + 16u) & SBV_M29) <= 32u; }
SBV_HD bool f29_is_zero_slow(const fe29& a) {
    fe29 c;
    f29_canon(c, a);
    i32 o = 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) o |= c.v[i];
    return o == 0;
}
SBV_HD bool f29_is_zero(const fe29& a) { return f29_maybe_zero(a) && f29_is_zero_slow(a); }
SBV_HD bool f29_limbs_all_zero(const fe29& a) {
    i32 o = 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) o |= a.v[i];
    return o == 0;
}

// ---- packing: 256-bit canonical integers (8 x 32-bit words, as stored in tables and scratch) <-> 9 x 29 -----------
SBV_HD void f29_unpack(fe29& r, const u32 w[8]) {
    r.v[0] = (i32)(w[0] & SBV_M29);
    SBV_UNROLL
    for (int i = 1; i < 8; ++i) {
        const int bit = 29 * i, lo = bit >> 5, sh = bit & 31;         // limb i = bits [29 i, 29 i + 29)
        const u32 x = sh == 0 ? w[lo] : ((w[lo] >> sh) | (lo + 1 < 8 && sh > 3 ? (w[lo + 1] << (32 - sh)) : 0u));
        r.v[i] = (i32)(x & SBV_M29);
    }
    r.v[8] = (i32)(w[7] >> 8);                                          // bits 232..255
}
// canonical limbs (f29_canon output) -> 8 words
SBV_HD void f29_pack(u32 w[8], const fe29& c) {
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) {
        const int bit = 32 * j, i = bit / 29, sh = bit - 29 * i;     // word j starts inside limb i at bit sh
        u32 x = (u32)c.v[i] >> sh;
        x |= (u32)c.v[i + 1] << (29 - sh);
        if (58 - sh < 32 && i + 2 < 9) x |= (u32)c.v[i + 2] << (58 - sh);
        w[j] = x;
    }
}

// ---- domain changes -------------------------------------------------------------------------------------------------
// plain integer x < 2^256 (8 words) -> x * R mod p
SBV_HD void f29_from_plain(fe29& r, const u256& x) {
    fe29 t;
    f29_unpack(t, x.v);
    f29_mul(r, t, f29_r2());
}
// 8 x 32 Montgomery form X = x * 2^256 mod p (p256_fe.h) -> x * 2^261
SBV_HD void f29_from_fe(fe29& r, const fe& X) {
    fe29 t;
    f29_unpack(t, X.v);
    f29_mul(r, t, f29_c266());
}
// x * 2^261 -> canonical 8-word x * 2^256 mod p (8 x 32 Montgomery form)
SBV_HD void f29_to_fe(fe& X, const fe29& a) {
    fe29 t, c;
    f29_mul(t, a, f29_c256());
    f29_canon(c, t);
    f29_pack(X.v, c);
}
// a^-1 (Montgomery domain in and out) by division steps on the canonical 256-bit integer (modinv30.h): the plain
// inverse of a R is a^-1 R^-1, one multiplication by R^3 brings it back to a^-1 R.  a = 0 (mod p) -> 0.
SBV_HD void f29_inv(fe29& r, const fe29& a) {
    fe29 c, t;
    f29_canon(c, a);
    u256 x, y;
    f29_pack(x.v, c);
    modinv30(y, x, modinfo30_p256());
    f29_unpack(t, y.v);
    f29_mul(r, t, f29_r3());
}

// the same through the constant-time division steps (the signing kernel: Z of k G depends on the nonce)
SBV_HD void f29_inv_ct(fe29& r, const fe29& a) {
    fe29 c, t;
    f29_canon(c, a);
    u256 x, y;
    f29_pack(x.v, c);
    modinv30_ct(y, x, modinfo30_p256());
    f29_unpack(t, y.v);
    f29_mul(r, t, f29_r3());
}

// x * 2^261 -> canonical 8-word value of the SAME domain (table / scratch storage: unpack gives it back)
SBV_HD void f29_store_canon(u32 w[8], const fe29& a) {
    fe29 c;
    f29_canon(c, a);
    f29_pack(w, c);
}

}  // namespace sbv
