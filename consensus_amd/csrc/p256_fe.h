// p256_fe.h — arithmetic in GF(p), p = 2^256 - 2^224 + 2^192 + 2^96 - 1 (NIST P-256).
//
// Replaces (for the device path) what Go's crypto/internal/nistec/p256_asm_amd64.s does on
// the CPU behind crypto/ecdsa.VerifyASN1 — the arithmetic the reference's api.Verifier
// implementations are expected to run (pkg/api/dependencies.go:54-71).
//
// Representation: 8 x 32-bit little-endian limbs, Montgomery domain (x*R mod p, R = 2^256),
// always fully reduced to [0, p).  One field element = 8 VGPRs per lane on gfx950.
// The 32x32->64 multiply-accumulate maps to v_mad_u64_u32; reduction uses no multiplies at
// all because -p^-1 = 1 (mod 2^32) and p is a sum of four signed powers of 2^32.
#pragma once
#include "sbv_common.h"
#include "modinv30.h"

namespace sbv {

typedef u256 fe;

#define SBV_P_LIMBS {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xFFFFFFFFu}

SBV_HD fe fe_p() { fe r = {SBV_P_LIMBS}; return r; }
// R mod p (Montgomery 1)
SBV_HD fe fe_one() { fe r = {{0x00000001u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu, 0x00000000u}}; return r; }
// R^2 mod p
SBV_HD fe fe_r2() { fe r = {{0x00000003u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFBu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFDu, 0x00000004u}}; return r; }
// curve b in Montgomery form (b*R mod p)
SBV_HD fe fe_b_mont() { fe r = {{0x29C4BDDFu, 0xD89CDF62u, 0x78843090u, 0xACF005CDu, 0xF7212ED6u, 0xE5A220ABu, 0x04874834u, 0xDC30061Du}}; return r; }

SBV_HD fe fe_zero() { fe r = {{0, 0, 0, 0, 0, 0, 0, 0}}; return r; }
SBV_HD bool fe_is_zero(const fe& a) { return is_zero256(a); }
SBV_HD bool fe_eq(const fe& a, const fe& b) { return eq256(a, b); }

// r = (t >= p || carry) ? t - p : t      (t < 2p as a 257-bit value carry:t)
SBV_HD void fe_cond_sub_p(fe& r, const fe& t, u32 carry) {
    // t - p = t + (2^256 - p) - 2^256;  2^256 - p = 2^224 - 2^192 - 2^96 + 1
    fe d;
    u32 bw = 0;
    d.v[0] = subb(t.v[0], 0xFFFFFFFFu, bw);
    d.v[1] = subb(t.v[1], 0xFFFFFFFFu, bw);
    d.v[2] = subb(t.v[2], 0xFFFFFFFFu, bw);
    d.v[3] = subb(t.v[3], 0u, bw);
    d.v[4] = subb(t.v[4], 0u, bw);
    d.v[5] = subb(t.v[5], 0u, bw);
    d.v[6] = subb(t.v[6], 1u, bw);
    d.v[7] = subb(t.v[7], 0xFFFFFFFFu, bw);
    // take d when the subtraction did not borrow out of the 257-bit value
    bool use_d = (carry != 0) | (bw == 0);
    select256(r, use_d, d, t);
}

// FAST mode (stage B's first pass): the conditional subtraction only handles the carry == 1 half
// (t - p = t + (2^256 - p) mod 2^256, one masked add chain, 8 instructions fewer) and the other
// case — carry == 0 but t >= p, which needs t's top limb to be 0xFFFFFFFF, i.e. probability
// 2^-32 per operation on random data — is only DETECTED: *st keeps the maximum top limb seen.
// A lane whose sticky word reached 0xFFFFFFFF re-runs the exact (FAST = false) code, so verdicts
// stay exact for every input while the common path loses the compare-and-select.
template <bool FAST>
SBV_HD void fe_cond_sub_p_t(fe& r, const fe& t, u32 carry, u32* st) {
    if (!FAST) { fe_cond_sub_p(r, t, carry); return; }
    *st = *st > t.v[7] ? *st : t.v[7];
    const u32 m = 0u - carry;          // 2^256 - p = {1, 0, 0, ~0, ~0, ~0, ~0 - 1, 0}
    u32 c = carry;                     // the +1 of limb 0 enters as the chain's carry-in
    r.v[0] = addc(t.v[0], 0u, c);
    r.v[1] = addc(t.v[1], 0u, c);
    r.v[2] = addc(t.v[2], 0u, c);
    r.v[3] = addc(t.v[3], m, c);
    r.v[4] = addc(t.v[4], m, c);
    r.v[5] = addc(t.v[5], m, c);
    r.v[6] = addc(t.v[6], m & 0xFFFFFFFEu, c);
    r.v[7] = addc(t.v[7], 0u, c);
}

template <bool FAST = false>
SBV_HD void fe_add(fe& r, const fe& a, const fe& b, u32* st = nullptr) {
    fe t;
    u32 c = add256(t, a, b);
    fe_cond_sub_p_t<FAST>(r, t, c, st);
}
template <bool FAST = false>
SBV_HD void fe_dbl(fe& r, const fe& a, u32* st = nullptr) { fe_add<FAST>(r, a, a, st); }

SBV_HD void fe_sub(fe& r, const fe& a, const fe& b) {
    fe d;
    u32 bw = sub256(d, a, b);
    // add p back when we borrowed
    u32 m = 0u - bw;  // 0 or 0xFFFFFFFF
    u32 c = 0;
    r.v[0] = addc(d.v[0], m, c);
    r.v[1] = addc(d.v[1], m, c);
    r.v[2] = addc(d.v[2], m, c);
    r.v[3] = addc(d.v[3], 0u, c);
    r.v[4] = addc(d.v[4], 0u, c);
    r.v[5] = addc(d.v[5], 0u, c);
    r.v[6] = addc(d.v[6], m & 1u, c);
    r.v[7] = addc(d.v[7], m, c);
}
SBV_HD void fe_neg(fe& r, const fe& a) {
    fe z = fe_zero();
    fe_sub(r, z, a);
}
// r = neg ? -a : a
SBV_HD void fe_cneg(fe& r, const fe& a, bool neg) {
    fe n;
    fe_neg(n, a);
    select256(r, neg, n, a);
}

// ---- 256 x 256 -> 512 products -----------------------------------------------------------------
// Row-wise: row_i = a * b[i] is a chain of v_mad_u64_u32 whose 64-bit addend carries the
// previous product's high word (a*b + hi <= 2^64 - 2^32, never overflows, no VCC); the rows
// are then summed with v_add_co / v_addc_co chains.  On gfx950 a VALU that writes VCC must be
// 2 wait states ahead of the VALU that consumes it, and hipcc fills those slots with the next
// row's multiplies — 64 mad + ~57 mov + ~63 addc per product, no inline asm.
SBV_HD void mul_wide(u32 t[16], const u32 a[8], const u32 b[8]) {
    u64 q = 0;
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) {
        q = (u64)a[j] * b[0] + (q >> 32);
        t[j] = (u32)q;
    }
    t[8] = (u32)(q >> 32);
    SBV_UNROLL
    for (int i = 1; i < 8; ++i) {
        u32 row[9];
        q = 0;
        SBV_UNROLL
        for (int j = 0; j < 8; ++j) {
            q = (u64)a[j] * b[i] + (q >> 32);
            row[j] = (u32)q;
        }
        row[8] = (u32)(q >> 32);
        u32 c = 0;
        SBV_UNROLL
        for (int j = 0; j < 8; ++j) t[i + j] = addc(t[i + j], row[j], c);
        t[i + 8] = row[8] + c;       // cannot overflow: the partial sum is < 2^(32(i+9))
    }
}

SBV_HD void sqr_wide(u32 t[16], const u32 a[8]) {
    // x = sum_{i<j} a[i] a[j] 2^(32(i+j)), built row-wise; t = 2x + sum a[i]^2 2^(64i)
    u32 x[16];
    x[0] = 0;
    u64 q = 0;
    SBV_UNROLL
    for (int j = 1; j < 8; ++j) {
        q = (u64)a[j] * a[0] + (q >> 32);
        x[j] = (u32)q;
    }
    x[8] = (u32)(q >> 32);
    SBV_UNROLL
    for (int i = 1; i < 7; ++i) {
        u32 row[8];
        q = 0;
        SBV_UNROLL
        for (int j = i + 1; j < 8; ++j) {
            q = (u64)a[j] * a[i] + (q >> 32);
            row[j - i - 1] = (u32)q;
        }
        const int len = 7 - i;              // products in this row
        row[len] = (u32)(q >> 32);
        u32 c = 0;
        SBV_UNROLL
        for (int j = 0; j < len; ++j) x[2 * i + 1 + j] = addc(x[2 * i + 1 + j], row[j], c);
        x[2 * i + 1 + len] = row[len] + c;  // fresh limb: index i + 8
    }
    x[15] = 0;
    u32 c = 0;      // carry of the running (2x + diag) sum
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) {
        const u64 d = (u64)a[i] * a[i];
        const u32 x0 = (x[2 * i] << 1) | (i ? (x[2 * i - 1] >> 31) : 0u);
        const u32 x1 = (x[2 * i + 1] << 1) | (x[2 * i] >> 31);
        t[2 * i] = addc(x0, (u32)d, c);
        t[2 * i + 1] = addc(x1, (u32)(d >> 32), c);
    }
}

// ---- Montgomery reduction for p: r = T / 2^256 mod p, T < p * 2^256 ------------------------------
// M = T_lo * (-p^-1) mod 2^256 with -p^-1 = 1 + 2^96 + 2^193 - 2^224 (mod 2^256), then
// r = (T + M*p) / 2^256 = T_hi + M + (M >> 64) + (M >> 160) - (M >> 32) + k, where k is the
// (small, signed) carry of the vanishing low half.  Shifts and adds only.
template <bool FAST = false>
SBV_HD void fe_mont_reduce(fe& r, const u32 t[16], u32* st = nullptr) {
    u32 m[8];
    {
        // M = T_lo + (T_lo << 96) + (T_lo << 193) - (T_lo << 224)   (mod 2^256)
        u32 c = 0;
        m[0] = t[0]; m[1] = t[1]; m[2] = t[2];
        m[3] = addc(t[3], t[0], c);
        m[4] = addc(t[4], t[1], c);
        m[5] = addc(t[5], t[2], c);
        m[6] = addc(t[6], t[3], c);
        m[7] = addc(t[7], t[4], c);
        c = 0;
        m[6] = addc(m[6], t[0] << 1, c);
        m[7] = addc(m[7], (t[1] << 1) | (t[0] >> 31), c);
        m[7] -= t[0];
    }
    // k = round((t7 + M4 + M1 - M7 - M0) / 2^32): carry from the low half, in {-1..2}
    int64_t V = (int64_t)((u64)t[7] + m[4] + m[1]) - (int64_t)((u64)m[7] + m[0]);
    int32_t k = (int32_t)((V + (int64_t)0x80000000ll) >> 32);
    const u32 kp1 = (u32)(k + 1);   // 0..3, fed in as the carry-ins of the three add chains;
                                    // the matching -1 is the borrow-in of the subtract chain
    // acc (9 limbs) = T_hi + M + [kp1 >= 1]
    u32 acc[9];
    u32 c = kp1 >= 1u ? 1u : 0u;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) acc[i] = addc(t[8 + i], m[i], c);
    acc[8] = c;
    // acc += (M >> 64) + [kp1 >= 2]
    c = kp1 >= 2u ? 1u : 0u;
    SBV_UNROLL
    for (int i = 0; i < 6; ++i) acc[i] = addc(acc[i], m[i + 2], c);
    acc[6] = addc(acc[6], 0u, c);
    acc[7] = addc(acc[7], 0u, c);
    acc[8] += c;
    // acc += (M >> 160) + [kp1 >= 3]
    c = kp1 >= 3u ? 1u : 0u;
    acc[0] = addc(acc[0], m[5], c);
    acc[1] = addc(acc[1], m[6], c);
    acc[2] = addc(acc[2], m[7], c);
    SBV_UNROLL
    for (int i = 3; i < 8; ++i) acc[i] = addc(acc[i], 0u, c);
    acc[8] += c;
    // acc -= (M >> 32) + 1
    u32 bw = 1;
    SBV_UNROLL
    for (int i = 0; i < 7; ++i) acc[i] = subb(acc[i], m[i + 1], bw);
    acc[7] = subb(acc[7], 0u, bw);
    acc[8] -= bw;
    fe tt;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) tt.v[i] = acc[i];
    fe_cond_sub_p_t<FAST>(r, tt, acc[8], st);
}

template <bool FAST = false>
SBV_HD void fe_mul(fe& r, const fe& a, const fe& b, u32* st = nullptr) {
    u32 t[16];
    mul_wide(t, a.v, b.v);
    fe_mont_reduce<FAST>(r, t, st);
}
template <bool FAST = false>
SBV_HD void fe_sqr(fe& r, const fe& a, u32* st = nullptr) {
    u32 t[16];
    sqr_wide(t, a.v);
    fe_mont_reduce<FAST>(r, t, st);
}

// plain integer (< p) -> Montgomery form, and back
template <bool FAST = false>
SBV_HD void fe_to_mont(fe& r, const u256& a, u32* st = nullptr) { fe r2 = fe_r2(); fe_mul<FAST>(r, a, r2, st); }
SBV_HD void fe_from_mont(u256& r, const fe& a) {
    u32 t[16];
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { t[i] = a.v[i]; t[8 + i] = 0; }
    fe_mont_reduce(r, t);
}

// a^(p-2): addition chain for p-2 = 2^256 - 2^224 + 2^192 + 2^96 - 3 (255 squarings, 12 multiplies).
// Host-side table generation only; the verify kernels never invert in GF(p).
SBV_HD void fe_inv(fe& r, const fe& a) {
    fe x2, x3, x6, x12, x15, x30, x32, t;
    fe_sqr(t, a); fe_mul(x2, t, a);                       // 2^2 - 1
    fe_sqr(t, x2); fe_mul(x3, t, a);                      // 2^3 - 1
    t = x3; for (int i = 0; i < 3; ++i) fe_sqr(t, t); fe_mul(x6, t, x3);
    t = x6; for (int i = 0; i < 6; ++i) fe_sqr(t, t); fe_mul(x12, t, x6);
    t = x12; for (int i = 0; i < 3; ++i) fe_sqr(t, t); fe_mul(x15, t, x3);
    t = x15; for (int i = 0; i < 15; ++i) fe_sqr(t, t); fe_mul(x30, t, x15);
    t = x30; for (int i = 0; i < 2; ++i) fe_sqr(t, t); fe_mul(x32, t, x2);
    // exponent bits, high to low: 32 ones | 31 zeros, 1 | 96 zeros | 94 ones, 0, 1
    t = x32;
    for (int i = 0; i < 32; ++i) fe_sqr(t, t);
    fe_mul(t, t, a);                                       // ... 0^31 1
    for (int i = 0; i < 96; ++i) fe_sqr(t, t);             // 96 zeros
    for (int i = 0; i < 32; ++i) fe_sqr(t, t); fe_mul(t, t, x32);
    for (int i = 0; i < 32; ++i) fe_sqr(t, t); fe_mul(t, t, x32);
    for (int i = 0; i < 30; ++i) fe_sqr(t, t); fe_mul(t, t, x30);
    fe_sqr(t, t); fe_sqr(t, t); fe_mul(r, t, a);           // 0 1
}

// The same inverse by division steps (modinv30.h): ~4x fewer instructions than the Fermat chain and no
// long multiply dependency chain.  a = xR (Montgomery form, < p): the integer inverse is x^-1 R^-1, one
// Montgomery multiplication by R^3 mod p brings it back to x^-1 R.  a = 0 -> 0, like fe_inv.
SBV_HD void fe_inv_gcd(fe& r, const fe& a) {
    const fe r3 = {{0x0000000Au, 0xFFFFFFFDu, 0xFFFFFFF7u, 0xFFFFFFEDu, 0xFFFFFFFCu, 0x00000005u, 0x00000001u, 0x00000018u}};
    u256 t;
    modinv30(t, a, modinfo30_p256());
    fe_mul(r, t, r3);
}

}  // namespace sbv
