// p256_sc29.h — arithmetic modulo the group order N of P-256 on the carry-free representation of p256_fe29.h
// (9 signed 29-bit limbs, Montgomery domain R = 2^261): s^-1, u1 = e * s^-1, u2 = r * s^-1 of
// crypto/ecdsa.verifyNISTEC (Go does this with crypto/internal/bigmod), for stage A (p256_core.h: prep_chunk29).
//
// Why: stage A is one inversion per thread plus six multiplications per signature.  On the 8 x 32 form (p256_sc.h)
// every multiplication is a CIOS loop of dependent v_addc chains; measured, the kernel issued one VALU instruction
// per 8 cycles per SIMD (0.35 - 0.42 ms per 2^20 batch during which stage B cannot start).  Here a product is 81
// independent multiply-accumulates into 64-bit columns; N has no special form, so each of the nine reduction steps
// costs a v_mul_lo_u32 for the multiplier m = column * (-N^-1) mod 2^29 and nine multiply-accumulates m * N[i].
#pragma once
#include "p256_fe29.h"
#include "p256_sc.h"

namespace sbv {

SBV_HD fe29 s29_n() { fe29 r = {{0x1C632551, 0x1DCE5617, 0x05E7A13C, 0x0DF55B4E, 0x1FFFFBCE, 0x1FFFFFFF, 0x0003FFFF, 0x1FE00000, 0x00FFFFFF}}; return r; }
SBV_HD fe29 s29_one() { fe29 r = {{0x139B55E0, 0x06353D03, 0x030BD862, 0x0154963A, 0x00008632, 0x00000000, 0x1F800000, 0x03FFFFFF, 0x00000000}}; return r; }
SBV_HD fe29 s29_r2() { fe29 r = {{0x148D9EF5, 0x0F4E7F75, 0x14C6A651, 0x03F8B765, 0x165861F1, 0x1256D7D8, 0x1C185B23, 0x0D4CAB0F, 0x0084B655}}; return r; }
SBV_HD fe29 s29_r3() { fe29 r = {{0x1E021DD3, 0x1D50B165, 0x1CC6C8AE, 0x02FC7618, 0x19313170, 0x1AE97A4F, 0x0A137248, 0x041BE64D, 0x002A73DA}}; return r; }
#define SBV_N29_PRIME 0x0E00BC4Fu          // -N^-1 mod 2^29

// r = a * b / R mod N; value in (A*B/R, A*B/R + N), limbs 0..7 exact 29-bit limbs.  Operands: as f29_mul.
SBV_HD void s29_mul(fe29& r, const fe29& a, const fe29& b) {
    SBV_F29_CHECK_OPERANDS(a, b, false);
    const fe29 N = s29_n();
    const f29_consts K = f29_load_consts();
    i64 c[18];
    SBV_UNROLL
    for (int k = 0; k < 18; ++k) c[k] = 0;
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) {
        SBV_UNROLL
        for (int j = 0; j < 9; ++j) c[i + j] = f29_mad(a.v[i], b.v[j], c[i + j]);
    }
    SBV_UNROLL
    for (int k = 0; k < 9; ++k) {
        const u32 m = ((u32)c[k] * SBV_N29_PRIME) & SBV_M29;
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) c[k + i] = f29_mad((i32)m, N.v[i], c[k + i]);
        const u32 lo = (u32)c[k];                       // low 29 bits are zero now
        const i32 hi = (i32)(c[k] >> 32);
        c[k + 1] = f29_mad(hi, K.k8, c[k + 1]);
        c[k + 1] = f29_madu(lo >> 29, K.k1, c[k + 1]);
    }
    SBV_UNROLL
    for (int j = 9; j < 17; ++j) {
        const u32 lo = (u32)c[j];
        const i32 hi = (i32)(c[j] >> 32);
        r.v[j - 9] = (i32)(lo & SBV_M29);
        c[j + 1] = f29_mad(hi, K.k8, c[j + 1]);
        c[j + 1] = f29_madu(lo >> 29, K.k1, c[j + 1]);
    }
    r.v[8] = (i32)c[17];
}

// any value in (-2N, 3N) with limbs |v[i]| < 2^31  ->  the representative in [0, N), exact limbs
SBV_HD void s29_canon(fe29& r, const fe29& a) {
    const fe29 N = s29_n();
    i64 v[9];
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) v[i] = a.v[i];
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { v[i + 1] += v[i] >> 29; v[i] &= (i64)SBV_M29; }
    SBV_NOUNROLL
    for (int pass = 0; pass < 2; ++pass) {
        const i64 neg = v[8] >> 63;
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) v[i] += neg & (i64)N.v[i];
        SBV_UNROLL
        for (int i = 0; i < 8; ++i) { v[i + 1] += v[i] >> 29; v[i] &= (i64)SBV_M29; }
    }
    SBV_NOUNROLL
    for (int pass = 0; pass < 2; ++pass) {
        i64 t[9];
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) t[i] = v[i] - N.v[i];
        SBV_UNROLL
        for (int i = 0; i < 8; ++i) { t[i + 1] += t[i] >> 29; t[i] &= (i64)SBV_M29; }
        const bool ge = t[8] >= 0;
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) v[i] = ge ? t[i] : v[i];
    }
    SBV_UNROLL
    for (int i = 0; i < 9; ++i) r.v[i] = (i32)v[i];
}
// canonical 256-bit words of a residue (the value itself, whatever domain it is in)
SBV_HD void s29_store_canon(u256& w, const fe29& a) {
    fe29 c;
    s29_canon(c, a);
    f29_pack(w.v, c);
}
// a^-1 (Montgomery domain in and out) by division steps; 0 -> 0
SBV_HD void s29_inv(fe29& r, const fe29& a) {
    u256 x, y;
    s29_store_canon(x, a);
    modinv30(y, x, modinfo30_p256_order());
    fe29 t;
    f29_unpack(t, y.v);
    s29_mul(r, t, s29_r3());
}

// constant-time form (the signing kernel inverts the secret nonce)
SBV_HD void s29_inv_ct(fe29& r, const fe29& a) {
    u256 x, y;
    s29_store_canon(x, a);
    modinv30_ct(y, x, modinfo30_p256_order());
    fe29 t;
    f29_unpack(t, y.v);
    s29_mul(r, t, s29_r3());
}

}  // namespace sbv
