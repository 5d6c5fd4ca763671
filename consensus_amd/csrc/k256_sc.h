// k256_sc.h — integers mod n, the order of the secp256k1 base point, for stage A of the secp256k1 variant:
// w = s^-1, u1 = e * w, u2 = r * w (SEC 1 v2.0 §4.1.4 steps 4-5).  n = 2^256 - c with c < 2^129, so a 512-bit product is
// reduced by folding the high half three times (hi * c + lo) and one or two conditional subtractions.  Plain 8 x 32-bit
// words: stage A is two of these products and one inversion per signature, a few per cent of the step.
//
// Shared host/device source.
#pragma once
#include "modinv30.h"
#include "sbv_common.h"

namespace sbv {

SBV_HD u256 k256_n_words() { u256 r = {{0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}}; return r; }
SBV_HD modinfo30 modinfo30_k256_n() {
    modinfo30 r = {{{0x10364141, 0x3F497A33, 0x348A03BB, 0x2BB739AB, 0x3FFFFEBA, 0x3FFFFFFF, 0x3FFFFFFF, 0x3FFFFFFF, 0xFFFF}}, 0x2A774EC1u};
    return r;
}

// out[0..no) += a[0..na) * b[0..nb)  (no >= na + nb; the caller guarantees that the sum fits)
template <int NO, int NA, int NB>
SBV_HD void ksc_muladd(u32 (&out)[NO], const u32* a, const u32* b) {
    SBV_UNROLL
    for (int i = 0; i < NA; ++i) {
        u64 carry = 0;
        SBV_UNROLL
        for (int j = 0; j < NB; ++j) {
            const u64 t = (u64)a[i] * b[j] + out[i + j] + carry;
            out[i + j] = (u32)t;
            carry = t >> 32;
        }
        SBV_UNROLL
        for (int k = i + NB; k < NO; ++k) {
            const u64 t = (u64)out[k] + carry;
            out[k] = (u32)t;
            carry = t >> 32;
        }
    }
}

// x (16 words) mod n
SBV_HD void ksc_reduce512(u256& r, const u32 x[16]) {
    const u32 c[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 0x00000001u};      // 2^256 - n
    // fold 1: t = lo + hi * c < 2^256 + 2^385
    u32 t[14];
    SBV_UNROLL
    for (int k = 0; k < 14; ++k) t[k] = k < 8 ? x[k] : 0u;
    ksc_muladd<14, 8, 5>(t, x + 8, c);
    // fold 2: u = t[0..8) + t[8..14) * c < 2^256 + 2^(130 + 129)
    u32 u[12];
    SBV_UNROLL
    for (int k = 0; k < 12; ++k) u[k] = k < 8 ? t[k] : 0u;
    ksc_muladd<12, 6, 5>(u, t + 8, c);
    // fold 3: v = u[0..8) + u[8..12) * c; u[8..12) < 2^4, so v < 2^256 + 2^134
    u32 v[10];
    SBV_UNROLL
    for (int k = 0; k < 10; ++k) v[k] = k < 8 ? u[k] : 0u;
    ksc_muladd<10, 4, 5>(v, u + 8, c);
    // fold 4: at most one multiple of 2^256 is left
    u32 w[9];
    SBV_UNROLL
    for (int k = 0; k < 9; ++k) w[k] = k < 8 ? v[k] : 0u;
    const u32 top[1] = {v[8]};
    ksc_muladd<9, 1, 5>(w, top, c);
    // w < 2^256 + small and w[8] is 0 unless v wrapped again (it cannot: v[0..8) + c < 2^256 when v[8] = 1, since then v[0..8) < 2^134)
    u256 y;
    SBV_UNROLL
    for (int k = 0; k < 8; ++k) y.v[k] = w[k];
    const u256 n = k256_n_words();
    SBV_UNROLL
    for (int rep = 0; rep < 2; ++rep) {
        u256 d;
        const u32 bw = sub256(d, y, n);
        select256(y, bw == 0, d, y);
    }
    r = y;
}

SBV_HD void ksc_mul(u256& r, const u256& a, const u256& b) {
    u32 x[16];
    SBV_UNROLL
    for (int k = 0; k < 16; ++k) x[k] = 0;
    ksc_muladd<16, 8, 8>(x, a.v, b.v);
    ksc_reduce512(r, x);
}

// x in [0, 2^256) -> x mod n (n > 2^255: one conditional subtraction)
SBV_HD void ksc_cond_sub_n(u256& r, const u256& x) {
    u256 d;
    const u32 bw = sub256(d, x, k256_n_words());
    select256(r, bw == 0, d, x);
}

SBV_HD void ksc_inv(u256& r, const u256& a) { modinv30(r, a, modinfo30_k256_n()); }

// ---- the GLV decomposition (round 6) ----------------------------------------------------------------------------
// secp256k1 has the endomorphism phi(x, y) = (beta x, y) = lambda (x, y) with beta^3 = 1 mod p, lambda^3 = 1 mod n (Gallant, Lambert,
// Vanstone 2001), so k P = k1 P + k2 phi(P) with k = k1 + k2 lambda mod n and |k1|, |k2| < 2^128: half the doublings of a variable-base
// multiplication.  The lattice basis (a1, b1), (a2, b2) with a_i + b_i lambda = 0 mod n comes from the extended Euclidean algorithm on
// (n, lambda); c1 = round(b2 k / n), c2 = round(-b1 k / n) are taken from the precomputed g1 = round(2^384 b2 / n),
// g2 = round(2^384 (-b1) / n) as (k g_i + 2^383) >> 384; then k2 = c1 (-b1) + c2 (-b2), k1 = k - k2 lambda (mod n).  Constants and the
// 128-bit bound re-derived and checked numerically (tests/test_k256_cpu.py: 200 000 random scalars and the edge scalars); they are the
// ones every secp256k1 implementation uses.  A residue above n / 2 is returned as its negative (the caller flips the point's sign).
SBV_HD u256 k256_lambda_words() { u256 r = {{0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu, 0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu}}; return r; }
SBV_HD u256 k256_beta_words() { u256 r = {{0x719501EEu, 0xC1396C28u, 0x12F58995u, 0x9CF04975u, 0xAC3434E9u, 0x6E64479Eu, 0x657C0710u, 0x7AE96A2Bu}}; return r; }
// (k * g + 2^383) >> 384 for g < 2^256: a value below 2^128
SBV_HD void ksc_mul_shift384(u256& r, const u256& k, const u256& g) {
    u32 x[16];
    SBV_UNROLL
    for (int i = 0; i < 16; ++i) x[i] = 0;
    ksc_muladd<16, 8, 8>(x, k.v, g.v);
    u32 carry = 0;
    x[11] = addc(x[11], 0x80000000u, carry);       // + 2^383: rounding
    SBV_UNROLL
    for (int i = 12; i < 16; ++i) x[i] = addc(x[i], 0u, carry);
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) r.v[i] = i < 4 ? x[12 + i] : 0u;
}
// k (any 256-bit value; reduced mod n first) -> k = (neg1 ? -k1 : k1) + (neg2 ? -k2 : k2) * lambda (mod n), k1, k2 < 2^128 + a few
SBV_HD void ksc_split_lambda(u256& k1, bool& neg1, u256& k2, bool& neg2, const u256& kin) {
    const u256 g1 = {{0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u}};
    const u256 g2 = {{0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u}};
    const u256 minus_b1 = {{0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u, 0u, 0u, 0u, 0u}};
    const u256 minus_b2 = {{0x3DB1562Cu, 0xD765CDA8u, 0x0774346Du, 0x8A280AC5u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}};
    const u256 n = k256_n_words();
    u256 k;
    ksc_cond_sub_n(k, kin);
    u256 c1, c2, t1, t2, r2, r1, d;
    ksc_mul_shift384(c1, k, g1);
    ksc_mul_shift384(c2, k, g2);
    ksc_mul(t1, c1, minus_b1);
    ksc_mul(t2, c2, minus_b2);
    u32 cy = add256(r2, t1, t2);                    // t1, t2 < n: one conditional subtraction of n (also when the sum wrapped 2^256)
    u32 bw = sub256(d, r2, n);
    select256(r2, cy != 0 || bw == 0, d, r2);
    ksc_mul(t1, r2, k256_lambda_words());
    bw = sub256(r1, k, t1);                         // k - r2 lambda mod n
    (void)add256(d, r1, n);
    select256(r1, bw != 0, d, r1);
    // sign: a residue above n / 2 becomes its negative
    (void)sub256(d, n, r1);
    neg1 = lt256(d, r1);
    select256(k1, neg1, d, r1);
    (void)sub256(d, n, r2);
    neg2 = lt256(d, r2);
    select256(k2, neg2, d, r2);
}

}  // namespace sbv
