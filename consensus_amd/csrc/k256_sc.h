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

}  // namespace sbv
