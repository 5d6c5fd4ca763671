// p256_sc.h — arithmetic modulo the group order N of P-256 (the "s^-1, u1 = e*w, u2 = r*w"
// part of crypto/ecdsa.verifyNISTEC; Go does this with crypto/internal/bigmod).
//
// 8 x 32-bit limbs, generic CIOS Montgomery multiplication with -N^-1 mod 2^32 = 0xEE00BC4F.
// This is <2% of the per-signature work once s^-1 is amortised by Montgomery's trick over a
// thread's chunk of tuples (p256_core.h: prep_chunk), so it is written for clarity.
#pragma once
#include "modinv30.h"
#include "sbv_common.h"

namespace sbv {

typedef u256 sc;

SBV_HD sc sc_n() { sc r = {{0xFC632551u, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu}}; return r; }
SBV_HD sc sc_one_mont() { sc r = {{0x039CDAAFu, 0x0C46353Du, 0x58E8617Bu, 0x43190552u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0x00000000u}}; return r; }
SBV_HD sc sc_r2() { sc r = {{0xBE79EEA2u, 0x83244C95u, 0x49BD6FA6u, 0x4699799Cu, 0x2B6BEC59u, 0x2845B239u, 0xF3D95620u, 0x66E12D94u}}; return r; }
#define SBV_N0_INV 0xEE00BC4Fu

// r = (carry:t >= N) ? t - N : t
SBV_HD void sc_cond_sub_n(sc& r, const sc& t, u32 carry) {
    const sc n = sc_n();
    sc d;
    u32 bw = sub256(d, t, n);
    bool use_d = (carry != 0) | (bw == 0);
    select256(r, use_d, d, t);
}

// Montgomery product a*b*2^-256 mod N; a, b < N
SBV_HD void sc_mul(sc& r, const sc& a, const sc& b) {
    const sc n = sc_n();
    u32 t[10];
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) t[i] = 0;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) {
        u32 c = 0;
        SBV_UNROLL
        for (int j = 0; j < 8; ++j) {
            u64 acc = (u64)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (u32)acc;
            c = (u32)(acc >> 32);
        }
        u64 s = (u64)t[8] + c;
        t[8] = (u32)s;
        t[9] = (u32)(s >> 32);
        const u32 m = t[0] * SBV_N0_INV;
        u64 acc = (u64)m * n.v[0] + t[0];
        c = (u32)(acc >> 32);
        SBV_UNROLL
        for (int j = 1; j < 8; ++j) {
            acc = (u64)m * n.v[j] + t[j] + c;
            t[j - 1] = (u32)acc;
            c = (u32)(acc >> 32);
        }
        s = (u64)t[8] + c;
        t[7] = (u32)s;
        t[8] = t[9] + (u32)(s >> 32);
    }
    sc tt;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) tt.v[i] = t[i];
    sc_cond_sub_n(r, tt, t[8]);
}
SBV_HD void sc_sqr(sc& r, const sc& a) { sc_mul(r, a, a); }
SBV_HD void sc_to_mont(sc& r, const u256& a) { sc r2 = sc_r2(); sc_mul(r, a, r2); }
SBV_HD void sc_from_mont(u256& r, const sc& a) {
    sc one = {{1, 0, 0, 0, 0, 0, 0, 0}};
    sc_mul(r, a, one);
}

// a^(N-2) mod N in the Montgomery domain (Fermat).  N-2 = FFFFFFFF 00000000 FFFFFFFF FFFFFFFF ||
// BCE6FAAD A7179E84 F3B9CAC2 FC63254F: the top half uses the a^(2^32-1) ladder, the bottom
// 128 bits plain square-and-multiply (the exponent is a constant, so the branch is uniform
// across the wavefront).  256 squarings + ~95 multiplications, no table (no scratch memory).
SBV_HD void sc_inv(sc& r, const sc& a) {
    sc x2, x4, x8, x16, x32, t;
    sc_sqr(t, a); sc_mul(x2, t, a);
    t = x2; for (int i = 0; i < 2; ++i) sc_sqr(t, t); sc_mul(x4, t, x2);
    t = x4; for (int i = 0; i < 4; ++i) sc_sqr(t, t); sc_mul(x8, t, x4);
    t = x8; for (int i = 0; i < 8; ++i) sc_sqr(t, t); sc_mul(x16, t, x8);
    t = x16; for (int i = 0; i < 16; ++i) sc_sqr(t, t); sc_mul(x32, t, x16);
    t = x32;
    for (int i = 0; i < 64; ++i) sc_sqr(t, t);
    sc_mul(t, t, x32);
    for (int i = 0; i < 32; ++i) sc_sqr(t, t);
    sc_mul(t, t, x32);
    const u32 low[4] = {0xFC63254Fu, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu};
    for (int i = 127; i >= 0; --i) {
        sc_sqr(t, t);
        if ((low[i >> 5] >> (i & 31)) & 1u) sc_mul(t, t, a);
    }
    r = t;
}

// The same inverse by division steps (modinv30.h; see fe_inv_gcd): Montgomery form in and out, R^3 mod N.
SBV_HD void sc_inv_gcd(sc& r, const sc& a) {
    const sc r3 = {{0x0B65A624u, 0xAC8EBEC9u, 0x0C0555C9u, 0x111F28AEu, 0x6BA5E93Fu, 0x2543B924u, 0x6407BE65u, 0x503A54E7u}};
    u256 t;
    modinv30(t, a, modinfo30_p256_order());
    sc_mul(r, t, r3);
}

}  // namespace sbv
