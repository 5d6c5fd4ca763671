// ed25519_core.h — per-lane Ed25519 verification (BASELINE.json configs[4]) with the semantics of
// Go crypto/ed25519.Verify (crypto/internal/edwards25519): see oracle/ed25519_oracle.c for the rules.
//
// ABI tuple, 128 bytes, all little-endian as on the wire:
//     R_enc (32) | S (32) | A_enc = public key (32) | k (32) = SHA-512(R_enc || A_enc || msg) mod L
// k is produced on the host (sbv_ed25519_hram_batch); a k >= L cannot come out of a reduction and
// makes the tuple invalid, like a non-canonical S.  Accept <=> encode([S]B + [k](-A)) == R_enc.
//
// Twisted Edwards a = -1 with d non-square: the unified addition is COMPLETE, so unlike P-256
// there are no exceptional cases to route.  Extended coordinates (X:Y:Z:T); [k](-A) by 64 signed
// 4-bit windows from a per-signature table of projective-Niels points in HBM, [S]B by 32 signed
// 8-bit comb windows of affine-Niels points (393 KiB, L2 resident).  One inversion per lane to
// re-encode R (the comparison is byte-wise by specification).
#pragma once
#include "ed25519_fe.h"
#include "p256_core.h"   // fe_store16 / fe_load16 / add_const_limbs / vec4

namespace sbv {

struct ept { fe25 X, Y, Z, T; };
struct pniels { fe25 YpX, YmX, Z, T2d; };   // 128 bytes
struct aniels { fe25 ypx, ymx, xy2d; };     // 96 bytes

#define SBV_ED_QTAB_ENTRIES 8
#define SBV_ED_BTAB_WINDOWS 32
#define SBV_ED_BTAB_PER_WINDOW 128

SBV_HD void ed_set_ident(ept& p) { p.X = fe25_zero(); p.Y = fe25_one(); p.Z = fe25_one(); p.T = fe25_zero(); }

// r = 2p   (4S + 4M)
SBV_HD void ed_dbl(ept& r, const ept& p) {
    fe25 xx, yy, zz2, xy2, e, g, h, f;
    fe25_sqr(xx, p.X);
    fe25_sqr(yy, p.Y);
    fe25_sqr(zz2, p.Z);
    fe25_add(zz2, zz2, zz2);
    fe25_add(xy2, p.X, p.Y);
    fe25_sqr(xy2, xy2);
    fe25_add(h, yy, xx);
    fe25_sub(g, yy, xx);
    fe25_sub(e, xy2, h);
    fe25_sub(f, zz2, g);
    fe25_mul(r.X, e, f);
    fe25_mul(r.Y, h, g);
    fe25_mul(r.Z, g, f);
    fe25_mul(r.T, e, h);
}

// R += (+-)q for a projective-Niels q; no-op when skip   (8M)
SBV_HD void ed_add_pniels(ept& R, const pniels& q, bool neg, bool skip) {
    fe25 a, b, c, d, e, f, g, h, ypx, ymx, t2d;
    select256(ypx, neg, q.YmX, q.YpX);
    select256(ymx, neg, q.YpX, q.YmX);
    fe25_cneg(t2d, q.T2d, neg);
    fe25_sub(a, R.Y, R.X);
    fe25_mul(a, a, ymx);
    fe25_add(b, R.Y, R.X);
    fe25_mul(b, b, ypx);
    fe25_mul(c, R.T, t2d);
    fe25_mul(d, R.Z, q.Z);
    fe25_add(d, d, d);
    fe25_sub(e, b, a);
    fe25_sub(f, d, c);
    fe25_add(g, d, c);
    fe25_add(h, b, a);
    ept n;
    fe25_mul(n.X, e, f);
    fe25_mul(n.Y, g, h);
    fe25_mul(n.Z, f, g);
    fe25_mul(n.T, e, h);
    select256(R.X, skip, R.X, n.X);
    select256(R.Y, skip, R.Y, n.Y);
    select256(R.Z, skip, R.Z, n.Z);
    select256(R.T, skip, R.T, n.T);
}
// R += (+-)q for an affine-Niels q (Z2 = 1); no-op when skip   (7M)
SBV_HD void ed_add_aniels(ept& R, const aniels& q, bool neg, bool skip) {
    fe25 a, b, c, d, e, f, g, h, ypx, ymx, t2d;
    select256(ypx, neg, q.ymx, q.ypx);
    select256(ymx, neg, q.ypx, q.ymx);
    fe25_cneg(t2d, q.xy2d, neg);
    fe25_sub(a, R.Y, R.X);
    fe25_mul(a, a, ymx);
    fe25_add(b, R.Y, R.X);
    fe25_mul(b, b, ypx);
    fe25_mul(c, R.T, t2d);
    fe25_add(d, R.Z, R.Z);
    fe25_sub(e, b, a);
    fe25_sub(f, d, c);
    fe25_add(g, d, c);
    fe25_add(h, b, a);
    ept n;
    fe25_mul(n.X, e, f);
    fe25_mul(n.Y, g, h);
    fe25_mul(n.Z, f, g);
    fe25_mul(n.T, e, h);
    select256(R.X, skip, R.X, n.X);
    select256(R.Y, skip, R.Y, n.Y);
    select256(R.Z, skip, R.Z, n.Z);
    select256(R.T, skip, R.T, n.T);
}
SBV_HD void ed_to_pniels(pniels& o, const ept& p) {
    const fe25 d2 = fe25_2d();
    fe25_add(o.YpX, p.Y, p.X);
    fe25_sub(o.YmX, p.Y, p.X);
    o.Z = p.Z;
    fe25_mul(o.T2d, p.T, d2);
}

// edwards25519.Point.SetBytes: `w` = the 8 little-endian dwords of the encoding.  false = not a point.
SBV_HD bool ed_decompress(ept& A, const u32 w[8]) {
    fe25 y, y2, u, v, v3, v7, t, rr, check, nu, nui;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) y.v[i] = w[i];
    const bool sign = (w[7] >> 31) != 0;
    y.v[7] &= 0x7FFFFFFFu;                       // non-canonical y (>= p) is accepted as y mod p
    const fe25 one = fe25_one(), dd = fe25_d(), sm1 = fe25_sqrtm1();
    fe25_sqr(y2, y);
    fe25_sub(u, y2, one);
    fe25_mul(v, y2, dd);
    fe25_add(v, v, one);
    fe25_sqr(t, v); fe25_mul(v3, t, v);
    fe25_sqr(t, v3); fe25_mul(v7, t, v);
    fe25_mul(t, u, v7);
    fe25_pow22523(t, t);
    fe25_mul(rr, u, v3);
    fe25_mul(rr, rr, t);                         // r = u v^3 (u v^7)^((p-5)/8)
    fe25_sqr(t, rr);
    fe25_mul(check, v, t);
    fe25_neg(nu, u);
    fe25_mul(nui, nu, sm1);
    const bool correct = fe25_eq(check, u), flipped = fe25_eq(check, nu), flipped_i = fe25_eq(check, nui);
    fe25 rp;
    fe25_mul(rp, rr, sm1);
    select256(rr, flipped || flipped_i, rp, rr);
    fe25_cneg(rr, rr, fe25_is_negative(rr));     // Absolute(): the even root
    fe25_cneg(rr, rr, sign);                     // "-0" stays 0 and is accepted, as in Go
    A.X = rr;
    A.Y = y;
    A.Z = one;
    fe25_mul(A.T, rr, y);
    return correct || flipped;
}

// L = 2^252 + 27742317777372353535851937790883648493
SBV_HD u256 ed_L() { u256 r = {{0x5CF5D3EDu, 0x5812631Au, 0xA2F79CD6u, 0x14DEF9DEu, 0x00000000u, 0x00000000u, 0x00000000u, 0x10000000u}}; return r; }

SBV_HD void pn_store(u32* dst, const pniels& p) {
    fe_store16(dst, p.YpX); fe_store16(dst + 8, p.YmX); fe_store16(dst + 16, p.Z); fe_store16(dst + 24, p.T2d);
}
SBV_HD void pn_load(pniels& p, const u32* src) {
    fe_load16(p.YpX, src); fe_load16(p.YmX, src + 8); fe_load16(p.Z, src + 16); fe_load16(p.T2d, src + 24);
}

// One tuple -> accept?  `w` indexes the tuple's 32 little-endian dwords, `qtab` = 8 x 32 dwords of
// private table space (16-byte aligned), `btab` = 32 x 128 affine-Niels multiples of B:
// btab[j*128 + (k-1)] = k * 2^(8j) * B.
template <typename Words>
SBV_HD bool ed25519_verify_lane(Words w, u32* qtab, const aniels* btab) {
    u32 renc[8], pk[8];
    u256 S, k;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { renc[i] = w[i]; S.v[i] = w[8 + i]; pk[i] = w[16 + i]; k.v[i] = w[24 + i]; }
    const u256 L = ed_L();
    bool ok = lt256(S, L) && lt256(k, L);        // SetCanonicalBytes(S); also covers sig[63] & 0xE0
    ept A;
    ok = ed_decompress(A, pk) && ok;
    // table of k * (-A), k = 1..8
    {
        ept nA = A;
        fe25_neg(nA.X, A.X);
        fe25_neg(nA.T, A.T);
        pniels one, cur;
        ed_to_pniels(one, nA);
        pn_store(qtab, one);
        ept t;
        ed_dbl(t, nA);
        ed_to_pniels(cur, t);
        pn_store(qtab + 32, cur);
        for (int i = 3; i <= SBV_ED_QTAB_ENTRIES; ++i) {
            ed_add_pniels(t, one, false, false);
            ed_to_pniels(cur, t);
            pn_store(qtab + (i - 1) * 32, cur);
        }
    }
    // signed windows: k < 2^253 so k + 0x88..8 and S + 0x80..80 do not carry out of 256 bits
    u256 kk, ss;
    (void)add_const_limbs(kk, k, 0x88888888u);
    (void)add_const_limbs(ss, S, 0x80808080u);
    ept R;
    ed_set_ident(R);
    for (int win = 63; win >= 0; --win) {
        SBV_NOUNROLL
        for (int t = 0; t < 4; ++t) ed_dbl(R, R);
        const int d = (int)((kk.v[win >> 3] >> ((win & 7) * 4)) & 15u) - 8;
        const int ad = d < 0 ? -d : d;
        pniels e;
        pn_load(e, qtab + (ad == 0 ? 0 : ad - 1) * 32);
        ed_add_pniels(R, e, d < 0, d == 0);
    }
    for (int j = 0; j < SBV_ED_BTAB_WINDOWS; ++j) {
        const int d = (int)((ss.v[j >> 2] >> ((j & 3) * 8)) & 255u) - 128;
        const int ad = d < 0 ? -d : d;
        const u32* bp = reinterpret_cast<const u32*>(btab + (size_t)j * SBV_ED_BTAB_PER_WINDOW + (ad == 0 ? 0 : ad - 1));
        aniels e;
        fe_load16(e.ypx, bp); fe_load16(e.ymx, bp + 8); fe_load16(e.xy2d, bp + 16);
        ed_add_aniels(R, e, d < 0, d == 0);
    }
    // encode(R) == R_enc, byte for byte
    fe25 zi, x, y;
    fe25_inv(zi, R.Z);
    fe25_mul(x, R.X, zi);
    fe25_mul(y, R.Y, zi);
    fe25_freeze(y, y);
    y.v[7] |= (fe25_is_negative(x) ? 1u : 0u) << 31;
    u32 diff = 0;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) diff |= y.v[i] ^ renc[i];
    return ok && diff == 0;
}

// ---- base-point comb (host, once per init; also tests/emul) --------------------------------------------
inline void build_ed_btable(aniels* out) {
    const fe25 bx = {{0x8F25D51Au, 0xC9562D60u, 0x9525A7B2u, 0x692CC760u, 0xFDD6DC5Cu, 0xC0A4E231u, 0xCD6E53FEu, 0x216936D3u}};
    const fe25 by = {{0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u}};
    const fe25 d2 = fe25_2d();
    ept base;
    base.X = bx; base.Y = by; base.Z = fe25_one(); fe25_mul(base.T, bx, by);
    for (int j = 0; j < SBV_ED_BTAB_WINDOWS; ++j) {
        pniels bn;
        ed_to_pniels(bn, base);
        ept t = base;
        for (int kx = 1; kx <= SBV_ED_BTAB_PER_WINDOW; ++kx) {
            if (kx > 1) ed_add_pniels(t, bn, false, false);
            fe25 zi, x, y;
            fe25_inv(zi, t.Z);
            fe25_mul(x, t.X, zi);
            fe25_mul(y, t.Y, zi);
            aniels a;
            fe25_add(a.ypx, y, x);
            fe25_sub(a.ymx, y, x);
            fe25_mul(a.xy2d, x, y);
            fe25_mul(a.xy2d, a.xy2d, d2);
            out[(size_t)j * SBV_ED_BTAB_PER_WINDOW + (kx - 1)] = a;
        }
        // next base = 2^8 * base = 2 * (128 * base)
        ept nb;
        ed_dbl(nb, t);
        fe25 zi;
        fe25_inv(zi, nb.Z);
        fe25_mul(base.X, nb.X, zi);
        fe25_mul(base.Y, nb.Y, zi);
        base.Z = fe25_one();
        fe25_mul(base.T, base.X, base.Y);
    }
}

}  // namespace sbv
