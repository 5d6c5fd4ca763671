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
// 4-bit windows from a per-signature table of projective-Niels points in HBM, [S]B by 16 signed
// 16-bit comb windows of affine-Niels points (50 MB, HBM / Infinity Cache resident).  One inversion per lane to
// re-encode R (the comparison is byte-wise by specification).
#pragma once
#include "ed25519_fe.h"
#include "p256_core.h"   // fe_store16 / fe_load16 / add_const_limbs / vec4

namespace sbv {

struct ept { fe25 X, Y, Z, T; };
struct pniels { fe25 YpX, YmX, Z, T2d; };   // 128 bytes
struct aniels { fe25 ypx, ymx, xy2d; };     // 96 bytes

#define SBV_ED_QTAB_ENTRIES 8
// 16-bit comb of B: [S]B is 16 additions (an 8-bit comb needs 32).  16 x 32768 x 96 B = 50 MB: nothing for HBM / the
// 256 MB Infinity Cache (same trade as the P-256 G16 table).
#define SBV_ED_B16_WINDOWS 16
#define SBV_ED_B16_PER_WINDOW 32768
#define SBV_ED_B16_ENTRIES ((size_t)SBV_ED_B16_WINDOWS * SBV_ED_B16_PER_WINDOW)

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

// R += [S]B from the 16-bit comb b16[j * 32768 + (k-1)] = k * 2^(16j) * B; S < 2^253, so S + 0x8000...8000 does not
// carry out of 256 bits and its 16-bit digits minus 32768 are the signed digits.
SBV_HD void ed_add_sB(ept& R, const u256& S, const aniels* b16) {
    u256 ss;
    (void)add_const_limbs(ss, S, 0x80008000u);
    SBV_NOUNROLL
    for (int j = 0; j < SBV_ED_B16_WINDOWS; ++j) {
        const int d = (int)((ss.v[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu) - 32768;
        const int ad = d < 0 ? -d : d;
        const u32* bp = reinterpret_cast<const u32*>(b16 + (size_t)j * SBV_ED_B16_PER_WINDOW + (ad == 0 ? 0 : ad - 1));
        aniels e;
        fe_load16(e.ypx, bp); fe_load16(e.ymx, bp + 8); fe_load16(e.xy2d, bp + 16);
        ed_add_aniels(R, e, d < 0, d == 0);
    }
}

// One tuple -> accept?  `w` indexes the tuple's 32 little-endian dwords, `qtab` = 8 x 32 dwords of
// private table space (16-byte aligned), `btab` = the 16-bit comb of B (ed_add_sB).
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
    u256 kk;
    (void)add_const_limbs(kk, k, 0x88888888u);
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
    ed_add_sB(R, S, btab);
    // encode(R) == R_enc, byte for byte
    fe25 zi, x, y;
    fe25_inv_gcd(zi, R.Z);            // division steps (modinv30.h): ~4x cheaper than the z^(p-2) chain
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
// window j (0..15) of the 16-bit comb, callable from several host threads
inline void build_ed_b16_window(int j, aniels* out_row) {
    const fe25 bx = {{0x8F25D51Au, 0xC9562D60u, 0x9525A7B2u, 0x692CC760u, 0xFDD6DC5Cu, 0xC0A4E231u, 0xCD6E53FEu, 0x216936D3u}};
    const fe25 by = {{0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u}};
    const fe25 d2 = fe25_2d();
    ept base;
    base.X = bx; base.Y = by; base.Z = fe25_one(); fe25_mul(base.T, bx, by);
    for (int i = 0; i < 16 * j; ++i) ed_dbl(base, base);          // 2^(16j) * B, projective
    pniels bn;
    ed_to_pniels(bn, base);
    const int n = SBV_ED_B16_PER_WINDOW;
    fe25* X = new fe25[n]; fe25* Y = new fe25[n]; fe25* Z = new fe25[n]; fe25* pre = new fe25[n];
    ept t = base;
    fe25 acc = fe25_one();
    for (int k = 0; k < n; ++k) {                                  // (k+1) * base, Z's multiplied up for Montgomery's trick
        if (k > 0) ed_add_pniels(t, bn, false, false);
        X[k] = t.X; Y[k] = t.Y; Z[k] = t.Z;
        pre[k] = acc;
        fe25_mul(acc, acc, t.Z);
    }
    fe25 inv;
    fe25_inv(inv, acc);
    for (int k = n - 1; k >= 0; --k) {
        fe25 zi, x, y;
        fe25_mul(zi, inv, pre[k]);
        fe25_mul(inv, inv, Z[k]);
        fe25_mul(x, X[k], zi);
        fe25_mul(y, Y[k], zi);
        aniels a;
        fe25_add(a.ypx, y, x);
        fe25_sub(a.ymx, y, x);
        fe25_mul(a.xy2d, x, y);
        fe25_mul(a.xy2d, a.xy2d, d2);
        out_row[k] = a;
    }
    delete[] X; delete[] Y; delete[] Z; delete[] pre;
}

}  // namespace sbv
