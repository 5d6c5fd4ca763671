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

struct ept { fe25 X, Y, Z, T; };            // coordinates tight (ed25519_fe.h) between operations
struct pniels { fe25 YpX, YmX, Z, T2d; };   // in registers; parked as 4 x 10 raw limbs (per-lane scratch)
struct aniels_r { fe25 ypx, ymx, xy2d; };   // in registers
struct aniels { u256 ypx, ymx, xy2d; };     // table entry: 96 bytes, canonical residues (the comb of B: 50 MB, cache resident)
#define SBV_ED_PT_WORDS 40                  // an extended or projective-Niels point as raw limbs

#define SBV_ED_QTAB_ENTRIES 8
// 16-bit comb of B: [S]B is 16 additions (an 8-bit comb needs 32).  16 x 32768 x 96 B = 50 MB: nothing for HBM / the
// 256 MB Infinity Cache (same trade as the P-256 G16 table).
#define SBV_ED_B16_WINDOWS 16
#define SBV_ED_B16_PER_WINDOW 32768
#define SBV_ED_B16_ENTRIES ((size_t)SBV_ED_B16_WINDOWS * SBV_ED_B16_PER_WINDOW)

SBV_HD void ed_set_ident(ept& p) { p.X = fe25_zero(); p.Y = fe25_one(); p.Z = fe25_one(); p.T = fe25_zero(); }

// r = 2p   (4S + 4M)
// with_t = false: T of the result is left as it was (a doubling reads X, Y, Z only: inside a run of doublings only the last one,
// in front of an addition, has to produce T — 7 multiplications instead of 8)
SBV_HD void ed_dbl(ept& r, const ept& p, bool with_t = true) {
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
    // limb bounds in units of "tight": h, g <= 2, e <= 3, f <= 4; the second operand of fe25_mul takes <= 3
    fe25_mul(r.X, f, e);
    fe25_mul(r.Y, h, g);
    fe25_mul(r.Z, f, g);
    if (with_t) fe25_mul(r.T, e, h);
}

// R += (+-)q for a projective-Niels q; no-op when skip   (8M)
SBV_HD void ed_add_pniels(ept& R, const pniels& q, bool neg, bool skip) {
    if (skip) return;                  // a zero digit: rare, so a branch, not 40 selects
    fe25 a, b, c, d, e, f, g, h, ypx, ymx, t2d;
    fe25_select(ypx, neg, q.YmX, q.YpX);
    fe25_select(ymx, neg, q.YpX, q.YmX);
    fe25_cneg(t2d, q.T2d, neg);
    fe25_sub(a, R.Y, R.X);
    fe25_mul(a, a, ymx);
    fe25_add(b, R.Y, R.X);
    fe25_mul(b, b, ypx);
    fe25_mul(c, R.T, t2d);
    fe25_mul(d, R.Z, q.Z);
    fe25_add(d, d, d);
    fe25_sub(e, b, a);                 // bounds: a, b, c tight; d, e, h <= 2; f, g <= 3
    fe25_sub(f, d, c);
    fe25_add(g, d, c);
    fe25_add(h, b, a);
    ept n;
    fe25_mul(n.X, f, e);
    fe25_mul(n.Y, g, h);
    fe25_mul(n.Z, f, g);
    fe25_mul(n.T, e, h);
    R = n;
}
// R += (+-)q for an affine-Niels q (Z2 = 1); no-op when skip   (7M).  q's limbs come from a table: [0, 2^w) = 2x tight.
SBV_HD void ed_add_aniels(ept& R, const aniels_r& q, bool neg, bool skip) {
    if (skip) return;                  // a zero digit: rare, so a branch, not 40 selects
    fe25 a, b, c, d, e, f, g, h, ypx, ymx, t2d;
    fe25_select(ypx, neg, q.ymx, q.ypx);
    fe25_select(ymx, neg, q.ypx, q.ymx);
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
    fe25_mul(n.X, f, e);
    fe25_mul(n.Y, g, h);
    fe25_mul(n.Z, f, g);
    fe25_mul(n.T, e, h);
    R = n;
}
SBV_HD void aniels_load(aniels_r& e, const aniels* p) {
    const u32* bp = reinterpret_cast<const u32*>(p);
    fe25_load_packed(e.ypx, bp); fe25_load_packed(e.ymx, bp + 8); fe25_load_packed(e.xy2d, bp + 16);
}
// Tried and dropped (profiles/r02/ed25519_ab_r02.txt): 64-byte (y + x, y - x) entries for the per-batch key combs with
// 2 d x y recomputed as (d / 2)((y + x)^2 - (y - x)^2).  The comb phases are bound by VALU issue, not by their gathers
// (SQ_INSTS_VALU / SQ_BUSY_CYCLES is the same 7.5 as in the P-256 comb kernel), so the extra 1M + 2S cost 8 % and
// the smaller table bought nothing.

// A table entry as fetched (24 words): held in registers while the previous addition runs, unpacked at use — the comb
// loops fetch one addition ahead, otherwise every addition waits out a random 96-byte gather (measured: the additions ran
// at ~14 k cycles per wavefront against ~6 k of issued instructions).
struct raw_aniels { u256 ypx, ymx, xy2d; };
SBV_HD void raw_u256_load(u256& o, const f25_q4* s) {
    const f25_q4 lo = s[0], hi = s[1];
    o.v[0] = lo.x; o.v[1] = lo.y; o.v[2] = lo.z; o.v[3] = lo.w; o.v[4] = hi.x; o.v[5] = hi.y; o.v[6] = hi.z; o.v[7] = hi.w;
}
SBV_HD void raw_aniels_load(raw_aniels& e, const aniels* p) {
    const f25_q4* s = reinterpret_cast<const f25_q4*>(p);
    raw_u256_load(e.ypx, s); raw_u256_load(e.ymx, s + 2); raw_u256_load(e.xy2d, s + 4);
}
SBV_HD void raw_aniels_unpack(aniels_r& q, const raw_aniels& e) {
    fe25_from_words(q.ypx, e.ypx.v); fe25_from_words(q.ymx, e.ymx.v); fe25_from_words(q.xy2d, e.xy2d.v);
}
SBV_HD void aniels_store(aniels* p, const aniels_r& e) {
    u32* bp = reinterpret_cast<u32*>(p);
    fe25_store_packed(bp, e.ypx); fe25_store_packed(bp + 8, e.ymx); fe25_store_packed(bp + 16, e.xy2d);
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
    fe25_from_words(y, w);                       // bit 255 is the sign; a non-canonical y (>= p) is accepted as y mod p
    fe25_carry(y, y);                            // unsigned limbs -> tight: A.Y feeds sums that must stay within the contracts
    const bool sign = (w[7] >> 31) != 0;
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
    fe25_select(rr, flipped || flipped_i, rp, rr);
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
    fe25_store_raw(dst, p.YpX); fe25_store_raw(dst + 10, p.YmX); fe25_store_raw(dst + 20, p.Z); fe25_store_raw(dst + 30, p.T2d);
}
SBV_HD void pn_load(pniels& p, const u32* src) {
    fe25_load_raw(p.YpX, src); fe25_load_raw(p.YmX, src + 10); fe25_load_raw(p.Z, src + 20); fe25_load_raw(p.T2d, src + 30);
}

// k.v[idx] without indexing the array by a run-time value: a dynamically indexed private array goes to scratch, and a
// scratch access inside the comb loops makes every iteration wait for ALL outstanding vector-memory operations — including the
// table entry fetched one addition ahead (vmcnt is in-order), which cost more than the additions themselves.
SBV_HD u32 ed_word_at(const u256& k, int idx) {
    u32 w = 0;
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) w = idx == l ? k.v[l] : w;
    return w;
}

// R += [S]B from the 16-bit comb b16[j * 32768 + (k-1)] = k * 2^(16j) * B; S < 2^253, so S + 0x8000...8000 does not
// carry out of 256 bits and its 16-bit digits minus 32768 are the signed digits.
SBV_HD void ed_add_sB(ept& R, const u256& S, const aniels* b16) {
    u256 ss;
    (void)add_const_limbs(ss, S, 0x80008000u);
    int d = (int)(ss.v[0] & 0xFFFFu) - 32768;
    raw_aniels cur;
    raw_aniels_load(cur, b16 + ((d < 0 ? -d : d) == 0 ? 0 : (d < 0 ? -d : d) - 1));
    SBV_NOUNROLL
    for (int j = 0; j < SBV_ED_B16_WINDOWS; ++j) {
        const int jn = j + 1 < SBV_ED_B16_WINDOWS ? j + 1 : j;
        const int dn = (int)((ed_word_at(ss, jn >> 1) >> ((jn & 1) * 16)) & 0xFFFFu) - 32768;
        const int adn = dn < 0 ? -dn : dn;
        raw_aniels nxt;
        raw_aniels_load(nxt, b16 + (size_t)jn * SBV_ED_B16_PER_WINDOW + (adn == 0 ? 0 : adn - 1));
        aniels_r e;
        raw_aniels_unpack(e, cur);
        ed_add_aniels(R, e, d < 0, d == 0);
        cur = nxt; d = dn;
    }
}

// encode(R) == renc (8 little-endian dwords), byte for byte
SBV_HD bool ed_encoding_matches(const ept& R, const u32* renc) {
    fe25 zi, x, y;
    fe25_inv_gcd(zi, R.Z);            // division steps (modinv30.h): ~4x cheaper than the z^(p-2) chain
    fe25_mul(x, R.X, zi);
    fe25_mul(y, R.Y, zi);
    u256 yw;
    fe25_freeze(yw, y);
    yw.v[7] |= (fe25_is_negative(x) ? 1u : 0u) << 31;
    u32 diff = 0;
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) diff |= yw.v[j] ^ renc[j];
    return diff == 0;
}

// One tuple -> accept?  `w` indexes the tuple's 32 little-endian dwords, `qtab` = 8 x 32 dwords of
// private table space (16-byte aligned), `btab` = the 16-bit comb of B (ed_add_sB).
// (xy != nullptr: X | Y of the key, 20 raw limbs, as a key check in front left them — the grouped step's ungrouped list, whose keys were
// all decompressed once already: the square root is not taken twice)
template <typename Words>
SBV_HD bool ed25519_verify_lane(Words w, u32* qtab, const aniels* btab, const u32* xy = nullptr) {
    u32 renc[8], pk[8];
    u256 S, k;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { renc[i] = w[i]; S.v[i] = w[8 + i]; pk[i] = w[16 + i]; k.v[i] = w[24 + i]; }
    const u256 L = ed_L();
    bool ok = lt256(S, L) && lt256(k, L);        // SetCanonicalBytes(S); also covers sig[63] & 0xE0
    ept A;
    if (xy) {
        fe25_load_raw(A.X, xy);
        fe25_load_raw(A.Y, xy + 10);
        A.Z = fe25_one();
        fe25_mul(A.T, A.X, A.Y);
    } else {
        ok = ed_decompress(A, pk) && ok;
    }
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
        pn_store(qtab + SBV_ED_PT_WORDS, cur);
        for (int i = 3; i <= SBV_ED_QTAB_ENTRIES; ++i) {
            ed_add_pniels(t, one, false, false);
            ed_to_pniels(cur, t);
            pn_store(qtab + (i - 1) * SBV_ED_PT_WORDS, cur);
        }
    }
    // signed windows: k < 2^253 so k + 0x88..8 and S + 0x80..80 do not carry out of 256 bits
    u256 kk;
    (void)add_const_limbs(kk, k, 0x88888888u);
    ept R;
    ed_set_ident(R);
    for (int win = 63; win >= 0; --win) {
        SBV_NOUNROLL
        for (int t = 0; t < 3; ++t) ed_dbl(R, R, false);
        ed_dbl(R, R);
        const int d = (int)((ed_word_at(kk, win >> 3) >> ((win & 7) * 4)) & 15u) - 8;
        const int ad = d < 0 ? -d : d;
        pniels e;
        pn_load(e, qtab + (ad == 0 ? 0 : ad - 1) * SBV_ED_PT_WORDS);
        ed_add_pniels(R, e, d < 0, d == 0);
    }
    ed_add_sB(R, S, btab);
    // encode(R) == R_enc, byte for byte
    return ok && ed_encoding_matches(R, renc);
}

// the grouped step's comb of B (ed25519_group.h: ed_add_sB_comb)
// pitch: bytes from one entry to the next — 96 (the entries packed, as the host builder writes them) or 128 (round 6: every entry in a
// 128-byte line of its own: a 96-byte entry at a 96-byte pitch straddles two lines of the 654 MB table in 3 positions of 4, and the G
// phase's gathers out of a table far beyond the 256 MB Infinity Cache moved 2.8 GB per launch for 1.3 GB of entries, profiles/traffic.json)
struct edcomb { const aniels* tab; int bits; int windows; u32 pitch; };
SBV_HD int edcomb_windows(int bits) { return (254 + bits - 1) / bits; }
SBV_HD size_t edcomb_entries(int bits) { return (size_t)edcomb_windows(bits) << (bits - 1); }
SBV_HD edcomb edcomb_make(const aniels* tab, int bits, u32 pitch = (u32)sizeof(aniels)) { edcomb c = {tab, bits, edcomb_windows(bits), pitch}; return c; }
SBV_HD const aniels* edcomb_entry(const edcomb& c, size_t index) {
    return reinterpret_cast<const aniels*>(reinterpret_cast<const uint8_t*>(c.tab) + index * c.pitch);
}

#define SBV_ED_UNGXY_WORDS 20     // ed25519_group.h: X | Y of an ungrouped tuple's key, from the key check to the quad form of the one-lane kernel
// hot keys (ed25519_group.h): the 16-bit comb of -A of a promoted cache slot
#define SBV_ED_HOT_BITS 16
#define SBV_ED_HOT_WINDOWS 16
#define SBV_ED_HOT_PER_WINDOW 32768u
#define SBV_ED_HOT_PITCH 128u
#define SBV_ED_HOT_COMB_BYTES ((size_t)SBV_ED_HOT_WINDOWS * SBV_ED_HOT_PER_WINDOW * SBV_ED_HOT_PITCH)      // 64 MiB
#define SBV_ED_HOT_LANE_ENTRIES 32
#define SBV_ED_HOT_PARTS (SBV_ED_HOT_PER_WINDOW / SBV_ED_HOT_LANE_ENTRIES)                                 // 1024 lanes per window
#define SBV_ED_HOT_TMP_WORDS (SBV_ED_HOT_LANE_ENTRIES * 40)                                                 // per resident lane (SBV_ED_WINDOW_TMP_WORDS each)
#define SBV_ED_HOT_BUILD_BLOCKS 2048u                                                                      // the builder's grid: 64 lanes each, 8 wavefronts per CU

// ---- base-point comb (host, once per init; also tests/emul) --------------------------------------------
// window j of a `bits`-wide comb of B: out_row[k - 1] = k * 2^(bits j) * B for k = 1 .. 2^(bits-1), canonical affine-Niels entries.
// Callable from several host threads; the row is produced in blocks of 32 768 entries (one inversion each) so that a 20-bit window
// (524 288 entries) needs no more host memory than a 16-bit one.
// (build_ed_window_of: the same row for any point P — the host reference of the hot keys' combs of -A, ed25519_group.h)
inline void build_ed_window_of(const ept& P, int bits, int j, aniels* out_row) {
    const fe25 d2 = fe25_2d();
    ept base = P;
    for (int i = 0; i < bits * j; ++i) ed_dbl(base, base);          // 2^(bits j) * P, projective
    pniels bn;
    ed_to_pniels(bn, base);
    const size_t total = (size_t)1 << (bits - 1);
    const size_t blk = total < 32768 ? total : 32768;
    fe25* X = new fe25[blk]; fe25* Y = new fe25[blk]; fe25* Z = new fe25[blk]; fe25* pre = new fe25[blk];
    ept t = base;
    for (size_t b0 = 0; b0 < total; b0 += blk) {
        fe25 acc = fe25_one();
        for (size_t k = 0; k < blk; ++k) {                           // (b0 + k + 1) * base, Z's multiplied up for Montgomery's trick
            if (b0 + k > 0) ed_add_pniels(t, bn, false, false);
            X[k] = t.X; Y[k] = t.Y; Z[k] = t.Z;
            pre[k] = acc;
            fe25_mul(acc, acc, t.Z);
        }
        fe25 inv;
        fe25_inv(inv, acc);
        for (size_t k = blk; k-- > 0;) {
            fe25 zi, x, y;
            fe25_mul(zi, inv, pre[k]);
            fe25_mul(inv, inv, Z[k]);
            fe25_mul(x, X[k], zi);
            fe25_mul(y, Y[k], zi);
            aniels_r a;
            fe25_add(a.ypx, y, x);
            fe25_sub(a.ymx, y, x);
            fe25_mul(a.xy2d, x, y);
            fe25_mul(a.xy2d, a.xy2d, d2);
            aniels_store(out_row + b0 + k, a);
        }
    }
    delete[] X; delete[] Y; delete[] Z; delete[] pre;
}
inline void build_ed_b_window(int bits, int j, aniels* out_row) {
    const u32 bxw[8] = {0x8F25D51Au, 0xC9562D60u, 0x9525A7B2u, 0x692CC760u, 0xFDD6DC5Cu, 0xC0A4E231u, 0xCD6E53FEu, 0x216936D3u};
    const u32 byw[8] = {0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u};
    fe25 bx, by;
    fe25_from_words(bx, bxw); fe25_carry(bx, bx);
    fe25_from_words(by, byw); fe25_carry(by, by);
    ept base;
    base.X = bx; base.Y = by; base.Z = fe25_one(); fe25_mul(base.T, bx, by);
    build_ed_window_of(base, bits, j, out_row);
}
inline void build_ed_b16_window(int j, aniels* out_row) { build_ed_b_window(16, j, out_row); }

}  // namespace sbv
