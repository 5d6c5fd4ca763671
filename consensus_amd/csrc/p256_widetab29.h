// p256_widetab29.h — the wide comb of a registered key (p256_comb29.h: widekeys), built on the device.
//
// A `bits`-wide comb is W = ceil(257 / bits) windows of H = 2^(bits-1) affine entries, tab[(j << (bits-1)) + (m-1)] = m * B_j with
// B_j = 2^(bits j) * Q — 557 056 entries (35.7 MB) at 16 bits, 6.8 M (436 MB) at 20.  On the host that is 0.1 - 1.3 s per key
// (sbv_api.hip: the first form of sbv_p256_widen_keys); here it is two launches for any number of keys:
//
//   chains   two lanes per (key, window).  With babies = 2^hb, hb = bits / 2, and giants = H / babies:
//              role 0: b * B_j, b = 1 .. babies - 1      -> entries [0, babies - 1) of the window
//              role 1: g * C_j, g = 1 .. giants, C_j = babies * B_j   -> the entries m = g * babies
//            as chains of exact XYZZ mixed additions (pt29_madd) normalised with ONE inversion per lane (Montgomery's trick; the
//            raw records and running prefixes park in a scratch strip).  B_j and C_j come from the host: 256 + W * hb doublings
//            per key are microseconds there and a ~2 ms serial chain on one lane here.
//   fill     every other entry m = g * babies + b, 1 <= g < giants, 1 <= b < babies, is giant_g + baby_b as an AFFINE + AFFINE
//            addition; a lane takes SBV_WIDETAB_T consecutive b of one giant and shares one inversion among their denominators
//            x_b - x_g (the running prefix parks in the 64-byte slot the entry itself will occupy): 6 multiplications per entry.
//
// Exceptional cases: a denominator is 0 only if b * B = +- g * babies * B, impossible for 0 < b < babies <= g * babies with
// g * babies + b <= H < n (B has prime order n > 2^255); the chains use the exact addition anyway (their second step is a doubling).
// A key that is no point of the curve gets no table at all (kvalid = 0 rejects its signatures; the host zeroes the comb).
// The result is byte for byte what host_build_wide_key_table produces (canonical residues): tests/emul and the GPU tier compare them.
#pragma once
#include "p256_comb29.h"
#include "p256_keytab29.h"

namespace sbv {

#define SBV_WIDETAB_T 16                           // babies per fill lane (one inversion each)
#define SBV_WIDETAB_REC_WORDS 45                   // chain scratch per element: X, Y, ZZ, ZZZ raw limbs + the running prefix

struct widebuild { int bits, windows, hb; u32 babies, giants; size_t per_window; };
SBV_HD widebuild widebuild_make(int bits) {
    widebuild w;
    w.bits = bits;
    w.windows = (257 + bits - 1) / bits;
    w.hb = bits / 2;
    w.babies = 1u << w.hb;
    w.giants = 1u << (bits - 1 - w.hb);
    w.per_window = (size_t)1 << (bits - 1);
    return w;
}
SBV_HD u32 widebuild_chain_len(const widebuild& w) { return w.babies - 1 > w.giants ? w.babies - 1 : w.giants; }     // longest chain: scratch elements per chain lane
SBV_HD u32 widebuild_fill_chunks(const widebuild& w) { return (w.babies - 1 + SBV_WIDETAB_T - 1) / SBV_WIDETAB_T; }

// k * P for k = 1 .. count into out[(k - 1) * stride] (affine, canonical); P affine canonical; tmp: count x SBV_WIDETAB_REC_WORDS words
SBV_HD void widetab_chain_lane(const apt* base, u32 count, u32* tmp, apt* out, size_t stride) {
    if (count == 0) return;
    apt29 P;
    apt29_load(P, reinterpret_cast<const u32*>(base));
    out[0] = *base;
    xyzz T;
    T.X = P.x; T.Y = P.y; T.ZZ = f29_one(); T.ZZZ = f29_one(); T.inf = false;
    fe29 acc = f29_one();
    SBV_NOUNROLL
    for (u32 k = 2; k <= count; ++k) {
        pt29_madd(T, P, false);                 // k = 2 is P + P: the doubling branch
        u32* rec = tmp + (size_t)(k - 2) * SBV_WIDETAB_REC_WORDS;
        f29_store_raw(rec, T.X); f29_store_raw(rec + 9, T.Y); f29_store_raw(rec + 18, T.ZZ); f29_store_raw(rec + 27, T.ZZZ);
        f29_store_raw(rec + 36, acc);
        f29_mul(acc, acc, T.ZZZ);
    }
    if (count < 2) return;
    fe29 inv;
    f29_inv(inv, acc);
    SBV_NOUNROLL
    for (u32 k = count; k >= 2; --k) {
        const u32* rec = tmp + (size_t)(k - 2) * SBV_WIDETAB_REC_WORDS;
        fe29 X, Y, ZZ, ZZZ, pre, i3, wv, w2;
        f29_load_raw(X, rec); f29_load_raw(Y, rec + 9); f29_load_raw(ZZ, rec + 18); f29_load_raw(ZZZ, rec + 27); f29_load_raw(pre, rec + 36);
        f29_mul(i3, inv, pre);                  // 1 / ZZZ_k
        f29_mul(inv, inv, ZZZ);
        f29_mul(wv, ZZ, i3);                    // ZZ / ZZZ = 1 / Z
        f29_sqr(w2, wv);
        apt29 a;
        f29_mul(a.x, X, w2);
        f29_mul(a.y, Y, i3);
        apt29_store_canon(out + (size_t)(k - 1) * stride, a);
    }
}
// one chain lane of window `row` (H entries): role 0 the babies from B, role 1 the giants from C
SBV_HD void widetab_chain_role(const widebuild& w, const apt* B, const apt* C, int role, u32* tmp, apt* row) {
    if (role == 0) widetab_chain_lane(B, w.babies - 1, tmp, row, 1);
    else widetab_chain_lane(C, w.giants, tmp, row + (w.babies - 1), w.babies);      // entry m = g * babies sits at index g * babies - 1
}

// Hot keys (p256_group.h): base point e of promotion i, gathered from the 8-bit comb of its cache slot.  e < W: B_j = 2^(16 j) Q = entry 1
// of row 2 j; e >= W: C_j = 2^8 B_j = entry 1 of row 2 j + 1 (rows and full tables alike hold entry 1 of every row).  The top window
// (j = 16) only ever serves its entry 1, the carry of the signed recoding of a scalar below 2^255; its giants are never read, and
// C_16 = Q keeps every sum of the builder's fill there well defined (g Q = +- b 2^256 Q would need g = +- b (2^256 mod n) mod n, far
// outside 1..127).  16-bit combs only (two 8-bit rows per window).
SBV_HD void promote_base_lane(u32 i, u32 e, const u32* plist, const apt* ktab, apt* pbases) {
    const u32 W = (257 + 16 - 1) / 16;          // 17
    const u32 slot = plist[2 * i];
    if (slot == 0xFFFFFFFFu) return;
    const apt* tab = ktab + (size_t)slot * (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW);
    const u32 j = e < W ? e : e - W;
    const u32 row = e < W ? 2 * j : (j + 1 < W ? 2 * j + 1 : 0u);
    pbases[(size_t)i * 2 * W + e] = tab[(size_t)row * SBV_GTAB_PER_WINDOW];
}

// entries m = g * babies + b for b = b0 .. b0 + SBV_WIDETAB_T - 1 (clipped to babies - 1) of one window, 1 <= g < giants
SBV_HD void widetab_fill_lane(const widebuild& w, u32 g, u32 b0, apt* row) {
    const u32 b1 = b0 + SBV_WIDETAB_T - 1 < w.babies - 1 ? b0 + SBV_WIDETAB_T - 1 : w.babies - 1;
    if (b0 > b1) return;
    apt29 G;
    apt29_load(G, reinterpret_cast<const u32*>(row + (size_t)g * w.babies - 1));
    apt* dst = row + (size_t)g * w.babies - 1;                 // dst[b] = entry g * babies + b
    fe29 acc = f29_one();
    SBV_NOUNROLL
    for (u32 b = b0; b <= b1; ++b) {
        apt29 S;
        apt29_load(S, reinterpret_cast<const u32*>(row + b - 1));
        fe29 d;
        f29_sub(d, S.x, G.x);
        f29_store_raw(reinterpret_cast<u32*>(dst + b), acc);   // the slot the entry will occupy holds its prefix until then
        f29_mul(acc, acc, d);
    }
    fe29 inv;
    f29_inv(inv, acc);
    SBV_NOUNROLL
    for (u32 b = b1; b >= b0; --b) {
        apt29 S, r;
        apt29_load(S, reinterpret_cast<const u32*>(row + b - 1));
        fe29 d, pre, dinv;
        f29_sub(d, S.x, G.x);
        f29_load_raw(pre, reinterpret_cast<const u32*>(dst + b));
        f29_mul(dinv, inv, pre);
        f29_mul(inv, inv, d);
        apt29_add_with_inverse(r, G, S, dinv);
        apt29_store_canon(dst + b, r);
    }
}

}  // namespace sbv
