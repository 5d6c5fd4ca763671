// p256_keytab29.h — per-batch comb tables of the keys that repeat inside a batch, built on the carry-free field.
//
// A key's comb is ktab[j * 128 + (k-1)] = k * 2^(8j) * Q for j = 0..32, k = 1..128 (affine, canonical words of the
// R = 2^261 domain; window 32 only ever uses k = 1).  It is rebuilt inside every call for the keys that repeat often
// enough in that call (p256_group.h); nothing survives the call.  Three kernels, each a handful of lanes per unit of
// work, all latency-bound chains that run beside the throughput kernels (G phase / Q phase):
//
//   bases   one lane per key:    the doubling chain 2^(8j) Q (Jacobian, 3M + 5S each), recording B_j = 2^(8j) Q and 16 B_j
//                                for every window of the chunk, normalised to affine with ONE inversion per chunk
//                                (Montgomery's trick along the lane)
//   rows    two lanes per (key, window):   lane 0: the "babies" b * B_j, b = 1..16; lane 1: the "giants" 16 a * B_j,
//                                a = 2..8 — chains of XYZZ mixed additions (8M + 2S), normalised with one inversion per lane
//   fill    lanes per (key, window, rows of 16): entry 16 a + b = giant_a + baby_b as AFFINE + AFFINE additions sharing
//                                one inversion per lane (Montgomery's trick): 5M + 1S per entry instead of the
//                                ~17M + 4S of a Jacobian addition followed by a normalisation
//
// Exceptional cases: every sum formed here is (16 a + b) * B with 0 < 16 a + b <= 128 and B of prime order n > 2^255,
// so the two summands of an affine addition never share an x coordinate; the chains use the exact pt29_madd anyway.
// A key that pointFromAffine refuses (off the curve, coordinate >= p) gets valid = 0 and garbage tables nobody reads.
#pragma once
#include "p256_comb29.h"
#include "p256_group.h"

namespace sbv {

#define SBV_KT29_POINTS_PER_WINDOW 2                       // bases buffer: B_j and 16 B_j
#define SBV_KT29_STATE_WORDS 27                            // running Jacobian point between chunks (X, Y, Z limbs)
#define SBV_KT29_BASES_TMP_WORDS (66 * 36)                 // per key: up to 66 recorded points x (X, Y, Z, prefix product)
#define SBV_KT29_ROWS_TMP_WORDS (15 * 45)                  // per lane of the rows kernel: 15 points x (X, Y, ZZ, ZZZ, prefix)
#define SBV_KT29_FILL_TMP_WORDS (15 * 4 * 9)               // per lane of the fill kernel: up to 4 rows x 15 prefix products

SBV_HD void f29_store_raw(u32* dst, const fe29& a) {
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) dst[l] = (u32)a.v[l];
}
SBV_HD void f29_load_raw(fe29& a, const u32* src) {
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) a.v[l] = (i32)src[l];
}
SBV_HD void apt29_store_canon(apt* dst, const apt29& a) {
    u32 w[16];
    f29_store_canon(w, a.x);
    f29_store_canon(w + 8, a.y);
    struct alignas(16) q4 { u32 x, y, z, w; };
    q4* d = reinterpret_cast<q4*>(dst);
    SBV_UNROLL
    for (int k = 0; k < 4; ++k) { q4 v = {w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]}; d[k] = v; }
}

// ---- bases -------------------------------------------------------------------------------------------------------------
// bases[(gidx * 33 + j) * 2 + {0, 1}] = B_j, 16 B_j (affine, canonical).  jstate[gidx * 27 ..]: the chain between chunks
// (16 B_{j_last} after a chunk).  tmp: SBV_KT29_BASES_TMP_WORDS private words.
// valid: the byte of this key's TABLE SLOT (written by the first chunk)
SBV_HD void keytab29_bases_lane(const uint8_t* tuples, u32 gidx, const GroupState& g, u32* jstate, apt* bases, u32* tmp,
                                uint8_t* valid, int j_first, int j_last) {
    jpt29 t;
    u32* st = jstate + (size_t)gidx * SBV_KT29_STATE_WORDS;
    if (j_first == 0) {
        fe x, y;
        const bool ok = tuple_key_load(tuples, g.group_rep[gidx], x, y);
        *valid = ok ? 1 : 0;
        f29_from_fe(t.X, x);
        f29_from_fe(t.Y, y);
        t.Z = f29_one();
    } else {
        f29_load_raw(t.X, st); f29_load_raw(t.Y, st + 9); f29_load_raw(t.Z, st + 18);
    }
    fe29 acc = f29_one();
    int cnt = 0;
    SBV_NOUNROLL
    for (int j = j_first; j <= j_last; ++j) {
        SBV_NOUNROLL
        for (int half = 0; half < 2; ++half) {
            if (j > 0 || half > 0) {
                SBV_NOUNROLL
                for (int d = 0; d < 4; ++d) pt29_dbl_jac(t);
            }
            u32* rec = tmp + cnt * 36;
            f29_store_raw(rec, t.X); f29_store_raw(rec + 9, t.Y); f29_store_raw(rec + 18, t.Z); f29_store_raw(rec + 27, acc);
            f29_mul(acc, acc, t.Z);
            ++cnt;
        }
    }
    f29_store_raw(st, t.X); f29_store_raw(st + 9, t.Y); f29_store_raw(st + 18, t.Z);
    fe29 inv;
    f29_inv(inv, acc);
    apt* out = bases + ((size_t)gidx * SBV_GTAB_WINDOWS + j_first) * SBV_KT29_POINTS_PER_WINDOW;
    SBV_NOUNROLL
    for (int k = cnt - 1; k >= 0; --k) {
        const u32* rec = tmp + k * 36;
        fe29 X, Y, Z, pre, zi, zi2, zi3;
        f29_load_raw(X, rec); f29_load_raw(Y, rec + 9); f29_load_raw(Z, rec + 18); f29_load_raw(pre, rec + 27);
        f29_mul(zi, inv, pre);
        f29_mul(inv, inv, Z);
        f29_sqr(zi2, zi);
        f29_mul(zi3, zi2, zi);
        apt29 a;
        f29_mul(a.x, X, zi2);
        f29_mul(a.y, Y, zi3);
        apt29_store_canon(out + k, a);
    }
}

// ---- rows ----------------------------------------------------------------------------------------------------------------
// which = 0: babies b * B, b = 1..16 -> row[b - 1];  which = 1: giants 16 a * B, a = 2..8 -> row[16 a - 1].
// base2 = &bases[(key * 33 + j) * 2]; row = the window's 128 entries; tmp: SBV_KT29_ROWS_TMP_WORDS private words.
// top_window (j == 32): only entry 1 exists (the comb's carry digit is 0 or 1).
SBV_HD void keytab29_rows_lane(const apt* base2, int which, bool top_window, u32* tmp, apt* row) {
    apt29 step;
    apt29_load(step, reinterpret_cast<const u32*>(base2 + which));        // B or 16 B
    if (which == 0) apt29_store_canon(row, step);                         // entry 1 = B itself
    if (top_window) return;
    const int n = which == 0 ? 15 : 7;                                    // points of the chain beyond its first
    xyzz R;
    R.X = step.x; R.Y = step.y; R.ZZ = f29_one(); R.ZZZ = f29_one(); R.inf = false;
    fe29 acc = f29_one();
    SBV_NOUNROLL
    for (int k = 0; k < n; ++k) {
        pt29_madd(R, step, false);
        u32* rec = tmp + k * 45;
        f29_store_raw(rec, R.X); f29_store_raw(rec + 9, R.Y); f29_store_raw(rec + 18, R.ZZ); f29_store_raw(rec + 27, R.ZZZ);
        f29_store_raw(rec + 36, acc);
        f29_mul(acc, acc, R.ZZZ);
    }
    fe29 inv;
    f29_inv(inv, acc);
    SBV_NOUNROLL
    for (int k = n - 1; k >= 0; --k) {
        const u32* rec = tmp + k * 45;
        fe29 X, Y, ZZ, ZZZ, pre, i3, w, w2;
        f29_load_raw(X, rec); f29_load_raw(Y, rec + 9); f29_load_raw(ZZ, rec + 18); f29_load_raw(ZZZ, rec + 27); f29_load_raw(pre, rec + 36);
        f29_mul(i3, inv, pre);                  // 1 / ZZZ
        f29_mul(inv, inv, ZZZ);
        f29_mul(w, ZZ, i3);                     // ZZ / ZZZ = 1 / Z
        f29_sqr(w2, w);                         // 1 / ZZ
        apt29 a;
        f29_mul(a.x, X, w2);
        f29_mul(a.y, Y, i3);
        const int mult = which == 0 ? k + 2 : 16 * (k + 2);               // this point is mult * B
        apt29_store_canon(row + mult - 1, a);
    }
}

// ---- fill ----------------------------------------------------------------------------------------------------------------
// rows a = a_first .. a_last (within 1..7): entry 16 a + b = row[16 a - 1] + row[b - 1], b = 1..15.
// tmp: SBV_KT29_FILL_TMP_WORDS private words.
SBV_HD void keytab29_fill_lane(int a_first, int a_last, u32* tmp, apt* row) {
    fe29 acc = f29_one();
    int cnt = 0;
    SBV_NOUNROLL
    for (int a = a_first; a <= a_last; ++a) {
        apt29 G;
        apt29_load(G, reinterpret_cast<const u32*>(row + 16 * a - 1));
        SBV_NOUNROLL
        for (int b = 1; b <= 15; ++b) {
            apt29 S;
            apt29_load(S, reinterpret_cast<const u32*>(row + b - 1));
            fe29 d;
            f29_sub(d, S.x, G.x);
            f29_store_raw(tmp + cnt * 9, acc);
            f29_mul(acc, acc, d);
            ++cnt;
        }
    }
    fe29 inv;
    f29_inv(inv, acc);
    SBV_NOUNROLL
    for (int a = a_last; a >= a_first; --a) {
        apt29 G;
        apt29_load(G, reinterpret_cast<const u32*>(row + 16 * a - 1));
        SBV_NOUNROLL
        for (int b = 15; b >= 1; --b) {
            --cnt;
            apt29 S, r;
            apt29_load(S, reinterpret_cast<const u32*>(row + b - 1));
            fe29 d, pre, dinv;
            f29_sub(d, S.x, G.x);
            f29_load_raw(pre, tmp + cnt * 9);
            f29_mul(dinv, inv, pre);
            f29_mul(inv, inv, d);
            apt29_add_with_inverse(r, G, S, dinv);
            apt29_store_canon(row + 16 * a + b - 1, r);
        }
    }
}

}  // namespace sbv
