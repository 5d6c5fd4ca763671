// p256_comb29.h — the comb phases of stage B over the carry-free field (p256_fe29.h, p256_pt29.h):
//
//   gphase29_lane        R = u1 * G from the comb of G (13 mixed additions with 20-bit windows, 17 with 16-bit), parked in gacc
//   qphase29_lane        R += windows [j0, j1) of u2 * Q from a key's 8-bit comb; the last chunk checks R.x == r
//   verify29_lane_keyed  both for a registered key in one pass (50 mixed additions, no doublings)
//
// Same verdicts as verify_lane / verify_lane_keyed of p256_core.h (every addition exact), the arithmetic that
// crypto/ecdsa.VerifyASN1 performs per signature behind the reference's api.Verifier (pkg/api/dependencies.go:54-71).
// Tables hold affine points as 64-byte entries x | y, each coordinate the canonical 8-word residue of the
// R = 2^261 Montgomery domain (f29_store_canon); the accumulator is XYZZ, parked between launches as 36 raw limbs
// per tuple, limb-major (word w of tuple i at gacc[w * cap + i]); infinity is parked as ZZ = 0 in every limb.
#pragma once
#include "p256_core.h"
#include "p256_pt29.h"

namespace sbv {

#define SBV_GACC29_WORDS 36

SBV_HD void gacc29_store(u32* gacc, size_t cap, size_t i, const xyzz& R) {
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) {
        gacc[(size_t)l * cap + i] = (u32)R.X.v[l];
        gacc[(size_t)(9 + l) * cap + i] = (u32)R.Y.v[l];
        gacc[(size_t)(18 + l) * cap + i] = R.inf ? 0u : (u32)R.ZZ.v[l];
        gacc[(size_t)(27 + l) * cap + i] = (u32)R.ZZZ.v[l];
    }
}
SBV_HD void gacc29_load(xyzz& R, const u32* gacc, size_t cap, size_t i) {
    SBV_UNROLL
    for (int l = 0; l < 9; ++l) {
        R.X.v[l] = (i32)gacc[(size_t)l * cap + i];
        R.Y.v[l] = (i32)gacc[(size_t)(9 + l) * cap + i];
        R.ZZ.v[l] = (i32)gacc[(size_t)(18 + l) * cap + i];
        R.ZZZ.v[l] = (i32)gacc[(size_t)(27 + l) * cap + i];
    }
    R.inf = f29_limbs_all_zero(R.ZZ);       // a finite point has ZZ != 0 (mod p), so never the all-zero limbs
}

// A table entry as fetched (16 words): held in registers while the previous addition runs, unpacked at use.
struct alignas(16) raw_apt { u32 w[16]; };
SBV_HD void raw_apt_load(raw_apt& e, const apt* p) {
    struct alignas(16) q4 { u32 x, y, z, w; };
    const q4* s = reinterpret_cast<const q4*>(p);
    const q4 a = s[0], b = s[1], c = s[2], d = s[3];
    e.w[0] = a.x; e.w[1] = a.y; e.w[2] = a.z; e.w[3] = a.w; e.w[4] = b.x; e.w[5] = b.y; e.w[6] = b.z; e.w[7] = b.w;
    e.w[8] = c.x; e.w[9] = c.y; e.w[10] = c.z; e.w[11] = c.w; e.w[12] = d.x; e.w[13] = d.y; e.w[14] = d.z; e.w[15] = d.w;
}
SBV_HD void raw_apt_unpack(apt29& q, const raw_apt& e) {
    f29_unpack(q.x, e.w);
    f29_unpack(q.y, e.w + 8);
}

// The fixed-base comb of G for the carry-free kernels: `bits`-wide signed windows, windows = ceil(257 / bits) rows of
// 2^(bits-1) entries, tab[(j << (bits-1)) + (m-1)] = m * 2^(bits j) * G (R = 2^261 domain).  u1 * G is `windows` mixed
// additions: 17 for the 16-bit comb (35.7 MB), 13 for the 20-bit comb (436 MB) — the table is static, HBM is 288 GB,
// and a random 64-byte gather per addition is prefetched one addition ahead, so wider windows only cost memory.
struct gcomb { const apt* tab; int bits; int windows; };
SBV_HD gcomb gcomb_make(const apt* tab, int bits) { gcomb g = {tab, bits, (257 + bits - 1) / bits}; return g; }
SBV_HD size_t gcomb_entries(int bits) { return (size_t)((257 + bits - 1) / bits) << (bits - 1); }
struct u288 { u32 v[9]; };
// k = u + sum_j 2^(bits j + bits - 1): window j of k, minus 2^(bits-1), is the signed digit (no carry chain between digits)
SBV_HD void gcomb_recode(u288& k, const u256& u, int bits, int windows) {
    u32 off[9];
    SBV_UNROLL
    for (int w = 0; w < 9; ++w) off[w] = 0;
    SBV_NOUNROLL
    for (int j = 0; j < windows; ++j) {
        const int pos = bits * j + bits - 1;
        SBV_UNROLL
        for (int w = 0; w < 9; ++w) off[w] |= (pos >> 5) == w ? (1u << (pos & 31)) : 0u;
    }
    u32 c = 0;
    SBV_UNROLL
    for (int w = 0; w < 8; ++w) k.v[w] = addc(u.v[w], off[w], c);
    k.v[8] = off[8] + c;
}
SBV_HD void gcomb_digit(const u288& k, int bits, int j, u32& idx, bool& neg, bool& skip) {
    const int pos = bits * j, word = pos >> 5, sh = pos & 31;
    u32 lo = 0, hi = 0;
    SBV_UNROLL
    for (int w = 0; w < 9; ++w) {
        lo = word == w ? k.v[w] : lo;
        hi = word + 1 == w ? k.v[w] : hi;
    }
    const u64 two = ((u64)hi << 32) | lo;
    const u32 win = (u32)(two >> sh) & ((1u << bits) - 1u);
    const i32 d = (i32)win - (i32)(1u << (bits - 1));
    const i32 ad = d < 0 ? -d : d;
    idx = ad == 0 ? 0u : (u32)ad - 1u;
    neg = d < 0;
    skip = d == 0;
}

// R = u1 * G   (add_to_R: R += u1 * G; flip: every digit's sign is inverted, i.e. R (+)= u1 * (-P) for the comb of P)
SBV_HD bool wave_any(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __any(x) != 0;
#else
    return x;
#endif
}
SBV_HD bool wave_all(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __all(x) != 0;
#else
    return x;
#endif
}
// (j0, j1): only the windows [j0, j1) — the key-comb chunks of the grouped step; top_on_demand: the top window holds nothing but the carry of
// the signed recoding, and is walked only when a lane of the wavefront carries (wave-uniform loop bound)
SBV_HD void gphase29_point(xyzz& R, const u256& u1, const gcomb& gc, bool add_to_R = false, bool flip = false, int j0 = 0, int j1 = -1,
                           bool top_on_demand = false) {
    u288 k1;
    gcomb_recode(k1, u1, gc.bits, gc.windows);
    if (!add_to_R) pt29_set_inf(R);
    if (j1 < 0) j1 = gc.windows;
    u32 idx; bool neg, skip;
    if (top_on_demand && j1 == gc.windows) {
        gcomb_digit(k1, gc.bits, gc.windows - 1, idx, neg, skip);
        if (!wave_any(!skip)) j1 = gc.windows - 1;
    }
    if (j0 >= j1) return;
    gcomb_digit(k1, gc.bits, j0, idx, neg, skip);
    raw_apt cur;
    raw_apt_load(cur, gc.tab + ((size_t)j0 << (gc.bits - 1)) + idx);
    SBV_NOUNROLL
    for (int j = j0; j < j1; ++j) {
        const int jn = j + 1 < j1 ? j + 1 : j1 - 1;
        u32 idxn; bool negn, skipn;
        gcomb_digit(k1, gc.bits, jn, idxn, negn, skipn);
        raw_apt nxt;
        raw_apt_load(nxt, gc.tab + ((size_t)jn << (gc.bits - 1)) + idxn);
        if (!skip) {
            apt29 q;
            raw_apt_unpack(q, cur);
            pt29_madd(R, q, neg != flip);
        }
        cur = nxt; neg = negn; skip = skipn;
    }
}
SBV_HD void gphase29_lane(const Scratch& s, size_t i, const gcomb& gc, u32* gacc) {
    u256 u1;
    soa_load(u1, s.u1, s.cap, i);
    xyzz R;
    gphase29_point(R, u1, gc);
    gacc29_store(gacc, s.cap, i, R);
}

// Key-sorted grouped step: lane L of the sorted list handles tuple t; scalars come from t's record (Scratch::rec), the
// accumulator lives at the lane's own position L, so that the lanes of a wavefront (all of one key) stay coalesced.
SBV_HD void gphase29_lane_sorted(const Scratch& s, size_t t, size_t L, const gcomb& gc, u32* gacc) {
    u256 u1;
    rec_load256(u1, s.rec, t, SBV_REC_U1);
    xyzz R;
    gphase29_point(R, u1, gc);
    gacc29_store(gacc, s.cap, L, R);
}

// R += sum of windows [j0, j1) of u2 * Q; qtab[j * 128 + (k-1)] = k * 2^(8 j) * Q, j = 0..32
// Window 32 holds the carry of the signed recoding (digit 0 or 1).  For a uniformly random u2 it is 1 about half the time, so
// every wavefront paid a 33rd addition.  u2 * Q = (n - u2) * (-Q) (Q has order n), so a scalar with its top bit set is replaced
// by n - u2 < 2^255 with every digit's sign flipped: the carry then needs a top byte of 0x7F (0.4 % of the lanes), and the loop
// bound becomes wave-uniform — a wavefront runs window 32 only if one of its lanes carries (22 % of the wavefronts).

// Wide combs of the consenters' keys (round 4).  The consenters of a SmartBFT cluster are a handful of keys that sign every vote of
// every decision for a whole epoch (pkg/types/types.go:25-29; internal/bft/view.go:531-541, 631, 834): a registered slot the host
// marks as such (sbv_p256_widen_keys) owns, besides its 8-bit comb, a `bits`-wide one laid out exactly like the comb of G (gcomb) —
// tab[idx[slot] * stride + (j << (bits-1)) + (m-1)] = m * 2^(bits j) * Q — so u2 * Q costs ceil(257 / bits) additions instead of
// 33 (16 bits: 35.7 MB per key and 16 additions once the sign trick below has emptied the carry window; 20 bits: 436 MB and 13;
// HBM holds 288 GB).  idx[slot] = SBV_WIDE_NONE for every other slot.  A wavefront takes the wide path only when ALL its lanes hold
// wide slots (wave-uniform: no lane pays for both loops), any other wavefront uses the 8-bit combs every key keeps.
#define SBV_WIDE_NONE 0xFFFFFFFFu
struct widekeys { const apt* tab; const u32* idx; size_t stride; int bits; int windows; };
SBV_HD widekeys widekeys_none() { widekeys w = {nullptr, nullptr, 0, 16, 17}; return w; }
SBV_HD widekeys widekeys_make(const apt* tab, const u32* idx, int bits) {
    widekeys w = {tab, tab ? idx : nullptr, gcomb_entries(bits), bits, (257 + bits - 1) / bits};
    return w;
}
// slot: already clamped below the registry's size
SBV_HD u32 widekeys_index(const widekeys& wk, u32 slot) { return wk.idx ? wk.idx[slot] : SBV_WIDE_NONE; }
// R += u2 * Q from wide comb number widx.  u2 * Q = (n - u2) * (-Q): a scalar with its top bit set is replaced by n - u2 < 2^255 with
// every digit's sign flipped (as qphase29_point does), so that the top window of a comb whose width divides 256 is a carry that
// almost never comes (the loop skips it per lane, a wavefront pays for it only if one of its lanes carries).
SBV_HD void wide_qphase29_point(xyzz& R, const u256& u2in, const widekeys& wk, u32 widx) {
    const bool flip = (u2in.v[7] >> 31) != 0;
    u256 u2, nmu;
    (void)sub256(nmu, sc_n(), u2in);
    select256(u2, flip, nmu, u2in);
    gcomb kc = {wk.tab + (size_t)widx * wk.stride, wk.bits, wk.windows};
    gphase29_point(R, u2, kc, true, flip);
}

SBV_HD void qphase29_point(xyzz& R, const u256& u2in, const apt* qtab, int j0, int j1) {
    const bool flip = (u2in.v[7] >> 31) != 0;
    u256 u2, nmu;
    (void)sub256(nmu, sc_n(), u2in);                  // u2in < n always (stage A reduces it); garbage for an out-of-range lane, whose verdict is already false
    select256(u2, flip, nmu, u2in);
    // a key's 8-bit comb IS a comb in the layout of the comb of G (gcomb) with bits = 8 and 33 windows: the same walker (round 6 — the
    // dedicated loop of rounds 2-5 indexed the recoded scalar's register array dynamically and needed 247 VGPRs; this one fits 168)
    const gcomb kc = {qtab, 8, SBV_GTAB_WINDOWS};
    gphase29_point(R, u2, kc, true, flip, j0, j1, true);
}
// ---- the NARROW view of a key's comb (round 5) ------------------------------------------------------------------------------------
// The rows step of the table builder (p256_keytab29.h) leaves, in every 128-entry window row j, the babies b * B_j (b = 1..8, entries
// 0..7) and the giants 16 a * B_j (a = 1..8, entries 15, 31, ..., 127); the fill step derives the other 112 entries from them and costs
// three quarters of a table.  Babies and giants alone ARE a comb with 4-bit windows — b * 2^(8j) Q and a * 2^(8j+4) Q — so a key
// that signs too few tuples of a batch to earn a full table is verified from its rows alone: the same signed 8-bit digit d of u2,
// |d| = 16 a + b with a = (|d| + 7) >> 4 in 0..8 and b in -7..8, costs two exact additions (giant, then baby; zero parts skipped:
// 1.83 per window on average) instead of one.  A full table holds the same entries, so a wavefront that mixes both kinds of key
// takes this path for all its lanes.  Break-even against the one-lane doubling kernel: ~4 signatures per key (a table without its
// fill costs ~3 generic verifications, a narrow verification 72 additions instead of 256 doublings + 64 additions); against the
// full table: ~250 signatures per key (the fill's ~14 M instructions buy 26 fewer additions per signature).
// The pass reads a key's COMPACT rows (p256_keytab29.h: 16 entries per window — babies at 0..7, giant 16 a at 7 + a): gi / bi index them.
SBV_HD void narrow_split(int idx, bool skip, int& gi, int& bi, bool& bneg) {
    const int ad = skip ? 0 : idx + 1;
    const int a = (ad + 7) >> 4;
    const int b = ad - 16 * a;
    gi = a ? 7 + a : -1;
    bi = b ? (b < 0 ? -b : b) - 1 : -1;
    bneg = b < 0;
}
#define SBV_NARROW_PER_WINDOW 16
SBV_HD void qphase29_point_narrow(xyzz& R, const u256& u2in, const apt* qtab, int j0, int j1) {
    const bool flip = (u2in.v[7] >> 31) != 0;
    u256 u2, nmu;
    (void)sub256(nmu, sc_n(), u2in);
    select256(u2, flip, nmu, u2in);
    u256 k2;
    const u32 top2 = add_const_limbs(k2, u2, 0x80808080u);
    if (j1 == SBV_GTAB_WINDOWS && !wave_any(top2 != 0)) j1 = SBV_GTAB_WINDOWS - 1;
    if (j0 >= j1) return;
    int idx, gi, bi; bool neg, skip, bneg;
    comb_digit(k2, top2, j0, idx, neg, skip);
    narrow_split(idx, skip, gi, bi, bneg);
    raw_apt cg, cb;                                   // the current window's giant and baby (entry 0 stands in for an absent part: never added)
    raw_apt_load(cg, qtab + (size_t)j0 * SBV_NARROW_PER_WINDOW + (gi < 0 ? 0 : gi));
    raw_apt_load(cb, qtab + (size_t)j0 * SBV_NARROW_PER_WINDOW + (bi < 0 ? 0 : bi));
    SBV_NOUNROLL
    for (int j = j0; j < j1; ++j) {
        const int jn = j + 1 < j1 ? j + 1 : j1 - 1;
        int idxn, gin, bin; bool negn, skipn, bnegn;
        comb_digit(k2, top2, jn, idxn, negn, skipn);
        narrow_split(idxn, skipn, gin, bin, bnegn);
        raw_apt ng;                                   // the next giant is fetched one addition ahead, the next baby after the giant's addition
        raw_apt_load(ng, qtab + (size_t)jn * SBV_NARROW_PER_WINDOW + (gin < 0 ? 0 : gin));
        if (gi >= 0) {
            apt29 q;
            raw_apt_unpack(q, cg);
            pt29_madd(R, q, neg != flip);
        }
        raw_apt nb;
        raw_apt_load(nb, qtab + (size_t)jn * SBV_NARROW_PER_WINDOW + (bin < 0 ? 0 : bin));
        if (bi >= 0) {
            apt29 q;
            raw_apt_unpack(q, cb);
            pt29_madd(R, q, (neg != flip) != bneg);
        }
        cg = ng; cb = nb; neg = negn; gi = gin; bi = bin; bneg = bnegn;
    }
}

// Q phase of the grouped step.  `last` -> the verdict is returned; otherwise R goes back to gacc for the next chunk
// of windows and the return value is meaningless.
// NARROW: ktab = the pool of COMPACT rows (SBV_GTAB_WINDOWS x SBV_NARROW_PER_WINDOW entries per slot) instead of the pool of full tables
template <bool NARROW = false>
SBV_HD bool qphase29_lane(const Scratch& s, size_t i, u32 slot, u32 nkeys, const apt* ktab, const uint8_t* kvalid,
                          u32* gacc, int j0, int j1, bool last) {
    u256 u2;
    soa_load(u2, s.u2, s.cap, i);
    bool ok = s.ok[i] != 0 && slot < nkeys;
    if (slot >= nkeys) slot = 0;
    ok = ok && kvalid[slot] != 0;
    const apt* qtab = ktab + (size_t)slot * (SBV_GTAB_WINDOWS * (NARROW ? SBV_NARROW_PER_WINDOW : SBV_GTAB_PER_WINDOW));
    xyzz R;
    gacc29_load(R, gacc, s.cap, i);
    if (NARROW) qphase29_point_narrow(R, u2, qtab, j0, j1);
    else qphase29_point(R, u2, qtab, j0, j1);
    if (!last) { gacc29_store(gacc, s.cap, i, R); return false; }
    u256 r;
    soa_load(r, s.r, s.cap, i);
    return ok && pt29_rx_matches(R, r);
}

template <bool NARROW = false>
SBV_HD bool qphase29_lane_sorted(const Scratch& s, size_t t, size_t L, u32 slot, u32 nkeys, const apt* ktab, const uint8_t* kvalid,
                                 u32* gacc, int j0, int j1, bool last) {
    u256 u2;
    rec_load256(u2, s.rec, t, SBV_REC_U2);
    bool ok = slot < nkeys;
    if (slot >= nkeys) slot = 0;
    ok = ok && kvalid[slot] != 0;
    const apt* qtab = ktab + (size_t)slot * (SBV_GTAB_WINDOWS * (NARROW ? SBV_NARROW_PER_WINDOW : SBV_GTAB_PER_WINDOW));
    xyzz R;
    gacc29_load(R, gacc, s.cap, L);
    if (NARROW) qphase29_point_narrow(R, u2, qtab, j0, j1);
    else qphase29_point(R, u2, qtab, j0, j1);
    if (!last) {
        // The accumulator goes back where it came from.  Left alone, the compiler keeps the 36 load addresses (72 VGPRs: the planes are
        // `cap` words apart, no immediate offset reaches) alive across the whole comb loop for these stores — the difference between 168
        // VGPRs and 66 spilled dwords at 3 waves per SIMD.  An opaque copy of the position makes it compute them again (a hundred
        // instructions once per launch).
        size_t Ls = L;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(Ls));
#endif
        gacc29_store(gacc, s.cap, Ls, R);
        return false;
    }
    u256 r;
    rec_load256(r, s.rec, t, SBV_REC_R);
    ok = ok && s.rec[t * SBV_REC_WORDS + SBV_REC_OK] != 0;
    return ok && pt29_rx_matches(R, r);
}

// registered key: u1 * G + u2 * Q in one pass
SBV_HD bool verify29_lane_keyed(const Scratch& s, size_t i, u32 slot, u32 nkeys, const apt* ktab, const uint8_t* kvalid,
                                const gcomb& gc, const widekeys& wk) {
    u256 r, u1, u2;
    soa_load(r, s.r, s.cap, i);
    soa_load(u1, s.u1, s.cap, i);
    soa_load(u2, s.u2, s.cap, i);
    bool ok = s.ok[i] != 0 && slot < nkeys;
    if (slot >= nkeys) slot = 0;
    ok = ok && kvalid[slot] != 0;
    const apt* qtab = ktab + (size_t)slot * (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW);
    xyzz R;
    gphase29_point(R, u1, gc);
    const u32 widx = widekeys_index(wk, slot);
    if (wave_all(widx != SBV_WIDE_NONE)) wide_qphase29_point(R, u2, wk, widx);
    else qphase29_point(R, u2, qtab, 0, SBV_GTAB_WINDOWS);
    return ok && pt29_rx_matches(R, r);
}

// ---- generic form: the public key travels in the tuple and is seen once (no table worth building per batch) ---------------
// u2 * Q with 64 signed 4-bit windows over a per-signature table of the AFFINE multiples 1..8 of Q (chain of 7 exact mixed
// additions, one inversion per lane to normalise them: 64-byte entries instead of 160-byte Jacobian ones, 4.6 KB of table
// traffic per signature instead of 10 KB), a Jacobian accumulator (4 doublings of 3M + 5S and one mixed addition per
// window, fused reductions), then + u1 * G from the comb of G.  qtab: this lane's 8 x 16 words.
#define SBV_QTAB29_WORDS (8 * 16 + 7 * 45 + 5)        // table + raw chain records, rounded to 448 words = 1792 bytes
// Two halves, so that the lane fetches r and u1 BEHIND the doubling loop instead of holding their 16 words across it (round 6: the
// one-lane kernel's spills): verify29_generic_q leaves u2 * Q in S, verify29_generic_finish adds u1 * G and compares.
SBV_HD void verify29_generic_q(xyzz& S, bool& ok, const u256& u2, const u256& qx, const u256& qy, u32* qtab) {
    apt29 Q;
    f29_from_plain(Q.x, qx);
    f29_from_plain(Q.y, qy);
    {   // pointFromAffine: y^2 == x^3 - 3x + b
        fe29 lhs, t, rhs;
        f29_sqr(lhs, Q.y);
        f29_sqr(t, Q.x);
        f29_mul(rhs, t, Q.x);
        f29_sub(rhs, rhs, Q.x);
        f29_sub(rhs, rhs, Q.x);
        f29_sub(rhs, rhs, Q.x);
        f29_add(rhs, rhs, f29_b());
        f29_sub(t, lhs, rhs);
        ok = ok && f29_is_zero(t);
    }
    // table k * Q, k = 1..8: XYZZ chain, parked raw behind the table in the lane's strip, normalised with one inversion
    {
        u32 w[16];
        f29_store_canon(w, Q.x); f29_store_canon(w + 8, Q.y);
        SBV_UNROLL
        for (int l = 0; l < 16; ++l) qtab[l] = w[l];
        u32* raw = qtab + 8 * 16;               // 7 records of 45 words: X, Y, ZZ, ZZZ, prefix product
        xyzz T;
        T.X = Q.x; T.Y = Q.y; T.ZZ = f29_one(); T.ZZZ = f29_one(); T.inf = false;
        fe29 acc = f29_one();
        SBV_NOUNROLL
        for (int k = 0; k < 7; ++k) {
            pt29_madd(T, Q, false);             // k = 0 is Q + Q: the doubling branch
            u32* rec = raw + k * 45;
            SBV_UNROLL
            for (int l = 0; l < 9; ++l) {
                rec[l] = (u32)T.X.v[l]; rec[9 + l] = (u32)T.Y.v[l]; rec[18 + l] = (u32)T.ZZ.v[l]; rec[27 + l] = (u32)T.ZZZ.v[l];
                rec[36 + l] = (u32)acc.v[l];
            }
            f29_mul(acc, acc, T.ZZZ);
        }
        fe29 inv;
        f29_inv(inv, acc);                      // an off-curve "point" may give 0 here: garbage entries, ok is already false
        SBV_NOUNROLL
        for (int k = 6; k >= 0; --k) {
            const u32* rec = raw + k * 45;
            fe29 X, Y, ZZ, ZZZ, pre, i3, wv, w2;
            SBV_UNROLL
            for (int l = 0; l < 9; ++l) {
                X.v[l] = (i32)rec[l]; Y.v[l] = (i32)rec[9 + l]; ZZ.v[l] = (i32)rec[18 + l]; ZZZ.v[l] = (i32)rec[27 + l];
                pre.v[l] = (i32)rec[36 + l];
            }
            f29_mul(i3, inv, pre);
            f29_mul(inv, inv, ZZZ);
            f29_mul(wv, ZZ, i3);
            f29_sqr(w2, wv);
            apt29 a;
            f29_mul(a.x, X, w2);
            f29_mul(a.y, Y, i3);
            f29_store_canon(w, a.x); f29_store_canon(w + 8, a.y);
            SBV_UNROLL
            for (int l = 0; l < 16; ++l) qtab[(k + 1) * 16 + l] = w[l];
        }
    }
    // signed-window recoding: u2 + 0x88..8 has nibbles d + 8, d in [-8, 7]; bit 256 is a final +1 digit
    u256 k2;
    const u32 top2 = add_const_limbs(k2, u2, 0x88888888u);
    jpt29f R;
    R.X = Q.x; R.Y = Q.y; R.Z = f29_one();
    R.inf = top2 == 0;
    SBV_NOUNROLL
    for (int w = 63; w >= 0; --w) {
        if (!R.inf) {
            SBV_NOUNROLL
            for (int t = 0; t < 4; ++t) pt29_dbl_jacx(R);
        }
        const int d = (int)((k2.v[w >> 3] >> ((w & 7) * 4)) & 15u) - 8;
        if (d != 0) {
            const int ad = d < 0 ? -d : d;
            apt29 e;
            apt29_load(e, qtab + (ad - 1) * 16);
            pt29_madd_jacx(R, e, d < 0);
        }
    }
    S.inf = R.inf;
    S.X = R.X; S.Y = R.Y;
    f29_sqrx(S.ZZ, R.Z);
    f29_mulx(S.ZZZ, R.Z, S.ZZ);
}
SBV_HD bool verify29_generic_finish(xyzz& S, const u256& r, const u256& u1, bool ok, const gcomb& gc) {
    gphase29_point(S, u1, gc, true);
    return ok && pt29_rx_matches(S, r);
}
SBV_HD bool verify29_generic_core(const u256& r, const u256& u1, const u256& u2, const u256& qx, const u256& qy, bool ok,
                                  u32* qtab, const gcomb& gc) {
    xyzz S;
    verify29_generic_q(S, ok, u2, qx, qy, qtab);
    return verify29_generic_finish(S, r, u1, ok, gc);
}
SBV_HD bool verify29_lane_generic(const Scratch& s, size_t i, u32* qtab, const gcomb& gc) {
    u256 u2, qx, qy;
    soa_load(u2, s.u2, s.cap, i);
    soa_load(qx, s.qx, s.cap, i);
    soa_load(qy, s.qy, s.cap, i);
    bool ok = s.ok[i] != 0;
    xyzz S;
    verify29_generic_q(S, ok, u2, qx, qy, qtab);
    u256 r, u1;
    soa_load(u1, s.u1, s.cap, i);
    soa_load(r, s.r, s.cap, i);
    return verify29_generic_finish(S, r, u1, ok, gc);
}
// The same lane inside the key-sorted grouped step: stage A left no limb-major planes (prep_chunk29), so u1 | u2 | r | ok come
// from the tuple's record and the public key from the tuple itself (stage A's range checks on it are part of `ok`).
SBV_HD bool verify29_lane_generic_rec(const Scratch& s, const uint8_t* tuples, size_t i, u32* qtab, const gcomb& gc) {
    u256 u2, qx, qy;
    rec_load256(u2, s.rec, i, SBV_REC_U2);
    const u32* k = reinterpret_cast<const u32*>(tuples + i * 160 + 96);
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) { qx.v[l] = bswap32(k[7 - l]); qy.v[l] = bswap32(k[8 + 7 - l]); }
    bool ok = s.rec[i * SBV_REC_WORDS + SBV_REC_OK] != 0;
    xyzz S;
    verify29_generic_q(S, ok, u2, qx, qy, qtab);
    u256 r, u1;
    rec_load256(u1, s.rec, i, SBV_REC_U1);
    rec_load256(r, s.rec, i, SBV_REC_R);
    return verify29_generic_finish(S, r, u1, ok, gc);
}

// ---- registered-key form, several lanes per signature (the latency form, BASELINE.json's second metric) -------------------
// The 50 comb terms of u1 * G + u2 * Q are independent, so SBV_COOP_LANES lanes each sum every SBV_COOP_LANES-th term and
// the partial sums meet in a butterfly of exact XYZZ additions (pt29_add): ~10 additions deep instead of 50.
// lanes = lanes per signature (a power of two): SBV_COOP_LANES in the throughput-sized latency kernels, SBV_SMALL_LANES in the
// one-launch form of a commit quorum.
// wide: the key's wide comb kwv (widekeys above; the caller decided wave-uniformly) replaces the 33 windows of qtab.
SBV_HD void keyed29_partial_lane(xyzz& R, const u256& u1, const u256& u2in, const apt* qtab, const gcomb& gc, int sub, int lanes = SBV_COOP_LANES,
                                 bool wide = false, gcomb kwv = gcomb_make(nullptr, 16)) {
    u288 k1;
    gcomb_recode(k1, u1, gc.bits, gc.windows);
    // narrow comb: u2 + 0x80..80 (carry window 32); wide comb: the sign trick of wide_qphase29_point, then the comb-of-G recoding
    const bool flip = wide && (u2in.v[7] >> 31) != 0;
    u256 u2, nmu;
    (void)sub256(nmu, sc_n(), u2in);
    select256(u2, flip, nmu, u2in);
    u256 k2;
    const u32 top2 = add_const_limbs(k2, u2, 0x80808080u);
    u288 k2w;
    gcomb_recode(k2w, u2, kwv.bits, kwv.windows);
    pt29_set_inf(R);
    const int kSteps = gc.windows + (wide ? kwv.windows : SBV_GTAB_WINDOWS);
    auto locate = [&](int t, bool& neg, bool& skip) -> const apt* {
        if (t < gc.windows) {
            u32 idx;
            gcomb_digit(k1, gc.bits, t, idx, neg, skip);
            return gc.tab + ((size_t)t << (gc.bits - 1)) + idx;
        }
        if (wide) {
            u32 idx;
            gcomb_digit(k2w, kwv.bits, t - gc.windows, idx, neg, skip);
            neg = neg != flip;
            return kwv.tab + ((size_t)(t - gc.windows) << (kwv.bits - 1)) + idx;
        }
        int idx;
        comb_digit(k2, top2, t - gc.windows, idx, neg, skip);
        return qtab + (size_t)(t - gc.windows) * SBV_GTAB_PER_WINDOW + idx;
    };
    bool neg, skip;
    raw_apt cur;
    raw_apt_load(cur, locate(sub, neg, skip));
    SBV_NOUNROLL
    for (int t = sub; t < kSteps; t += lanes) {
        const int tn = t + lanes < kSteps ? t + lanes : t;
        bool negn, skipn;
        raw_apt nxt;
        raw_apt_load(nxt, locate(tn, negn, skipn));
        if (!skip) {
            apt29 q;
            raw_apt_unpack(q, cur);
            pt29_madd(R, q, neg);
        }
        cur = nxt; neg = negn; skip = skipn;
    }
}

// ---- table conversion: entries of the 8 x 32 Montgomery domain (R = 2^256, p256_core.h generators) -> R = 2^261 ----
SBV_HD void apt_to_r261(apt& out, const apt& in) {
    fe29 x, y;
    f29_from_fe(x, in.x);
    f29_from_fe(y, in.y);
    f29_store_canon(out.x.v, x);
    f29_store_canon(out.y.v, y);
}
// the same on 8 x 32 arithmetic only (used inside the 8 x 32 table-building kernels): x * 2^5 mod p
SBV_HD void fe_mul32(fe& r, const fe& a) {
    fe t = a;
    SBV_UNROLL
    for (int i = 0; i < 5; ++i) fe_dbl(t, t);
    r = t;
}

}  // namespace sbv
