// p256_core.h — the two per-lane stages of batch ECDSA P-256 verification.
//
// Semantics = Go >= 1.20 crypto/ecdsa.verifyNISTEC on one 160-byte tuple r|s|hash|Qx|Qy
// (5 x 32 B big-endian): what a stock implementation of the reference's api.Verifier
// (pkg/api/dependencies.go:54-71: VerifyRequest / VerifySignature / VerifyConsenterSig, and
// the K request signatures inside VerifyProposal, internal/bft/view.go:555) computes per
// signature after DER parsing and hashing.
//
//   stage A  prep_chunk   : range checks (bigmod SetBytes / IsZero, pointFromAffine's
//                           coordinate < p), e = hash mod N (hashToNat), s^-1 by Montgomery's
//                           trick over the thread's chunk of T tuples (one inversion by division
//                           steps, modinv30.h, per T signatures), u1 = e*s^-1, u2 = r*s^-1.  Writes a limb-major
//                           (SoA) scratch so stage B's loads are coalesced.
//   stage B  verify_lane  : on-curve check, R = u1*G + u2*Q with signed fixed windows
//                           (4-bit for Q from a per-signature table, 8-bit comb for G from a
//                           precomputed table), R != infinity, R.x == r (mod N) checked
//                           projectively: X == r*Z^2 or (r + N < p and X == (r+N)*Z^2).
//
// The same source is compiled by hipcc for gfx950 (kernels in p256_kernels.hip) and by g++
// for tests/emul (CPU test tier: lane-by-lane emulation diffed against the oracle).
#pragma once
#include "p256_fe.h"
#include "p256_pt.h"
#include "p256_sc.h"
#include "p256_sc29.h"

namespace sbv {

// ---- scratch between stage A and stage B ---------------------------------------------------------
// Limb-major arrays: word (l, i) of field F lives at F[l * cap + i] so that lane i of a
// wavefront reads consecutive dwords.  cap = n rounded up to the launch granularity.
struct Scratch {
    u32* r;      // signature r (plain integer)
    u32* u1;     // e * s^-1 mod N (plain); also temp: exclusive prefix products during stage A
    u32* u2;     // r * s^-1 mod N (plain); also temp: e during stage A
    u32* qx;     // public key x (plain)
    u32* qy;     // public key y (plain)
    u32* sm;     // temp: s in Montgomery form (stage A only)
    uint8_t* ok; // 1 = passed the range checks
    size_t cap;
    u32* rec = nullptr;   // optional: one 128-byte record per tuple (rec_* below), written by stage A beside the limb-major planes
};

// Tuple-major twin of (u1, u2, r, ok): SBV_REC_WORDS dwords per tuple = u1[8] | u2[8] | r[8] | ok | pad.  The key-sorted
// grouped step (p256_group.h) visits tuples in key order, i.e. in random tuple order: from the limb-major planes one
// u256 would touch 8 cache lines, from its record one.
#define SBV_REC_WORDS 32
#define SBV_REC_U1 0
#define SBV_REC_U2 8
#define SBV_REC_R 16
#define SBV_REC_OK 24
struct alignas(16) rec_q4 { u32 x, y, z, w; };
SBV_HD void rec_store256(u32* rec, size_t i, int off, const u32 v[8]) {
    rec_q4* d = reinterpret_cast<rec_q4*>(rec + i * SBV_REC_WORDS + off);
    const rec_q4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    d[0] = lo;
    d[1] = hi;
}

SBV_HD void soa_store(u32* base, size_t cap, size_t i, const u256& v) {
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) base[(size_t)l * cap + i] = v.v[l];
}
SBV_HD void soa_load(u256& v, const u32* base, size_t cap, size_t i) {
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) v.v[l] = base[(size_t)l * cap + i];
}
SBV_HD void rec_load256(u256& v, const u32* rec, size_t i, int off) {
    const rec_q4* s = reinterpret_cast<const rec_q4*>(rec + i * SBV_REC_WORDS + off);
    const rec_q4 lo = s[0], hi = s[1];
    v.v[0] = lo.x; v.v[1] = lo.y; v.v[2] = lo.z; v.v[3] = lo.w; v.v[4] = hi.x; v.v[5] = hi.y; v.v[6] = hi.z; v.v[7] = hi.w;
}

// field f (0..4 = r, s, hash, Qx, Qy) of a tuple given as 40 packed big-endian dwords
// (`w` points at the tuple's first dword, `stride` is the distance between dwords in u32 units).
template <typename WordPtr>
SBV_HD void tuple_field(u256& out, WordPtr w, int f) {
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) out.v[l] = bswap32(w[f * 8 + (7 - l)]);
}

// ---- stage A ----------------------------------------------------------------------------------------
// One thread handles tuples idx_k = first + k * step, k = 0..T-1 (those with idx_k < n).
// `TupleWords` is a callable (k, idx) -> indexable giving the 40 big-endian dwords of tuple
// idx; it is invoked by every thread for every k (it may contain workgroup barriers: the
// kernel stages each 64-tuple slab through LDS with coalesced 16-byte loads).
// HAS_Q = false is the registered-key form: tuples are r|s|hash (96 B), the public key comes from
// a key slot validated at registration, so only r and s are range-checked here.
template <bool HAS_Q, typename TupleWords>
SBV_HD void prep_chunk(TupleWords words, size_t n, const Scratch& sc_, size_t first, size_t step, int T) {
    const sc n_ = sc_n();
    const fe p_ = fe_p();
    sc acc = sc_one_mont();
    for (int k = 0; k < T; ++k) {
        const size_t idx = first + (size_t)k * step;
        auto w = words(k, idx);
        if (idx < n) {
            u256 r, s, e, qx, qy;
            tuple_field(r, w, 0);
            tuple_field(s, w, 1);
            tuple_field(e, w, 2);
            bool ok = !is_zero256(r) && lt256(r, n_) && !is_zero256(s) && lt256(s, n_);
            if (HAS_Q) {
                tuple_field(qx, w, 3);
                tuple_field(qy, w, 4);
                ok = ok && lt256(qx, p_) && lt256(qy, p_);
            }
            // hashToNat: e < 2^256 < 2N, one conditional subtraction
            sc_cond_sub_n(e, e, 0);
            sc sM;
            sc_to_mont(sM, s);                 // garbage if s >= N, replaced below
            const sc one = sc_one_mont();
            select256(sM, ok, sM, one);        // keep the product chain invertible
            soa_store(sc_.u1, sc_.cap, idx, acc);   // exclusive prefix product
            soa_store(sc_.sm, sc_.cap, idx, sM);
            soa_store(sc_.u2, sc_.cap, idx, e);
            soa_store(sc_.r, sc_.cap, idx, r);
            if (HAS_Q) {
                soa_store(sc_.qx, sc_.cap, idx, qx);
                soa_store(sc_.qy, sc_.cap, idx, qy);
            }
            sc_.ok[idx] = ok ? 1 : 0;
            sc_mul(acc, acc, sM);
        }
    }
    sc inv;
    sc_inv_gcd(inv, acc);                  // (prod s_k)^-1, Montgomery form; division steps, not the 350-multiplication Fermat chain
    for (int k = T - 1; k >= 0; --k) {
        const size_t idx = first + (size_t)k * step;
        if (idx >= n) continue;
        sc pre, sM, w;
        u256 e, r, u1, u2;
        soa_load(pre, sc_.u1, sc_.cap, idx);
        soa_load(sM, sc_.sm, sc_.cap, idx);
        soa_load(e, sc_.u2, sc_.cap, idx);
        soa_load(r, sc_.r, sc_.cap, idx);
        sc_mul(w, inv, pre);               // s_k^-1 (Montgomery)
        sc_mul(inv, inv, sM);              // drop s_k from the running inverse
        sc_mul(u1, w, e);                  // Montgomery(w) * plain(e) = plain(e * w)
        sc_mul(u2, w, r);
        soa_store(sc_.u1, sc_.cap, idx, u1);
        soa_store(sc_.u2, sc_.cap, idx, u2);
    }
}

// The same stage A on the carry-free representation (p256_sc29.h): identical results (u1, u2, r, ok, keys in the
// scratch), identical thread -> tuple mapping and Montgomery's trick along the thread's chunk, but every product is
// 81 independent multiply-accumulates instead of a CIOS loop of dependent carry chains.  Prefix products and s*R are
// parked between the two passes as canonical 256-bit words in the scratch planes stage B overwrites anyway.
template <bool HAS_Q, typename TupleWords>
SBV_HD void prep_chunk29(TupleWords words, size_t n, const Scratch& sc_, size_t first, size_t step, int T) {
    const sc n_ = sc_n();
    const fe p_ = fe_p();
    const fe29 one = s29_one();
    fe29 acc = one;
    // With per-tuple records (Scratch::rec: the key-sorted grouped step) NOTHING goes to the limb-major planes: every reader of
    // that step takes u1 | u2 | r | ok from the record and the public key from the tuple itself, and the values parked between
    // the two passes live in the record too (197 bytes written per tuple instead of 384).
    const bool rec_only = sc_.rec != nullptr;
    for (int k = 0; k < T; ++k) {
        const size_t idx = first + (size_t)k * step;
        auto w = words(k, idx);
        if (idx < n) {
            u256 r, s, e, qx, qy;
            tuple_field(r, w, 0);
            tuple_field(s, w, 1);
            tuple_field(e, w, 2);
            bool ok = !is_zero256(r) && lt256(r, n_) && !is_zero256(s) && lt256(s, n_);
            if (HAS_Q) {
                tuple_field(qx, w, 3);
                tuple_field(qy, w, 4);
                ok = ok && lt256(qx, p_) && lt256(qy, p_);
            }
            sc_cond_sub_n(e, e, 0);            // hashToNat: e < 2^256 < 2N, one conditional subtraction
            fe29 sL, sM;
            f29_unpack(sL, s.v);               // garbage if s >= N (still a bounded operand), replaced below
            s29_mul(sM, sL, s29_r2());
            f29_select(sM, ok, sM, one);       // keep the product chain invertible
            u256 tw;
            s29_store_canon(tw, acc);          // exclusive prefix product
            if (rec_only) {                    // parked in the tuple's own record: prefix | e | r | s (Montgomery)
                rec_store256(sc_.rec, idx, SBV_REC_U1, tw.v);
                s29_store_canon(tw, sM);
                rec_store256(sc_.rec, idx, SBV_REC_OK, tw.v);
                rec_store256(sc_.rec, idx, SBV_REC_U2, e.v);
                rec_store256(sc_.rec, idx, SBV_REC_R, r.v);
            } else {
                soa_store(sc_.u1, sc_.cap, idx, tw);
                s29_store_canon(tw, sM);
                soa_store(sc_.sm, sc_.cap, idx, tw);
                soa_store(sc_.u2, sc_.cap, idx, e);
                soa_store(sc_.r, sc_.cap, idx, r);
                if (HAS_Q) {
                    soa_store(sc_.qx, sc_.cap, idx, qx);
                    soa_store(sc_.qy, sc_.cap, idx, qy);
                }
            }
            sc_.ok[idx] = ok ? 1 : 0;
            s29_mul(acc, acc, sM);
        }
    }
    fe29 inv;
    s29_inv(inv, acc);                          // (prod s_k)^-1, Montgomery form
    for (int k = T - 1; k >= 0; --k) {
        const size_t idx = first + (size_t)k * step;
        if (idx >= n) continue;
        u256 tw, e, r;
        fe29 pre, sM, w, eL, rL, u;
        if (rec_only) {
            rec_load256(tw, sc_.rec, idx, SBV_REC_U1);
            f29_unpack(pre, tw.v);
            rec_load256(tw, sc_.rec, idx, SBV_REC_OK);
            f29_unpack(sM, tw.v);
            rec_load256(e, sc_.rec, idx, SBV_REC_U2);
            rec_load256(r, sc_.rec, idx, SBV_REC_R);
        } else {
            soa_load(tw, sc_.u1, sc_.cap, idx);
            f29_unpack(pre, tw.v);
            soa_load(tw, sc_.sm, sc_.cap, idx);
            f29_unpack(sM, tw.v);
            soa_load(e, sc_.u2, sc_.cap, idx);
            soa_load(r, sc_.r, sc_.cap, idx);
        }
        f29_unpack(eL, e.v);
        f29_unpack(rL, r.v);
        s29_mul(w, inv, pre);                  // s_k^-1 (Montgomery)
        s29_mul(inv, inv, sM);                 // drop s_k from the running inverse
        s29_mul(u, w, eL);                     // Montgomery(w) * plain(e) = plain(e * w)
        s29_store_canon(tw, u);
        if (rec_only) rec_store256(sc_.rec, idx, SBV_REC_U1, tw.v);
        else soa_store(sc_.u1, sc_.cap, idx, tw);
        s29_mul(u, w, rL);
        s29_store_canon(tw, u);
        if (rec_only) {                        // r is already in its place
            rec_store256(sc_.rec, idx, SBV_REC_U2, tw.v);
            sc_.rec[idx * SBV_REC_WORDS + SBV_REC_OK] = sc_.ok[idx];
        } else {
            soa_store(sc_.u2, sc_.cap, idx, tw);
        }
    }
}

// ---- stage B ----------------------------------------------------------------------------------------
#define SBV_QTAB_ENTRIES 8
#define SBV_GTAB_WINDOWS 33
#define SBV_GTAB_PER_WINDOW 128

#define SBV_G16_WINDOWS 17
#define SBV_G16_PER_WINDOW 32768
SBV_HD void comb16_digit(const u256& k, u32 top, int j, int& idx, bool& neg, bool& skip) {
    if (j == 16) { idx = 0; neg = false; skip = top == 0; return; }
    const int d = (int)((k.v[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu) - 32768;
    const int ad = d < 0 ? -d : d;
    idx = ad == 0 ? 0 : ad - 1;
    neg = d < 0;
    skip = d == 0;
}
struct alignas(16) vec4 { u32 x, y, z, w; };

SBV_HD void fe_store16(u32* dst, const fe& a) {
    vec4* d = reinterpret_cast<vec4*>(dst);
    vec4 lo = {a.v[0], a.v[1], a.v[2], a.v[3]}, hi = {a.v[4], a.v[5], a.v[6], a.v[7]};
    d[0] = lo;
    d[1] = hi;
}
SBV_HD void fe_load16(fe& a, const u32* src) {
    const vec4* s = reinterpret_cast<const vec4*>(src);
    const vec4 lo = s[0], hi = s[1];
    a.v[0] = lo.x; a.v[1] = lo.y; a.v[2] = lo.z; a.v[3] = lo.w;
    a.v[4] = hi.x; a.v[5] = hi.y; a.v[6] = hi.z; a.v[7] = hi.w;
}
template <bool FAST = false>
SBV_HD void qent_store(u32* dst, const jpt& p, u32* st = nullptr) {
    fe zz, zzz;
    fe_sqr<FAST>(zz, p.Z, st);
    fe_mul<FAST>(zzz, zz, p.Z, st);
    fe_store16(dst, p.X);
    fe_store16(dst + 8, p.Y);
    fe_store16(dst + 16, p.Z);
    fe_store16(dst + 24, zz);
    fe_store16(dst + 32, zzz);
}
SBV_HD void qent_load(qent& q, const u32* src) {
    fe_load16(q.X, src);
    fe_load16(q.Y, src + 8);
    fe_load16(q.Z, src + 16);
    fe_load16(q.ZZ, src + 24);
    fe_load16(q.ZZZ, src + 32);
}

// v + c as a 257-bit value: returns the low 256 bits, `top` = bit 256
SBV_HD u32 add_const_limbs(u256& out, const u256& v, u32 c_limb) {
    u32 c = 0;
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) out.v[l] = addc(v.v[l], c_limb, c);
    return c;
}

// R.x mod N == r  <=>  R != infinity and (X == r Z^2  or  (r + N < p and X == (r + N) Z^2))  (mod p),
// with r < N; no inversion.
template <bool FAST = false>
SBV_HD bool rx_matches(const jpt& R, const u256& r, u32* st = nullptr) {
    if (pt_is_inf(R)) return false;
    fe zz, rM, t;
    fe_sqr<FAST>(zz, R.Z, st);
    fe_to_mont<FAST>(rM, r, st);
    fe_mul<FAST>(t, rM, zz, st);
    bool match = fe_eq(t, R.X);
    const sc n_ = sc_n();
    const fe p_ = fe_p();
    u256 rn;
    const u32 carry = add256(rn, r, n_);
    const bool wrap_possible = (carry == 0) && lt256(rn, p_);
    if (wrap_possible) {                    // only for r < p - N ~ 2^128: essentially never
        fe_to_mont<FAST>(rM, rn, st);
        fe_mul<FAST>(t, rM, zz, st);
        match = match || fe_eq(t, R.X);
    }
    return match;
}

// ---- G phase (in-step grouping) ----------------------------------------------------------------------
// u1*G does not depend on the public key, so the grouped step computes it for every tuple while the
// per-batch key tables are still being built.  The Jacobian result is parked limb-major like Scratch:
// word w (0..23 = X, Y, Z limbs) of tuple i at gacc[w * cap + i].
SBV_HD void gacc_store(u32* gacc, size_t cap, size_t i, const jpt& R) {
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) {
        gacc[(size_t)l * cap + i] = R.X.v[l];
        gacc[(size_t)(8 + l) * cap + i] = R.Y.v[l];
        gacc[(size_t)(16 + l) * cap + i] = R.Z.v[l];
    }
}
SBV_HD void gacc_load(jpt& R, const u32* gacc, size_t cap, size_t i) {
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) {
        R.X.v[l] = gacc[(size_t)l * cap + i];
        R.Y.v[l] = gacc[(size_t)(8 + l) * cap + i];
        R.Z.v[l] = gacc[(size_t)(16 + l) * cap + i];
    }
}
SBV_HD void gphase_lane(const Scratch& s, size_t i, const apt* g16, u32* gacc) {
    u256 u1, k1;
    soa_load(u1, s.u1, s.cap, i);
    const u32 top1 = add_const_limbs(k1, u1, 0x80008000u);
    jpt R;
    pt_set_inf(R);
    apt cur;
    int idx; bool neg, skip;
    comb16_digit(k1, top1, 0, idx, neg, skip);
    {
        const u32* gp = reinterpret_cast<const u32*>(g16 + idx);
        fe_load16(cur.x, gp); fe_load16(cur.y, gp + 8);
    }
    SBV_NOUNROLL
    for (int j = 0; j < SBV_G16_WINDOWS; ++j) {
        const int jn = j + 1 < SBV_G16_WINDOWS ? j + 1 : SBV_G16_WINDOWS - 1;
        int idxn; bool negn, skipn;
        comb16_digit(k1, top1, jn, idxn, negn, skipn);
        const u32* gp = reinterpret_cast<const u32*>(g16 + (size_t)jn * SBV_G16_PER_WINDOW + idxn);
        apt nxt;
        fe_load16(nxt.x, gp); fe_load16(nxt.y, gp + 8);
        pt_add_mixed(R, cur, neg, skip);
        cur = nxt; neg = negn; skip = skipn;
    }
    gacc_store(gacc, s.cap, i, R);
}

// Returns accept (true) / reject for lane `i`.  `qtab` = this lane's private table space
// (SBV_QTAB_ENTRIES * 40 dwords, 16-byte aligned), `g16` = 17 x 32768 affine multiples of G:
// g16[j * 32768 + (k-1)] = k * 2^(16j) * G.
// FAST = true is the first pass (see fe_cond_sub_p_t): *st must start at 0 and the caller re-runs
// FAST = false for lanes whose sticky word came back 0xFFFFFFFF.
template <bool FAST = false>
SBV_HD bool verify_lane(const Scratch& s, size_t i, u32* qtab, const apt* g16, u32* st = nullptr) {
    u256 r, u1, u2, qx, qy;
    soa_load(r, s.r, s.cap, i);
    soa_load(u1, s.u1, s.cap, i);
    soa_load(u2, s.u2, s.cap, i);
    soa_load(qx, s.qx, s.cap, i);
    soa_load(qy, s.qy, s.cap, i);
    bool ok = s.ok[i] != 0;

    apt Q;
    fe_to_mont<FAST>(Q.x, qx, st);
    fe_to_mont<FAST>(Q.y, qy, st);
    ok = ok && pt_on_curve<FAST>(Q.x, Q.y, st);

    // per-signature table: k*Q for k = 1..8, Jacobian with cached Z^2, Z^3
    {
        jpt t;
        t.X = Q.x; t.Y = Q.y; t.Z = fe_one();
        qent_store<FAST>(qtab, t, st);
        pt_dbl<FAST>(t, t, st);
        qent_store<FAST>(qtab + 40, t, st);
        for (int k = 3; k <= SBV_QTAB_ENTRIES; ++k) {
            pt_add_mixed<FAST>(t, Q, false, false, st);
            qent_store<FAST>(qtab + (k - 1) * 40, t, st);
        }
    }

    // signed-window recoding: u + 0x88..8 has nibbles d+8, d in [-8,7]; bit 256 is a final +1 digit
    u256 k2, k1;
    const u32 top2 = add_const_limbs(k2, u2, 0x88888888u);
    const u32 top1 = add_const_limbs(k1, u1, 0x80008000u);

    jpt R;
    {
        // digit 64 of u2 (0 or 1)
        const fe one = fe_one();
        const bool t = top2 != 0;
        SBV_UNROLL
        for (int l = 0; l < 8; ++l) {
            R.X.v[l] = t ? Q.x.v[l] : 0u;
            R.Y.v[l] = t ? Q.y.v[l] : 0u;
            R.Z.v[l] = t ? one.v[l] : 0u;
        }
    }
    for (int w = 63; w >= 0; --w) {
        // keep ONE copy of the doubling in the instruction stream: dbl (13 KB) + add (21 KB) must
        // stay inside the 64 KB instruction cache two CUs share; 4 inlined copies did not.
        SBV_NOUNROLL
        for (int t = 0; t < 4; ++t) pt_dbl<FAST>(R, R, st);
        const int d = (int)((k2.v[w >> 3] >> ((w & 7) * 4)) & 15u) - 8;
        const int ad = d < 0 ? -d : d;
        const int idx = ad == 0 ? 0 : ad - 1;
        qent e;
        qent_load(e, qtab + idx * 40);
        pt_add_qent<FAST>(R, e, d < 0, d == 0, st);
    }
    // fixed-base part: 16 signed 16-bit comb windows + the carry window (17 mixed additions), with the
    // next window's entry prefetched while the current addition runs
    {
        apt cur;
        int idx; bool neg, skip;
        comb16_digit(k1, top1, 0, idx, neg, skip);
        {
            const u32* gp = reinterpret_cast<const u32*>(g16 + idx);
            fe_load16(cur.x, gp); fe_load16(cur.y, gp + 8);
        }
        SBV_NOUNROLL
        for (int j = 0; j < SBV_G16_WINDOWS; ++j) {
            const int jn = j + 1 < SBV_G16_WINDOWS ? j + 1 : SBV_G16_WINDOWS - 1;
            int idxn; bool negn, skipn;
            comb16_digit(k1, top1, jn, idxn, negn, skipn);
            const u32* gp = reinterpret_cast<const u32*>(g16 + (size_t)jn * SBV_G16_PER_WINDOW + idxn);
            apt nxt;
            fe_load16(nxt.x, gp); fe_load16(nxt.y, gp + 8);
            pt_add_mixed<FAST>(R, cur, neg, skip, st);
            cur = nxt; neg = negn; skip = skipn;
        }
    }

    return ok && rx_matches<FAST>(R, r, st);
}

// ---- stage B, registered-key form -------------------------------------------------------------------
// The public key was registered once (sbv_p256_register_keys): ktab holds, per key slot, the same
// 33 x 128 comb as gtab but for Q.  R = u1*G + u2*Q is then 66 mixed additions and NO doublings
// (~4.7x fewer field multiplications than the generic form).  Consenter keys are a fixed registry in
// SmartBFT (Signature.ID selects the key, pkg/types/types.go:25-29), so this is the shape of
// VerifyConsenterSig / decision replay (BASELINE.json config 4).
SBV_HD void comb_digit(const u256& k, u32 top, int j, int& idx, bool& neg, bool& skip) {
    if (j == 32) { idx = 0; neg = false; skip = top == 0; return; }
    const int d = (int)((k.v[j >> 2] >> ((j & 3) * 8)) & 255u) - 128;
    const int ad = d < 0 ? -d : d;
    idx = ad == 0 ? 0 : ad - 1;
    neg = d < 0;
    skip = d == 0;
}

template <bool FAST = false>
SBV_HD bool verify_lane_keyed(const Scratch& s, size_t i, u32 slot, u32 nkeys, const apt* ktab,
                              const uint8_t* kvalid, const apt* g16, u32* st = nullptr) {
    u256 r, u1, u2;
    soa_load(r, s.r, s.cap, i);
    soa_load(u1, s.u1, s.cap, i);
    soa_load(u2, s.u2, s.cap, i);
    bool ok = s.ok[i] != 0 && slot < nkeys;
    if (slot >= nkeys) slot = 0;
    ok = ok && kvalid[slot] != 0;
    const apt* qtab = ktab + (size_t)slot * (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW);
    u256 k1, k2;
    const u32 top1 = add_const_limbs(k1, u1, 0x80008000u);
    const u32 top2 = add_const_limbs(k2, u2, 0x80808080u);
    jpt R;
    pt_set_inf(R);
    // one rolled loop over 17 (G, 16-bit comb) + 33 (Q, 8-bit comb) steps: a single mixed-add body in
    // the instruction stream; the next step's entry is prefetched while this step's addition runs
    constexpr int kSteps = SBV_G16_WINDOWS + SBV_GTAB_WINDOWS;
    auto locate = [&](int t, int& idx, bool& neg, bool& skip) -> const apt* {
        if (t < SBV_G16_WINDOWS) {
            comb16_digit(k1, top1, t, idx, neg, skip);
            return g16 + (size_t)t * SBV_G16_PER_WINDOW + idx;
        }
        comb_digit(k2, top2, t - SBV_G16_WINDOWS, idx, neg, skip);
        return qtab + (size_t)(t - SBV_G16_WINDOWS) * SBV_GTAB_PER_WINDOW + idx;
    };
    apt cur;
    int idx; bool neg, skip;
    {
        const u32* gp = reinterpret_cast<const u32*>(locate(0, idx, neg, skip));
        fe_load16(cur.x, gp); fe_load16(cur.y, gp + 8);
    }
    SBV_NOUNROLL
    for (int t = 0; t < kSteps; ++t) {
        const int tn = t + 1 < kSteps ? t + 1 : kSteps - 1;
        int idxn; bool negn, skipn;
        const u32* gp = reinterpret_cast<const u32*>(locate(tn, idxn, negn, skipn));
        apt nxt;
        fe_load16(nxt.x, gp); fe_load16(nxt.y, gp + 8);
        pt_add_mixed<FAST>(R, cur, neg, skip, st);
        cur = nxt; neg = negn; skip = skipn;
    }
    return ok && rx_matches<FAST>(R, r, st);
}

// ---- registered-key form, several lanes per signature (latency form) ----------------------------------
// A quorum-sized micro-batch (BASELINE.json's second metric: N = 16 -> 15 concurrent VerifyConsenterSig) puts
// one wavefront on a 256-CU GPU, and a lane's chain of 50 dependent additions is what the caller waits for.
// With no doublings in the registered-key form the 50 comb terms are independent, so SBV_COOP_LANES lanes
// each sum every SBV_COOP_LANES-th term (7 or 6 additions) and the partial sums are combined by a butterfly
// of exact Jacobian additions across the lanes (3 levels): ~10 additions deep instead of 50.
#define SBV_COOP_LANES 8
SBV_HD void pt_add_jac(jpt& R, const jpt& Q) {           // exact, either operand may be the point at infinity
    qent e;
    e.X = Q.X; e.Y = Q.Y; e.Z = Q.Z;
    fe_sqr(e.ZZ, Q.Z);
    fe_mul(e.ZZZ, e.ZZ, Q.Z);
    pt_add_qent(R, e, false, pt_is_inf(Q));
}
// R = sum of the comb terms t = sub, sub + LANES, ... < 50 of u1*G + u2*Q (key comb `qtab` of the tuple's slot)
SBV_HD void keyed_partial_lane(jpt& R, const u256& u1, const u256& u2, const apt* qtab, const apt* g16, int sub) {
    u256 k1, k2;
    const u32 top1 = add_const_limbs(k1, u1, 0x80008000u);
    const u32 top2 = add_const_limbs(k2, u2, 0x80808080u);
    pt_set_inf(R);
    constexpr int kSteps = SBV_G16_WINDOWS + SBV_GTAB_WINDOWS;
    auto locate = [&](int t, int& idx, bool& neg, bool& skip) -> const apt* {
        if (t < SBV_G16_WINDOWS) {
            comb16_digit(k1, top1, t, idx, neg, skip);
            return g16 + (size_t)t * SBV_G16_PER_WINDOW + idx;
        }
        comb_digit(k2, top2, t - SBV_G16_WINDOWS, idx, neg, skip);
        return qtab + (size_t)(t - SBV_G16_WINDOWS) * SBV_GTAB_PER_WINDOW + idx;
    };
    apt cur;
    int idx; bool neg, skip;
    {
        const u32* gp = reinterpret_cast<const u32*>(locate(sub, idx, neg, skip));
        fe_load16(cur.x, gp); fe_load16(cur.y, gp + 8);
    }
    SBV_NOUNROLL
    for (int t = sub; t < kSteps; t += SBV_COOP_LANES) {
        const int tn = t + SBV_COOP_LANES < kSteps ? t + SBV_COOP_LANES : t;
        int idxn; bool negn, skipn;
        const u32* gp = reinterpret_cast<const u32*>(locate(tn, idxn, negn, skipn));
        apt nxt;
        fe_load16(nxt.x, gp); fe_load16(nxt.y, gp + 8);
        pt_add_mixed(R, cur, neg, skip);
        cur = nxt; neg = negn; skip = skipn;
    }
}

// Q phase of the grouped step: R (from gacc) += sum of key-comb windows [j0, j1) of u2*Q.  `last` -> the
// verdict is returned; otherwise R goes back to gacc for the next chunk of windows and the return value
// is meaningless.
SBV_HD bool verify_lane_keyed_q(const Scratch& s, size_t i, u32 slot, u32 nkeys, const apt* ktab, const uint8_t* kvalid,
                                u32* gacc, int j0, int j1, bool last) {
    u256 u2, k2;
    soa_load(u2, s.u2, s.cap, i);
    bool ok = s.ok[i] != 0 && slot < nkeys;
    if (slot >= nkeys) slot = 0;
    ok = ok && kvalid[slot] != 0;
    const apt* qtab = ktab + (size_t)slot * (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW);
    const u32 top2 = add_const_limbs(k2, u2, 0x80808080u);
    jpt R;
    gacc_load(R, gacc, s.cap, i);
    apt cur;
    int idx; bool neg, skip;
    comb_digit(k2, top2, j0, idx, neg, skip);
    {
        const u32* gp = reinterpret_cast<const u32*>(qtab + (size_t)j0 * SBV_GTAB_PER_WINDOW + idx);
        fe_load16(cur.x, gp); fe_load16(cur.y, gp + 8);
    }
    SBV_NOUNROLL
    for (int j = j0; j < j1; ++j) {
        const int jn = j + 1 < j1 ? j + 1 : j1 - 1;
        int idxn; bool negn, skipn;
        comb_digit(k2, top2, jn, idxn, negn, skipn);
        const u32* gp = reinterpret_cast<const u32*>(qtab + (size_t)jn * SBV_GTAB_PER_WINDOW + idxn);
        apt nxt;
        fe_load16(nxt.x, gp); fe_load16(nxt.y, gp + 8);
        pt_add_mixed(R, cur, neg, skip);
        cur = nxt; neg = negn; skip = skipn;
    }
    if (!last) { gacc_store(gacc, s.cap, i, R); return false; }
    u256 r;
    soa_load(r, s.r, s.cap, i);
    return ok && rx_matches(R, r);
}

// ---- fixed-base table generation (host, once per sbv_init; also used by tests/emul) -----------------
// out[j * 128 + (k-1)] = k * 2^(8j) * P for j = 0..32, k = 1..128 (affine, Montgomery form); P = (px, py)
// plain coordinates of a point ON the curve (callers validate first).
inline void build_comb_table(const u256& px, const u256& py, apt* out);
inline void build_gtable(apt* out) {
    const u256 gx = {{0xD898C296u, 0xF4A13945u, 0x2DEB33A0u, 0x77037D81u, 0x63A440F2u, 0xF8BCE6E5u, 0xE12C4247u, 0x6B17D1F2u}};
    const u256 gy = {{0x37BF51F5u, 0xCBB64068u, 0x6B315ECEu, 0x2BCE3357u, 0x7C0F9E16u, 0x8EE7EB4Au, 0xFE1A7F9Bu, 0x4FE342E2u}};
    build_comb_table(gx, gy, out);
}
// pointFromAffine's checks on a registered key: coordinates < p and on the curve
inline bool key_is_valid(const u256& px, const u256& py) {
    const fe p_ = fe_p();
    if (!lt256(px, p_) || !lt256(py, p_)) return false;
    fe x, y;
    fe_to_mont(x, px);
    fe_to_mont(y, py);
    return pt_on_curve(x, y);
}
// General form: `windows` windows of `per_window` = 2^(wbits-1) entries, out[j*per_window + (k-1)] =
// k * 2^(wbits*j) * P.  One window is independent of the others once its base is known.
inline void comb_window(const apt& base, int per_window, apt* out_row) {
    jpt* row = new jpt[per_window];
    fe* pre = new fe[per_window];
    jpt t;
    t.X = base.x; t.Y = base.y; t.Z = fe_one();
    row[0] = t;
    pt_dbl(t, t);
    row[1] = t;
    for (int k = 3; k <= per_window; ++k) {
        pt_add_mixed(t, base, false, false);
        row[k - 1] = t;
    }
    fe acc = fe_one();                          // batch-invert the Z's (Montgomery's trick)
    for (int k = 0; k < per_window; ++k) { pre[k] = acc; fe_mul(acc, acc, row[k].Z); }
    fe inv;
    fe_inv(inv, acc);
    for (int k = per_window - 1; k >= 0; --k) {
        fe zi, zi2, zi3;
        fe_mul(zi, inv, pre[k]);
        fe_mul(inv, inv, row[k].Z);
        fe_sqr(zi2, zi);
        fe_mul(zi3, zi2, zi);
        fe_mul(out_row[k].x, row[k].X, zi2);
        fe_mul(out_row[k].y, row[k].Y, zi3);
    }
    delete[] row;
    delete[] pre;
}
// bases[j] = 2^(wbits*j) * P, affine
inline void comb_bases(const u256& px, const u256& py, int wbits, int windows, apt* bases) {
    apt base;
    fe_to_mont(base.x, px);
    fe_to_mont(base.y, py);
    for (int j = 0; j < windows; ++j) {
        bases[j] = base;
        jpt nb;
        nb.X = base.x; nb.Y = base.y; nb.Z = fe_one();
        for (int i = 0; i < wbits; ++i) pt_dbl(nb, nb);
        fe zi, zi2, zi3;
        fe_inv(zi, nb.Z);
        fe_sqr(zi2, zi);
        fe_mul(zi3, zi2, zi);
        fe_mul(base.x, nb.X, zi2);
        fe_mul(base.y, nb.Y, zi3);
    }
}
// The two base points per window the device-side builder of a wide comb starts from (p256_widetab29.h):
// B[j] = 2^(wbits*j) * P and C[j] = 2^hb * B[j], affine.
inline void comb_bases_bc(const u256& px, const u256& py, int wbits, int hb, int windows, apt* B, apt* C) {
    apt base;
    fe_to_mont(base.x, px);
    fe_to_mont(base.y, py);
    auto to_affine = [](apt& out, const jpt& p) {
        fe zi, zi2, zi3;
        fe_inv(zi, p.Z);
        fe_sqr(zi2, zi);
        fe_mul(zi3, zi2, zi);
        fe_mul(out.x, p.X, zi2);
        fe_mul(out.y, p.Y, zi3);
    };
    for (int j = 0; j < windows; ++j) {
        B[j] = base;
        jpt nb;
        nb.X = base.x; nb.Y = base.y; nb.Z = fe_one();
        for (int i = 0; i < hb; ++i) pt_dbl(nb, nb);
        to_affine(C[j], nb);
        for (int i = hb; i < wbits; ++i) pt_dbl(nb, nb);
        to_affine(base, nb);
    }
}
inline void build_comb_table(const u256& px, const u256& py, apt* out) {
    apt bases[SBV_GTAB_WINDOWS];
    comb_bases(px, py, 8, SBV_GTAB_WINDOWS, bases);
    for (int j = 0; j < SBV_GTAB_WINDOWS; ++j) comb_window(bases[j], SBV_GTAB_PER_WINDOW, out + (size_t)j * SBV_GTAB_PER_WINDOW);
}

// ---- 16-bit comb for G (device verify kernels) ---------------------------------------------------------
// u1*G is then 17 mixed additions instead of 33.  17 x 32768 x 64 B = 35.7 MB: far beyond LDS or one
// XCD's L2 but nothing for HBM / the 256 MB Infinity Cache; the next window's entry is software-
// prefetched while the current addition runs.
// window j of the G16 table (j = 0..16), callable from several host threads
inline void build_g16_window(int j, apt* out_row) {
    const u256 gx = {{0xD898C296u, 0xF4A13945u, 0x2DEB33A0u, 0x77037D81u, 0x63A440F2u, 0xF8BCE6E5u, 0xE12C4247u, 0x6B17D1F2u}};
    const u256 gy = {{0x37BF51F5u, 0xCBB64068u, 0x6B315ECEu, 0x2BCE3357u, 0x7C0F9E16u, 0x8EE7EB4Au, 0xFE1A7F9Bu, 0x4FE342E2u}};
    apt bases[SBV_G16_WINDOWS];
    comb_bases(gx, gy, 16, j + 1, bases);
    comb_window(bases[j], SBV_G16_PER_WINDOW, out_row);
}

// window j of the `bits`-wide comb of the affine point (px, py): out_row[m-1] = m * 2^(bits j) * P, m = 1..2^(bits-1) (8 x 32
// Montgomery domain; converted for the carry-free kernels by apt_to_r261).  Windows are independent: one host thread each.
inline void build_comb_window_of(const u256& px, const u256& py, int bits, int j, apt* out_row) {
    apt* bases = new apt[j + 1];
    comb_bases(px, py, bits, j + 1, bases);
    comb_window(bases[j], 1 << (bits - 1), out_row);
    delete[] bases;
}
// the same for G: the fixed-base comb of the G phase
inline void build_gcomb_window(int bits, int j, apt* out_row) {
    const u256 gx = {{0xD898C296u, 0xF4A13945u, 0x2DEB33A0u, 0x77037D81u, 0x63A440F2u, 0xF8BCE6E5u, 0xE12C4247u, 0x6B17D1F2u}};
    const u256 gy = {{0x37BF51F5u, 0xCBB64068u, 0x6B315ECEu, 0x2BCE3357u, 0x7C0F9E16u, 0x8EE7EB4Au, 0xFE1A7F9Bu, 0x4FE342E2u}};
    build_comb_window_of(gx, gy, bits, j, out_row);
}

}  // namespace sbv
