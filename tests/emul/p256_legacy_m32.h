// p256_legacy_m32.h — the round-1 lane forms on the 8 x 32-bit Montgomery field (carry chains), kept OUTSIDE the product as a second
// implementation of stage A and of the one-lane stage B for the CPU emulator (tests/emul/emul.cc) and the on-device checker
// (tools/devcheck.hip).  libsbv.so runs the carry-free forms only (p256_core.h: prep_chunk29, p256_comb29.h: verify29_lane_generic);
// these were moved out of consensus_amd/csrc/p256_core.h in round 5.  Test infrastructure.
#pragma once
#include "../../consensus_amd/csrc/p256_core.h"

namespace sbv {

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

}  // namespace sbv
