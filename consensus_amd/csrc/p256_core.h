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

// v + c as a 257-bit value: returns the low 256 bits, `top` = bit 256
SBV_HD u32 add_const_limbs(u256& out, const u256& v, u32 c_limb) {
    u32 c = 0;
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) out.v[l] = addc(v.v[l], c_limb, c);
    return c;
}


// ---- G phase (in-step grouping) ----------------------------------------------------------------------


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


// ---- registered-key form, several lanes per signature (latency form) ----------------------------------
// A quorum-sized micro-batch (BASELINE.json's second metric: N = 16 -> 15 concurrent VerifyConsenterSig) puts
// one wavefront on a 256-CU GPU, and a lane's chain of 50 dependent additions is what the caller waits for.
// With no doublings in the registered-key form the 50 comb terms are independent, so SBV_COOP_LANES lanes
// each sum every SBV_COOP_LANES-th term (7 or 6 additions) and the partial sums are combined by a butterfly
// of exact Jacobian additions across the lanes (3 levels): ~10 additions deep instead of 50.
#define SBV_COOP_LANES 8


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
