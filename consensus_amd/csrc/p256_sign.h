// p256_sign.h — ECDSA P-256 signing, one signature per lane, with the deterministic nonce of RFC 6979 §3.2
// (HMAC-SHA256 DRBG).  SURVEY.md §8(f) row 4 ("batch signing"): the batch form of api.Signer.Sign / SignProposal
// (pkg/api/dependencies.go:46-52), bit-identical to the host Signer of consensus_amd/host (p256_host.cc: sign_rfc6979)
// and pinned on the RFC 6979 A.2.5 vectors.  Go's crypto/ecdsa.Sign draws a hedged random nonce, so a Go signature
// over the same input differs in k — any (r, s) this produces verifies under crypto/ecdsa.VerifyASN1, which is what the
// protocol needs of a Signer.
//
// Per lane: 22 SHA-256 compressions for the nonce, k * G from the comb of G (13 mixed additions with the 20-bit comb),
// one field inversion for the affine x, one scalar inversion for k^-1.
//
// NOT constant-time: the comb lookups are indexed by digits of the secret nonce and the table lives in HBM.  This is
// for test traffic and for a trusted, single-tenant host that already holds the keys in memory; a deployment that
// shares the GPU with untrusted work keeps signing on the CPU.
#pragma once
#include "p256_comb29.h"
#include "p256_sc29.h"
#include "sha256_dev.h"

namespace sbv {

SBV_HD_NOINLINE void sha256_compress_call(u32 st[8], const u32 w[16]) { sha256_compress(st, w); }

SBV_HD void sha256_iv(u32 st[8]) {
    const u32 iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) st[i] = iv[i];
}
// HMAC-SHA256 with a 32-byte key: the hash states after the key ^ ipad and key ^ opad blocks
struct hmac_key { u32 ist[8], ost[8]; };
SBV_HD void hmac_set_key(hmac_key& hk, const u32 K[8]) {
    u32 w[16];
    SBV_UNROLL
    for (int i = 0; i < 16; ++i) w[i] = (i < 8 ? K[i] : 0u) ^ 0x36363636u;
    sha256_iv(hk.ist);
    sha256_compress_call(hk.ist, w);
    SBV_UNROLL
    for (int i = 0; i < 16; ++i) w[i] = (i < 8 ? K[i] : 0u) ^ 0x5c5c5c5cu;
    sha256_iv(hk.ost);
    sha256_compress_call(hk.ost, w);
}
// outer hash over the 32-byte inner digest: one block, total length 64 + 32 bytes
SBV_HD void hmac_outer(const hmac_key& hk, const u32 inner[8], u32 out[8]) {
    u32 w[16];
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) w[i] = inner[i];
    w[8] = 0x80000000u;
    SBV_UNROLL
    for (int i = 9; i < 15; ++i) w[i] = 0;
    w[15] = (64 + 32) * 8;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) out[i] = hk.ost[i];
    sha256_compress_call(out, w);
}
// HMAC(K, V), V = 32 bytes
SBV_HD void hmac_v(const hmac_key& hk, const u32 V[8], u32 out[8]) {
    u32 w[16], in[8];
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) w[i] = V[i];
    w[8] = 0x80000000u;
    SBV_UNROLL
    for (int i = 9; i < 15; ++i) w[i] = 0;
    w[15] = (64 + 32) * 8;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) in[i] = hk.ist[i];
    sha256_compress_call(in, w);
    hmac_outer(hk, in, out);
}
// HMAC(K, V || tag || x || h), x and h 32 bytes each (97 bytes); tail_only: HMAC(K, V || tag) (33 bytes)
SBV_HD void hmac_v_tag(const hmac_key& hk, const u32 V[8], u32 tag, const u32* x, const u32* h, bool tail_only, u32 out[8]) {
    u32 w[16], in[8];
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { w[i] = V[i]; in[i] = hk.ist[i]; }
    if (tail_only) {
        w[8] = (tag << 24) | 0x00800000u;
        SBV_UNROLL
        for (int i = 9; i < 15; ++i) w[i] = 0;
        w[15] = (64 + 33) * 8;
        sha256_compress_call(in, w);
    } else {
        // the 65 bytes tag | x | h start at word 8, shifted by one byte against the word grid
        w[8] = (tag << 24) | (x[0] >> 8);
        SBV_UNROLL
        for (int j = 1; j < 8; ++j) w[8 + j] = (x[j - 1] << 24) | (x[j] >> 8);
        sha256_compress_call(in, w);
        w[0] = (x[7] << 24) | (h[0] >> 8);
        SBV_UNROLL
        for (int j = 1; j < 8; ++j) w[j] = (h[j - 1] << 24) | (h[j] >> 8);
        w[8] = (h[7] << 24) | 0x00800000u;
        SBV_UNROLL
        for (int i = 9; i < 15; ++i) w[i] = 0;
        w[15] = (64 + 97) * 8;
        sha256_compress_call(in, w);
    }
    hmac_outer(hk, in, out);
}

// big-endian words (w[0] most significant) <-> u256 (v[0] least significant)
SBV_HD void u256_from_be_words(u256& r, const u32 w[8]) {
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) r.v[i] = w[7 - i];
}
SBV_HD void u256_to_be_words(u32 w[8], const u256& a) {
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) w[i] = a.v[7 - i];
}

// (r, s) for a candidate nonce k (plain integers); false when k, r or s falls outside [1, N-1]
SBV_HD bool sign29_with_nonce(const u256& d, const u256& k, const u256& e, const gcomb& gc, u256& r, u256& s) {
    const sc n_ = sc_n();
    if (is_zero256(k) || !lt256(k, n_)) return false;
    xyzz R;
    gphase29_point(R, k, gc);
    if (R.inf) return false;                       // cannot happen for 0 < k < N
    fe29 zi, xM, x1, one_plain = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
    f29_inv_ct(zi, R.ZZ);
    f29_mul(xM, R.X, zi);                          // affine x, Montgomery domain
    f29_mul(x1, xM, one_plain);                    // plain
    u256 x;
    f29_store_canon(x.v, x1);
    sc_cond_sub_n(r, x, 0);                        // x < p < 2N
    if (is_zero256(r)) return false;
    fe29 kL, kM, kinv, rL, dL, dM, eL, t, sM;
    f29_unpack(kL, k.v);
    s29_mul(kM, kL, s29_r2());
    s29_inv_ct(kinv, kM);                          // k^-1 (Montgomery)
    f29_unpack(rL, r.v);
    f29_unpack(dL, d.v);
    f29_unpack(eL, e.v);
    s29_mul(dM, dL, s29_r2());
    s29_mul(t, dM, rL);                            // Montgomery(d) * plain(r) = plain(r d)
    f29_add(t, t, eL);                             // e + r d, limbs < 2^30
    s29_mul(sM, kinv, t);                          // plain(k^-1 (e + r d))
    s29_store_canon(s, sM);
    return !is_zero256(s);
}

// d_be, digest: 8 big-endian words each.  rs: r | s as 16 big-endian words.  false: d outside [1, N-1] (rs zeroed).
SBV_HD bool sign29_lane(const u32 d_be[8], const u32 digest[8], const gcomb& gc, u32 rs[16]) {
    const sc n_ = sc_n();
    u256 d, e;
    u256_from_be_words(d, d_be);
    u256_from_be_words(e, digest);
    sc_cond_sub_n(e, e, 0);                        // bits2int(h1) mod N; also the hashToNat of the signing equation
    SBV_UNROLL
    for (int i = 0; i < 16; ++i) rs[i] = 0;
    if (is_zero256(d) || !lt256(d, n_)) return false;
    u32 h1[8], V[8], K[8];
    u256_to_be_words(h1, e);                       // bits2octets
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { V[i] = 0x01010101u; K[i] = 0; }
    hmac_key hk;
    SBV_NOUNROLL
    for (u32 round = 0; round < 2; ++round) {      // RFC 6979 §3.2 d-g
        hmac_set_key(hk, K);
        hmac_v_tag(hk, V, round, d_be, h1, false, K);
        hmac_set_key(hk, K);
        hmac_v(hk, V, V);
    }
    SBV_NOUNROLL
    for (int attempt = 0; attempt < 64; ++attempt) {          // §3.2 h; a second pass has probability ~2^-32
        hmac_v(hk, V, V);
        u256 k, r, s;
        u256_from_be_words(k, V);
        if (sign29_with_nonce(d, k, e, gc, r, s)) {
            u256_to_be_words(rs, r);
            u256_to_be_words(rs + 8, s);
            return true;
        }
        hmac_v_tag(hk, V, 0, d_be, h1, true, K);
        hmac_set_key(hk, K);
        hmac_v(hk, V, V);
    }
    return false;
}

}  // namespace sbv
