// sha256_dev.h — SHA-256 (FIPS 180-4) and strict DER ECDSA-Sig-Value parsing as per-lane device code:
// SURVEY.md §8(f) row 1 — "device-side SHA-256 of Signature.Msg / request bytes -> hash field, fused in
// front of the verify kernel (and DER -> (r,s) parse on device)".  Removes the host-side ~1 us per
// signature per core that otherwise bounds large batches (§8e) long before the GPU does.
//
// One message per lane; a lane walks its own message block by block (lengths differ, so lanes of a
// wavefront finish at different block counts — accepted: SHA-256 is ~3 % of a registered-key verify).
// Same source compiled for the host by tests/emul.
#pragma once
#include "sbv_common.h"

namespace sbv {

SBV_HD u32 sha_rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }

SBV_HD void sha256_compress(u32 st[8], const u32 w_in[16]) {
    const u32 K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
        0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
        0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
        0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
        0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    u32 w[16];
    SBV_UNROLL
    for (int i = 0; i < 16; ++i) w[i] = w_in[i];
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    SBV_UNROLL
    for (int i = 0; i < 64; ++i) {
        if (i >= 16) {
            const u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const u32 s0 = sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3);
            const u32 s1 = sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        const u32 t1 = h + (sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i & 15];
        const u32 t2 = (sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// digest words (big-endian word values, i.e. out[0] holds bytes 0..3 of the digest as a number)
SBV_HD void sha256_msg(const uint8_t* msg, size_t len, u32 out[8]) {
    u32 st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    const size_t total_blocks = (len + 9 + 63) / 64;
    for (size_t blk = 0; blk < total_blocks; ++blk) {
        u32 w[16];
        SBV_UNROLL
        for (int i = 0; i < 16; ++i) {
            u32 word = 0;
            SBV_UNROLL
            for (int k = 0; k < 4; ++k) {
                const size_t pos = blk * 64 + (size_t)i * 4 + k;
                u32 byte = 0;
                if (pos < len) byte = msg[pos];
                else if (pos == len) byte = 0x80u;
                word = (word << 8) | byte;
            }
            w[i] = word;
        }
        if (blk == total_blocks - 1) {
            const u64 bits = (u64)len * 8;
            w[14] = (u32)(bits >> 32);
            w[15] = (u32)bits;
        }
        sha256_compress(st, w);
    }
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) out[i] = st[i];
}

// ---- strict DER (Go crypto/ecdsa.parseSignature = x/crypto/cryptobyte) ----------------------------------
// Parses SEQUENCE { INTEGER r, INTEGER s } from der[0..len) into 32-byte big-endian r and s (as 8
// big-endian-valued words each, word 0 = most significant).  Returns false (and all-zero words) on
// anything cryptobyte would refuse or on an integer of more than 32 significant bytes.
SBV_HD bool der_read_len(const uint8_t* p, size_t n, size_t& pos, size_t& out_len) {
    if (pos >= n) return false;
    const u32 lb = p[pos++];
    if (!(lb & 0x80u)) { out_len = lb; return true; }
    const u32 ll = lb & 0x7fu;
    if (ll == 0 || ll > 4 || pos + ll > n) return false;
    size_t v = 0;
    for (u32 i = 0; i < ll; ++i) v = (v << 8) | p[pos++];
    if (v < 128) return false;
    if ((v >> ((ll - 1) * 8)) == 0) return false;
    out_len = v;
    return true;
}
SBV_HD bool der_read_uint(const uint8_t* p, size_t n, size_t& pos, u32 out[8]) {
    if (pos >= n || p[pos] != 0x02u) return false;
    ++pos;
    size_t l = 0;
    if (!der_read_len(p, n, pos, l)) return false;
    if (l == 0 || pos + l > n) return false;
    const uint8_t* b = p + pos;
    if (l > 1) {
        if (b[0] == 0x00u && !(b[1] & 0x80u)) return false;
        if (b[0] == 0xffu && (b[1] & 0x80u)) return false;
    }
    if (b[0] & 0x80u) return false;
    size_t skip = 0;
    while (l - skip > 1 && b[skip] == 0) ++skip;
    const size_t sig = l - skip;
    if (sig > 32) return false;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (size_t k = 0; k < sig; ++k) {
        const size_t byte_index = 32 - sig + k;          // position inside the 32-byte big-endian field
        out[byte_index >> 2] |= (u32)b[skip + k] << (8 * (3 - (byte_index & 3)));
    }
    pos += l;
    return true;
}
SBV_HD bool der_parse_sig(const uint8_t* der, size_t len, u32 r[8], u32 s[8]) {
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { r[i] = 0; s[i] = 0; }
    size_t pos = 0, seq_len = 0;
    if (len < 2 || der[0] != 0x30u) return false;
    pos = 1;
    if (!der_read_len(der, len, pos, seq_len)) return false;
    if (pos + seq_len != len) return false;               // no trailing bytes, no truncation
    u32 rr[8], ss[8];
    if (!der_read_uint(der, len, pos, rr)) return false;
    if (!der_read_uint(der, len, pos, ss)) return false;
    if (pos != len) return false;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) { r[i] = rr[i]; s[i] = ss[i]; }
    return true;
}

// One lane of the message front end: msg -> hash, DER -> (r, s), written as the 96-byte r|s|hash record
// (24 big-endian dwords) that k_p256_prep_keyed consumes.  A DER failure leaves r = s = 0 (rejected later).
SBV_HD void msg_frontend_lane(const uint8_t* msg, size_t mlen, const uint8_t* der, size_t dlen, u32* rsh_out) {
    u32 r[8], s[8], h[8];
    (void)der_parse_sig(der, dlen, r, s);
    sha256_msg(msg, mlen, h);
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) {
        rsh_out[i] = bswap32(r[i]);        // stored as bytes in memory order (big-endian fields)
        rsh_out[8 + i] = bswap32(s[i]);
        rsh_out[16 + i] = bswap32(h[i]);
    }
}

}  // namespace sbv
