// sha512_dev.h — the Ed25519 message front end as per-lane device code: k = SHA-512(R | A | M) mod L
// (RFC 8032 §5.1.7 step 2; Go crypto/ed25519.verify), the fourth field of the 128-byte tuple.  The host
// computes it today for the tuple entry (sbv_ed25519_make_tuples); with this, a caller hands over raw
// signatures, keys and messages and the GPU does the hashing too — the counterpart of sha256_dev.h for
// the Ed25519 variant (BASELINE.json configs[4]).
//
// One message per lane, block by block (lengths differ; lanes of a wavefront finish at different block
// counts — accepted: two SHA-512 blocks are ~10 % of a grouped Ed25519 verification).
// Reduction mod L = 2^252 + c (c ~ 2^124.4): three folds x = x_lo + 2^252 x_hi -> x_lo - c x_hi (kept
// non-negative by adding a multiple of L), then conditional subtractions (mod_l_512).  Same source compiled
// for the host by tests/emul.
#pragma once
#include "sbv_common.h"

namespace sbv {

SBV_HD u64 sha_rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }

SBV_HD void sha512_compress(u64 st[8], const u64 w_in[16]) {
    const u64 K[80] = {
        0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL, 0x59f111f1b605d019ULL,
        0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL, 0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL,
        0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL, 0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL,
        0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL, 0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL,
        0x983e5152ee66dfabULL, 0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
        0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL, 0x53380d139d95b3dfULL,
        0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL, 0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL,
        0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL, 0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL,
        0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL, 0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL,
        0x5b9cca4f7763e373ULL, 0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
        0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL, 0xd186b8c721c0c207ULL,
        0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL, 0x113f9804bef90daeULL, 0x1b710b35131c471bULL,
        0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL, 0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL,
        0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
    u64 w[16];
    SBV_UNROLL
    for (int i = 0; i < 16; ++i) w[i] = w_in[i];
    u64 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    SBV_NOUNROLL
    for (int i = 0; i < 80; ++i) {
        if (i >= 16) {
            const u64 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const u64 s0 = sha_rotr64(w15, 1) ^ sha_rotr64(w15, 8) ^ (w15 >> 7);
            const u64 s1 = sha_rotr64(w2, 19) ^ sha_rotr64(w2, 61) ^ (w2 >> 6);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        const u64 t1 = h + (sha_rotr64(e, 14) ^ sha_rotr64(e, 18) ^ sha_rotr64(e, 41)) + ((e & f) ^ (~e & g)) + K[i] + w[i & 15];
        const u64 t2 = (sha_rotr64(a, 28) ^ sha_rotr64(a, 34) ^ sha_rotr64(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// SHA-512 of the concatenation r32 | a32 | msg[0..mlen): digest as 8 big-endian-valued 64-bit words
SBV_HD void sha512_ram(const uint8_t* r32, const uint8_t* a32, const uint8_t* msg, size_t mlen, u64 out[8]) {
    u64 st[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    const size_t len = 64 + mlen;
    const size_t total_blocks = (len + 17 + 127) / 128;
    for (size_t blk = 0; blk < total_blocks; ++blk) {
        u64 w[16];
        SBV_NOUNROLL
        for (int i = 0; i < 16; ++i) {
            u64 word = 0;
            SBV_NOUNROLL
            for (int k = 0; k < 8; ++k) {
                const size_t pos = blk * 128 + (size_t)i * 8 + k;
                u64 byte = 0;
                if (pos < 32) byte = r32[pos];
                else if (pos < 64) byte = a32[pos - 32];
                else if (pos < len) byte = msg[pos - 64];
                else if (pos == len) byte = 0x80u;
                word = (word << 8) | byte;
            }
            w[i] = word;
        }
        if (blk == total_blocks - 1) {
            w[14] = 0;                               // lengths here are far below 2^61 bytes
            w[15] = (u64)len * 8;
        }
        sha512_compress(st, w);
    }
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) out[i] = st[i];
}

// ---- 512-bit little-endian integer mod L ----------------------------------------------------------------
// x[16] = 32-bit limbs, little-endian.  L = 2^252 + c, c = 0x14DEF9DEA2F79CD65812631A5CF5D3ED (125 bits), so
// 2^252 == -c (mod L) and a value v = lo + 2^252 hi folds to lo - c*hi.  To stay unsigned a fold computes
//     v_next = lo + (M - c*hi),   M = L << (32 s)  a multiple of L chosen >= the bound of c*hi:
//   fold 0: v  < 2^512, hi < 2^260, c*hi < 2^385, M = L << 160 (>= 2^412)  ->  v1 < 2^414
//   fold 1:             hi < 2^162, c*hi < 2^287, M = L << 64  (>= 2^316)  ->  v2 < 2^318
//   fold 2:             hi < 2^66,  c*hi < 2^191, M = L                    ->  v3 < 2^252 + L < 2L
// then at most one subtraction of L (two are made).
SBV_HD void mod_l_fold(u32 v[17], int shift_limbs) {
    const u32 C[4] = {0x5CF5D3EDu, 0x5812631Au, 0xA2F79CD6u, 0x14DEF9DEu};
    const u32 Lw[8] = {0x5CF5D3EDu, 0x5812631Au, 0xA2F79CD6u, 0x14DEF9DEu, 0u, 0u, 0u, 0x10000000u};
    u32 hi[10];                                              // v >> 252: bit 252 = limb 7, bit 28
    SBV_UNROLL
    for (int i = 0; i < 10; ++i) {
        const u32 a = 7 + i < 17 ? v[7 + i] : 0u;
        const u32 b = 8 + i < 17 ? v[8 + i] : 0u;
        hi[i] = (a >> 28) | (b << 4);
    }
    u32 p[17];                                               // c * hi: 4 x 10 limbs
    SBV_UNROLL
    for (int i = 0; i < 17; ++i) p[i] = 0;
    SBV_UNROLL
    for (int i = 0; i < 4; ++i) {
        u64 carry = 0;
        SBV_UNROLL
        for (int j = 0; j < 10; ++j) {
            const u64 t = (u64)C[i] * hi[j] + p[i + j] + carry;
            p[i + j] = (u32)t;
            carry = t >> 32;
        }
        p[i + 10] = (u32)carry;
    }
    u32 carry = 0, borrow = 0;                               // v = (v mod 2^252) + (L << 32 s) - p
    SBV_UNROLL
    for (int i = 0; i < 17; ++i) {
        const u32 lo = i < 7 ? v[i] : (i == 7 ? (v[7] & 0x0FFFFFFFu) : 0u);
        const int src = i - shift_limbs;
        const u32 m = (src >= 0 && src < 8) ? Lw[src] : 0u;
        const u32 sum = addc(lo, m, carry);
        v[i] = subb(sum, p[i], borrow);
    }
}
SBV_HD void mod_l_512(const u32 x[16], u32 out[8]) {
    const u32 Lw[8] = {0x5CF5D3EDu, 0x5812631Au, 0xA2F79CD6u, 0x14DEF9DEu, 0u, 0u, 0u, 0x10000000u};
    u32 v[17];
    SBV_UNROLL
    for (int i = 0; i < 16; ++i) v[i] = x[i];
    v[16] = 0;
    mod_l_fold(v, 5);
    mod_l_fold(v, 2);
    mod_l_fold(v, 0);
    SBV_UNROLL
    for (int it = 0; it < 2; ++it) {
        u32 d[9];
        u32 borrow = 0;
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) d[i] = subb(v[i], i < 8 ? Lw[i] : 0u, borrow);
        const bool ge = borrow == 0;
        SBV_UNROLL
        for (int i = 0; i < 9; ++i) v[i] = ge ? d[i] : v[i];
    }
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) out[i] = v[i];
}

// One lane of the Ed25519 front end: sig64 = R | S, a32 = A_enc, msg -> the 128-byte tuple R | S | A | k
// as 32 little-endian dwords (the layout sbv_ed25519_verify_batch takes).
SBV_HD void ed_msg_frontend_lane(const uint8_t* sig64, const uint8_t* a32, const uint8_t* msg, size_t mlen, u32* tuple_out) {
    u64 h[8];
    sha512_ram(sig64, a32, msg, mlen, h);
    // digest bytes b[0..63] = big-endian words; as a little-endian integer: limb j (32-bit) = bytes 4j..4j+3, LSB first
    u32 x[16];
    SBV_UNROLL
    for (int j = 0; j < 16; ++j) {
        const u64 word = h[j >> 1];
        const u32 half = (j & 1) ? (u32)word : (u32)(word >> 32);      // bytes 4j..4j+3 in big-endian order
        x[j] = bswap32(half);
    }
    u32 k[8];
    mod_l_512(x, k);
    SBV_UNROLL
    for (int j = 0; j < 16; ++j) {                                      // R | S: 64 bytes as they are
        const uint8_t* b = sig64 + 4 * j;
        tuple_out[j] = (u32)b[0] | ((u32)b[1] << 8) | ((u32)b[2] << 16) | ((u32)b[3] << 24);
    }
    SBV_UNROLL
    for (int j = 0; j < 8; ++j) {
        const uint8_t* b = a32 + 4 * j;
        tuple_out[16 + j] = (u32)b[0] | ((u32)b[1] << 8) | ((u32)b[2] << 16) | ((u32)b[3] << 24);
        tuple_out[24 + j] = k[j];
    }
}

}  // namespace sbv
