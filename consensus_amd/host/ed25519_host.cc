#include "ed25519_host.h"

#include <string.h>

#include "../csrc/ed25519_core.h"

namespace sbvhost {
namespace {
using sbv::ept;
using sbv::fe25;
using sbv::pniels;
using sbv::u32;

// ---- SHA-512 (FIPS 180-4) ---------------------------------------------------------------------------
const uint64_t K512[80] = {
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
inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
void compress(uint64_t st[8], const uint8_t* blk) {
    uint64_t w[80];
    for (int i = 0; i < 16; ++i) {
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v = (v << 8) | blk[8 * i + b];
        w[i] = v;
    }
    for (int i = 16; i < 80; ++i) {
        const uint64_t s0 = rotr64(w[i - 15], 1) ^ rotr64(w[i - 15], 8) ^ (w[i - 15] >> 7);
        const uint64_t s1 = rotr64(w[i - 2], 19) ^ rotr64(w[i - 2], 61) ^ (w[i - 2] >> 6);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint64_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 80; ++i) {
        const uint64_t S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41);
        const uint64_t t1 = h + S1 + ((e & f) ^ (~e & g)) + K512[i] + w[i];
        const uint64_t S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39);
        const uint64_t t2 = S0 + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
struct Sha512 {
    uint64_t st[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                      0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    uint8_t buf[128];
    size_t fill = 0;
    uint64_t total = 0;
    void update(const void* p, size_t n) {
        const uint8_t* s = (const uint8_t*)p;
        total += n;
        while (n) {
            const size_t take = n < 128 - fill ? n : 128 - fill;
            memcpy(buf + fill, s, take);
            fill += take; s += take; n -= take;
            if (fill == 128) { compress(st, buf); fill = 0; }
        }
    }
    void finish(uint8_t out[64]) {
        const uint64_t bits = total * 8;
        uint8_t pad[256] = {0x80};
        const size_t padlen = (fill < 112 ? 112 : 240) - fill;
        uint8_t len[16] = {0};
        for (int i = 0; i < 8; ++i) len[15 - i] = (uint8_t)(bits >> (8 * i));
        update(pad, padlen);
        update(len, 16);
        for (int i = 0; i < 8; ++i)
            for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(st[i] >> (56 - 8 * b));
    }
};

// ---- arithmetic mod L = 2^252 + 27742317777372353535851937790883648493 -------------------------------
// Plain shift-and-subtract long division: a signer does this twice per signature.
const uint64_t L_LIMBS[4] = {0x5812631A5CF5D3EDULL, 0x14DEF9DEA2F79CD6ULL, 0, 0x1000000000000000ULL};
bool ge_l(const uint64_t x[5]) {
    if (x[4]) return true;
    for (int i = 3; i >= 0; --i) {
        if (x[i] > L_LIMBS[i]) return true;
        if (x[i] < L_LIMBS[i]) return false;
    }
    return true;
}
// x: 512-bit little-endian limbs -> out = x mod L
void mod_l(const uint64_t x[8], uint64_t out[4]) {
    uint64_t r[5] = {0, 0, 0, 0, 0};
    for (int bit = 511; bit >= 0; --bit) {
        for (int i = 4; i > 0; --i) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
        r[0] = (r[0] << 1) | ((x[bit >> 6] >> (bit & 63)) & 1);
        if (ge_l(r)) {
            unsigned __int128 borrow = 0;
            for (int i = 0; i < 5; ++i) {
                const unsigned __int128 li = i < 4 ? L_LIMBS[i] : 0;
                const unsigned __int128 d = (unsigned __int128)r[i] - li - borrow;
                r[i] = (uint64_t)d;
                borrow = (d >> 64) & 1;
            }
        }
    }
    memcpy(out, r, 32);
}
void load_le(uint64_t* dst, const uint8_t* src, int limbs) {
    for (int i = 0; i < limbs; ++i) {
        uint64_t v = 0;
        for (int b = 7; b >= 0; --b) v = (v << 8) | src[8 * i + b];
        dst[i] = v;
    }
}
void store_le(uint8_t* dst, const uint64_t* src, int limbs) {
    for (int i = 0; i < limbs; ++i)
        for (int b = 0; b < 8; ++b) dst[8 * i + b] = (uint8_t)(src[i] >> (8 * b));
}
// out = (a * b + c) mod L, all 256-bit little-endian
void muladd_mod_l(const uint64_t a[4], const uint64_t b[4], const uint64_t c[4], uint64_t out[4]) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 carry = 0;
        for (int j = 0; j < 4; ++j) {
            const unsigned __int128 cur = (unsigned __int128)a[i] * b[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)cur;
            carry = cur >> 64;
        }
        t[i + 4] = (uint64_t)carry;
    }
    unsigned __int128 carry = 0;
    for (int i = 0; i < 8; ++i) {
        const unsigned __int128 cur = (unsigned __int128)t[i] + (i < 4 ? c[i] : 0) + carry;
        t[i] = (uint64_t)cur;
        carry = cur >> 64;
    }
    mod_l(t, out);
}

// ---- [s]B and point encoding -------------------------------------------------------------------------
void scalar_mult_base(const uint8_t s[32], uint8_t enc[32]) {
    const uint32_t bxw[8] = {0x8F25D51Au, 0xC9562D60u, 0x9525A7B2u, 0x692CC760u, 0xFDD6DC5Cu, 0xC0A4E231u, 0xCD6E53FEu, 0x216936D3u};
    const uint32_t byw[8] = {0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u};
    fe25 bx, by;
    sbv::fe25_from_words(bx, bxw); sbv::fe25_carry(bx, bx);
    sbv::fe25_from_words(by, byw); sbv::fe25_carry(by, by);
    ept base;
    base.X = bx; base.Y = by; base.Z = sbv::fe25_one(); sbv::fe25_mul(base.T, bx, by);
    pniels bn;
    sbv::ed_to_pniels(bn, base);
    ept r;
    sbv::ed_set_ident(r);
    for (int bit = 255; bit >= 0; --bit) {
        sbv::ed_dbl(r, r);
        sbv::ed_add_pniels(r, bn, false, ((s[bit >> 3] >> (bit & 7)) & 1) == 0);
    }
    fe25 zi, x, y;
    sbv::fe25_inv(zi, r.Z);
    sbv::fe25_mul(x, r.X, zi);
    sbv::fe25_mul(y, r.Y, zi);
    sbv::u256 yw;
    sbv::fe25_freeze(yw, y);
    yw.v[7] |= (sbv::fe25_is_negative(x) ? 1u : 0u) << 31;
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 4; ++b) enc[4 * i + b] = (uint8_t)(yw.v[i] >> (8 * b));
}
void expand(const uint8_t seed[32], uint8_t a[32], uint8_t prefix[32]) {
    uint8_t h[64];
    sha512(seed, 32, h);
    memcpy(a, h, 32);
    a[0] &= 248; a[31] &= 127; a[31] |= 64;
    memcpy(prefix, h + 32, 32);
}
}  // namespace

void sha512(const void* msg, size_t len, uint8_t out[64]) {
    Sha512 s;
    s.update(msg, len);
    s.finish(out);
}

void ed25519_public_key(const uint8_t seed[32], uint8_t a_enc[32]) {
    uint8_t a[32], prefix[32];
    expand(seed, a, prefix);
    scalar_mult_base(a, a_enc);
}

void ed25519_hram(const uint8_t r_enc[32], const uint8_t a_enc[32], const void* msg, size_t len, uint8_t k[32]) {
    Sha512 s;
    s.update(r_enc, 32);
    s.update(a_enc, 32);
    s.update(msg, len);
    uint8_t h[64];
    s.finish(h);
    uint64_t x[8], o[4];
    load_le(x, h, 8);
    mod_l(x, o);
    store_le(k, o, 4);
}

void ed25519_sign(const uint8_t seed[32], const void* msg, size_t len, uint8_t sig[64]) {
    uint8_t a[32], prefix[32], a_enc[32], h[64];
    expand(seed, a, prefix);
    scalar_mult_base(a, a_enc);
    Sha512 s;
    s.update(prefix, 32);
    s.update(msg, len);
    s.finish(h);
    uint64_t x[8], r[4], k[4], al[4], S[4];
    load_le(x, h, 8);
    mod_l(x, r);
    uint8_t rb[32], kb[32];
    store_le(rb, r, 4);
    scalar_mult_base(rb, sig);                              // R
    ed25519_hram(sig, a_enc, msg, len, kb);
    load_le(k, kb, 4);
    load_le(al, a, 4);
    muladd_mod_l(k, al, r, S);                              // S = r + k * a mod L
    store_le(sig + 32, S, 4);
}

}  // namespace sbvhost
