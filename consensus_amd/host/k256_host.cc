// k256_host.cc — the api.Signer half (pkg/api/dependencies.go:46-52) of the secp256k1 scheme: key derivation and deterministic
// ECDSA signing (RFC 6979 with HMAC-SHA256) on the host, over the product's own secp256k1 arithmetic (consensus_amd/csrc/
// k256_core.h compiled for the host).  One signature per sequence (internal/bft/view.go:481): no device work is worth it here.
#include "k256_host.h"

#include <string.h>

#include <mutex>
#include <vector>

#include "../csrc/k256_core.h"
#include "p256_host.h"

namespace sbvhost {

using namespace sbv;

namespace {

void to_be32(uint8_t out[32], const u256& v) {
    for (int i = 0; i < 8; ++i) {
        const u32 w = v.v[7 - i];
        out[4 * i] = (uint8_t)(w >> 24); out[4 * i + 1] = (uint8_t)(w >> 16); out[4 * i + 2] = (uint8_t)(w >> 8); out[4 * i + 3] = (uint8_t)w;
    }
}
bool valid_scalar(const u256& d) { return !is_zero256(d) && lt256(d, k256_n_words()); }

// 8-bit signed comb of G for the signer: 33 windows x 128 entries (270 KB), built once
const kapt* gtable8() {
    static std::vector<kapt> tab;
    static std::once_flag once;
    std::call_once(once, [] {
        tab.resize((size_t)33 * 128);
        for (int j = 0; j < 33; ++j) k256_build_g_window_bits(8, j, tab.data() + (size_t)j * 128, j == 32 ? 1 : 128);
    });
    return tab.data();
}

// affine k * G, k in [1, n - 1]: 33 exact mixed additions from the comb (k + 0x80..80: byte j minus 128 is the digit of window j)
void base_mul_affine(const u256& k, u256& x, u256& y) {
    const kapt* gt = gtable8();
    u256 kk;
    const u32 top = add_const_limbs(kk, k, 0x80808080u);
    kjpt R;
    kpt_set_inf(R);
    for (int j = 0; j < 33; ++j) {
        int idx; bool neg, skip;
        comb_digit(kk, top, j, idx, neg, skip);
        kfe ex, ey;
        kapt_load(ex, ey, gt + (size_t)j * 128 + idx);
        kpt_madd(R, R, ex, ey, neg, skip);
    }
    kfe zi, zi2, zi3, ax, ay;
    kfe_inv(zi, R.Z);
    kfe_sqr(zi2, zi);
    kfe_mul(zi3, zi2, zi);
    kfe_mul(ax, R.X, zi2);
    kfe_mul(ay, R.Y, zi3);
    kfe_to_words(x, ax);
    kfe_to_words(y, ay);
}

void hmac_sha256(const uint8_t key[32], const bytes& data, uint8_t out[32]) {
    uint8_t ipad[64], opad[64];
    for (int i = 0; i < 64; ++i) { const uint8_t k = i < 32 ? key[i] : 0; ipad[i] = k ^ 0x36; opad[i] = k ^ 0x5c; }
    bytes inner((const char*)ipad, 64);
    inner += data;
    uint8_t ih[32];
    sha256(inner.data(), inner.size(), ih);
    bytes outer((const char*)opad, 64);
    outer.append((const char*)ih, 32);
    sha256(outer.data(), outer.size(), out);
}

}  // namespace

bool k256_pubkey_from_private(const uint8_t d_be[32], uint8_t q[64]) {
    u256 d, x, y;
    from_be32(d, d_be);
    if (!valid_scalar(d)) return false;
    base_mul_affine(d, x, y);
    to_be32(q, x);
    to_be32(q + 32, y);
    return true;
}

bool k256_sign_with_nonce(const uint8_t d_be[32], const uint8_t k_be[32], const uint8_t digest[32], uint8_t rs[64]) {
    u256 d, k, e, x, y;
    from_be32(d, d_be); from_be32(k, k_be); from_be32(e, digest);
    if (!valid_scalar(d) || !valid_scalar(k)) return false;
    ksc_cond_sub_n(e, e);
    base_mul_affine(k, x, y);
    u256 r;
    ksc_cond_sub_n(r, x);                      // x < p < 2 n
    if (is_zero256(r)) return false;
    u256 ki, t, s, sum;
    ksc_inv(ki, k);
    ksc_mul(t, r, d);
    const u32 c = add256(sum, t, e);           // r d + e < 2 n: one conditional subtraction (with the carry)
    {
        u256 dd;
        const u32 bw = sub256(dd, sum, k256_n_words());
        select256(t, c != 0 || bw == 0, dd, sum);
    }
    ksc_mul(s, ki, t);
    if (is_zero256(s)) return false;
    to_be32(rs, r);
    to_be32(rs + 32, s);
    return true;
}

// RFC 6979 §3.2 with HMAC-SHA256, qlen = hlen = 256
bool k256_sign_rfc6979(const uint8_t d_be[32], const uint8_t digest[32], uint8_t rs[64]) {
    u256 d, h;
    from_be32(d, d_be);
    if (!valid_scalar(d)) return false;
    from_be32(h, digest);
    ksc_cond_sub_n(h, h);
    uint8_t h1[32];
    to_be32(h1, h);                            // bits2octets
    uint8_t V[32], K[32];
    memset(V, 0x01, 32); memset(K, 0x00, 32);
    for (int round = 0; round < 2; ++round) {
        bytes m((const char*)V, 32);
        m.push_back((char)round);
        m.append((const char*)d_be, 32);
        m.append((const char*)h1, 32);
        hmac_sha256(K, m, K);
        hmac_sha256(K, bytes((const char*)V, 32), V);
    }
    for (;;) {
        hmac_sha256(K, bytes((const char*)V, 32), V);
        if (k256_sign_with_nonce(d_be, V, digest, rs)) return true;   // rejects k = 0, k >= n, r = 0, s = 0
        bytes m((const char*)V, 32);
        m.push_back('\0');
        hmac_sha256(K, m, K);
        hmac_sha256(K, bytes((const char*)V, 32), V);
    }
}

}  // namespace sbvhost
