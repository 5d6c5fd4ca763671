#include "p256_host.h"

#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/sbv.h"
#include "../csrc/p256_core.h"

namespace sbvhost {

using namespace sbv;

void sha256(const void* msg, size_t len, uint8_t out[32]) {
    const uint64_t offs[2] = {0, (uint64_t)len};
    static const uint8_t empty = 0;
    sbv_sha256_batch(len ? (const uint8_t*)msg : &empty, offs, 1, out);
}
bytes sha256(const bytes& msg) {
    uint8_t h[32];
    sha256(msg.data(), msg.size(), h);
    return bytes((const char*)h, 32);
}

namespace {

const apt* gtable() {
    static std::vector<apt> tab;
    static std::once_flag once;
    std::call_once(once, [] { tab.resize((size_t)SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW); build_gtable(tab.data()); });
    return tab.data();
}

void to_be32(uint8_t out[32], const u256& v) {
    for (int i = 0; i < 8; ++i) {
        const u32 w = v.v[7 - i];
        out[4 * i] = (uint8_t)(w >> 24); out[4 * i + 1] = (uint8_t)(w >> 16); out[4 * i + 2] = (uint8_t)(w >> 8); out[4 * i + 3] = (uint8_t)w;
    }
}

// affine (x, y) of k*G, k in [1, N-1], plain integers
void base_mul_affine(const u256& k, u256& x, u256& y) {
    const apt* gt = gtable();
    u256 kk;
    const u32 top = add_const_limbs(kk, k, 0x80808080u);
    jpt R;
    pt_set_inf(R);
    for (int j = 0; j < 32; ++j) {
        const int d = (int)((kk.v[j >> 2] >> ((j & 3) * 8)) & 255u) - 128;
        const int ad = d < 0 ? -d : d;
        pt_add_mixed(R, gt[(size_t)j * SBV_GTAB_PER_WINDOW + (ad ? ad - 1 : 0)], d < 0, d == 0);
    }
    pt_add_mixed(R, gt[(size_t)32 * SBV_GTAB_PER_WINDOW], false, top == 0);
    fe zi, zi2, zi3, t;
    fe_inv(zi, R.Z);
    fe_sqr(zi2, zi);
    fe_mul(zi3, zi2, zi);
    fe_mul(t, R.X, zi2); fe_from_mont(x, t);
    fe_mul(t, R.Y, zi3); fe_from_mont(y, t);
}

bool valid_scalar(const u256& d) { return !is_zero256(d) && lt256(d, sc_n()); }

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

bool pubkey_from_private(const uint8_t d_be[32], uint8_t q[64]) {
    u256 d, x, y;
    from_be32(d, d_be);
    if (!valid_scalar(d)) return false;
    base_mul_affine(d, x, y);
    to_be32(q, x);
    to_be32(q + 32, y);
    return true;
}

bool sign_with_nonce(const uint8_t d_be[32], const uint8_t k_be[32], const uint8_t digest[32], uint8_t rs[64]) {
    u256 d, k, e, x, y;
    from_be32(d, d_be); from_be32(k, k_be); from_be32(e, digest);
    if (!valid_scalar(d) || !valid_scalar(k)) return false;
    sc_cond_sub_n(e, e, 0);
    base_mul_affine(k, x, y);
    u256 r = x;
    sc_cond_sub_n(r, r, 0);                    // x < p < 2N
    if (is_zero256(r)) return false;
    sc kM, ki, rM, t, s;
    sc_to_mont(kM, k); sc_inv(ki, kM);
    sc_to_mont(rM, r);
    sc_mul(t, rM, d);                          // Montgomery(r) * plain(d) = plain(r d)
    {   // t = r d + e mod N
        u256 sum; const u32 c = add256(sum, t, e); sc_cond_sub_n(t, sum, c);
    }
    sc_mul(s, ki, t);                          // Montgomery(k^-1) * plain = plain
    if (is_zero256(s)) return false;
    to_be32(rs, r);
    to_be32(rs + 32, s);
    return true;
}

bool sign_rfc6979(const uint8_t d_be[32], const uint8_t digest[32], uint8_t rs[64]) {
    u256 d, h;
    from_be32(d, d_be);
    if (!valid_scalar(d)) return false;
    from_be32(h, digest);
    sc_cond_sub_n(h, h, 0);
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
        if (sign_with_nonce(d_be, V, digest, rs)) return true;   // rejects k = 0, k >= N, r = 0, s = 0
        bytes m((const char*)V, 32);
        m.push_back('\0');
        hmac_sha256(K, m, K);
        hmac_sha256(K, bytes((const char*)V, 32), V);
    }
}

bytes der_encode_sig(const uint8_t rs[64]) {
    bytes body;
    for (int f = 0; f < 2; ++f) {
        const uint8_t* v = rs + 32 * f;
        int off = 0;
        while (off < 31 && v[off] == 0) ++off;
        bytes num;
        if (v[off] & 0x80) num.push_back('\0');
        num.append((const char*)v + off, 32 - off);
        body.push_back(0x02);
        body.push_back((char)num.size());
        body += num;
    }
    bytes out;
    out.push_back(0x30);
    out.push_back((char)body.size());       // <= 70 < 128: short form
    return out + body;
}

}  // namespace sbvhost
