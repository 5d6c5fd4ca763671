// sbv_host_util.cc — host-side helpers of libsbv.so that turn wire data into 160-byte tuples:
// strict DER parsing of ECDSA-Sig-Value (Go crypto/ecdsa.parseSignature, i.e.
// x/crypto/cryptobyte rules) and SHA-256 of Signature.Msg (pkg/types/types.go:25-29; the
// reference hashes with crypto/sha256 at pkg/types/types.go:64-69).
#include <string.h>

#include "../../include/sbv.h"

namespace {

struct Reader {
    const uint8_t* p;
    size_t n;
};

// cryptobyte.String.ReadASN1 for a single-byte tag with DER definite length (<= 4 length bytes)
bool read_tlv(Reader& in, uint8_t want_tag, Reader& body) {
    if (in.n < 2) return false;
    const uint8_t tag = in.p[0], lb = in.p[1];
    if ((tag & 0x1f) == 0x1f) return false;          // high-tag-number form
    size_t len = 0, hdr = 2;
    if (lb & 0x80) {
        const unsigned ll = lb & 0x7f;
        if (ll == 0 || ll > 4) return false;         // indefinite, or longer than cryptobyte accepts
        if (in.n < 2 + (size_t)ll) return false;
        for (unsigned i = 0; i < ll; ++i) len = (len << 8) | in.p[2 + i];
        if (len < 128) return false;                 // must have used the short form
        if ((len >> ((ll - 1) * 8)) == 0) return false;   // leading zero length octet
        hdr = 2 + ll;
    } else {
        len = lb;
    }
    if (in.n < hdr + len) return false;
    if (tag != want_tag) return false;
    body.p = in.p + hdr;
    body.n = len;
    in.p += hdr + len;
    in.n -= hdr + len;
    return true;
}

// cryptobyte ReadASN1Integer into big-endian bytes: minimal, non-negative; at most 32
// significant bytes (bigmod setBytes would reject more)
bool read_uint256(Reader& in, uint8_t out[32]) {
    Reader b;
    if (!read_tlv(in, 0x02, b)) return false;
    if (b.n == 0) return false;
    if (b.n > 1) {
        if (b.p[0] == 0x00 && !(b.p[1] & 0x80)) return false;
        if (b.p[0] == 0xff && (b.p[1] & 0x80)) return false;
    }
    if (b.p[0] & 0x80) return false;
    while (b.n > 1 && b.p[0] == 0) { ++b.p; --b.n; }
    if (b.n > 32) return false;
    memset(out, 0, 32);
    memcpy(out + (32 - b.n), b.p, b.n);
    return true;
}

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

const uint32_t kSha256K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

void sha256_compress(uint32_t st[8], const uint8_t* blk) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
        w[i] = (uint32_t)blk[4 * i] << 24 | (uint32_t)blk[4 * i + 1] << 16 | (uint32_t)blk[4 * i + 2] << 8 | blk[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t v[8];
    memcpy(v, st, sizeof v);
    for (int i = 0; i < 64; ++i) {
        const uint32_t t1 = v[7] + (rotr(v[4], 6) ^ rotr(v[4], 11) ^ rotr(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + kSha256K[i] + w[i];
        const uint32_t t2 = (rotr(v[0], 2) ^ rotr(v[0], 13) ^ rotr(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
        v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1;
        v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
    }
    for (int i = 0; i < 8; ++i) st[i] += v[i];
}

void sha256(const uint8_t* msg, size_t len, uint8_t out[32]) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t off = 0;
    for (; off + 64 <= len; off += 64) sha256_compress(st, msg + off);
    uint8_t pad[128] = {0};
    const size_t rem = len - off;
    memcpy(pad, msg + off, rem);
    pad[rem] = 0x80;
    const size_t total = rem < 56 ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) pad[total - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_compress(st, pad);
    if (total == 128) sha256_compress(st, pad + 64);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
    }
}

}  // namespace

extern "C" int sbv_p256_parse_der(const uint8_t* der, size_t len, uint8_t out_rs[64]) {
    if (!out_rs) return SBV_EINVAL;
    memset(out_rs, 0, 64);
    if (!der) return SBV_EPARSE;
    Reader in{der, len}, seq;
    uint8_t tmp[64];
    if (!read_tlv(in, 0x30, seq) || in.n != 0) return SBV_EPARSE;
    if (!read_uint256(seq, tmp) || !read_uint256(seq, tmp + 32) || seq.n != 0) return SBV_EPARSE;
    memcpy(out_rs, tmp, 64);
    return SBV_OK;
}

extern "C" int sbv_sha256_batch(const uint8_t* msgs, const uint64_t* offsets, size_t n, uint8_t* out_hashes) {
    if (n == 0) return SBV_OK;
    if (!msgs || !offsets || !out_hashes) return SBV_EINVAL;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return SBV_EINVAL;
        sha256(msgs + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), out_hashes + 32 * i);
    }
    return SBV_OK;
}
