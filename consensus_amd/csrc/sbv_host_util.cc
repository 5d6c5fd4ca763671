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

// The same compression function on the SHA extensions of x86-64 (sha256rnds2 = two rounds, sha256msg1 / msg2 = the message
// schedule), chosen once at run time by CPUID; Go's crypto/sha256 — what the reference's applications hash with — does the
// same.  State registers: ABEF / CDGH as the instructions want them.  ~6x the portable loop's throughput on a 64-byte block.
#if defined(__x86_64__)
}  // namespace (the intrinsics headers must not be included inside it)
#include <cpuid.h>
#include <immintrin.h>
namespace {
__attribute__((target("sha,sse4.1,ssse3"))) void sha256_compress_shani(uint32_t st[8], const uint8_t* blk) {
    const __m128i mask = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);      // big-endian words
    __m128i tmp = _mm_loadu_si128((const __m128i*)&st[0]);          // DCBA
    __m128i s1 = _mm_loadu_si128((const __m128i*)&st[4]);           // HGFE
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                             // CDAB
    s1 = _mm_shuffle_epi32(s1, 0x1B);                               // EFGH
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                       // ABEF
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                            // CDGH
    const __m128i save0 = s0, save1 = s1;
    __m128i m[4];
    for (int i = 0; i < 16; ++i) {                                  // four rounds per step
        if (i < 4) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(blk + 16 * i)), mask);
        __m128i msg = _mm_add_epi32(m[i & 3], _mm_loadu_si128((const __m128i*)&kSha256K[4 * i]));
        s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
        if (i >= 3 && i <= 14) {                                    // W[4(i+1) ..] = sigma1 part: needs W[.. - 7] and W[.. - 2]
            const __m128i t = _mm_alignr_epi8(m[i & 3], m[(i - 1) & 3], 4);
            m[(i + 1) & 3] = _mm_sha256msg2_epu32(_mm_add_epi32(m[(i + 1) & 3], t), m[i & 3]);
        }
        msg = _mm_shuffle_epi32(msg, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
        if (i >= 1 && i <= 12) m[(i - 1) & 3] = _mm_sha256msg1_epu32(m[(i - 1) & 3], m[i & 3]);   // sigma0 part of the word group after next
    }
    s0 = _mm_add_epi32(s0, save0);
    s1 = _mm_add_epi32(s1, save1);
    tmp = _mm_shuffle_epi32(s0, 0x1B);                              // FEBA
    s1 = _mm_shuffle_epi32(s1, 0xB1);                               // DCHG
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);                            // DCBA
    s1 = _mm_alignr_epi8(s1, tmp, 8);                               // HGFE
    _mm_storeu_si128((__m128i*)&st[0], s0);
    _mm_storeu_si128((__m128i*)&st[4], s1);
}
bool cpu_has_sha() {
    unsigned a = 0, b = 0, c = 0, d = 0;
    if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
    const bool sha = (b >> 29) & 1u;
    if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
    return sha && ((c >> 19) & 1u) && ((c >> 9) & 1u);              // + SSE4.1, SSSE3
}
#else
void sha256_compress_shani(uint32_t st[8], const uint8_t* blk) { sha256_compress(st, blk); }
bool cpu_has_sha() { return false; }
#endif
// SBV_SHA_PORTABLE=1 keeps the portable loop (tests compare the two)
const bool g_sha_ext = [] { const char* e = getenv("SBV_SHA_PORTABLE"); return !(e && e[0] == '1') && cpu_has_sha(); }();
inline void sha256_block(uint32_t st[8], const uint8_t* blk) {
    if (g_sha_ext) sha256_compress_shani(st, blk);
    else sha256_compress(st, blk);
}

void sha256(const uint8_t* msg, size_t len, uint8_t out[32]) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t off = 0;
    for (; off + 64 <= len; off += 64) sha256_block(st, msg + off);
    uint8_t pad[128] = {0};
    const size_t rem = len - off;
    memcpy(pad, msg + off, rem);
    pad[rem] = 0x80;
    const size_t total = rem < 56 ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) pad[total - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_block(st, pad);
    if (total == 128) sha256_block(st, pad + 64);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
    }
}

// ---- SHA-512 and reduction mod L for Ed25519's k = H(R || A || M) ------------------------------------
const uint64_t kSha512K[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull,
    0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull,
    0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull, 0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull,
    0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
    0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull, 0x983e5152ee66dfabull,
    0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
    0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull,
    0x53380d139d95b3dfull, 0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
    0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull,
    0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull, 0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull,
    0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull,
    0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull,
    0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull,
    0x113f9804bef90daeull, 0x1b710b35131c471bull, 0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull,
    0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
void sha512_compress(uint64_t st[8], const uint8_t* blk) {
    uint64_t w[80];
    for (int i = 0; i < 16; ++i) { uint64_t x = 0; for (int j = 0; j < 8; ++j) x = (x << 8) | blk[8 * i + j]; w[i] = x; }
    for (int i = 16; i < 80; ++i)
        w[i] = w[i - 16] + (rotr64(w[i - 15], 1) ^ rotr64(w[i - 15], 8) ^ (w[i - 15] >> 7)) + w[i - 7] +
               (rotr64(w[i - 2], 19) ^ rotr64(w[i - 2], 61) ^ (w[i - 2] >> 6));
    uint64_t v[8];
    memcpy(v, st, sizeof v);
    for (int i = 0; i < 80; ++i) {
        const uint64_t t1 = v[7] + (rotr64(v[4], 14) ^ rotr64(v[4], 18) ^ rotr64(v[4], 41)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + kSha512K[i] + w[i];
        const uint64_t t2 = (rotr64(v[0], 28) ^ rotr64(v[0], 34) ^ rotr64(v[0], 39)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
        v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1;
        v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
    }
    for (int i = 0; i < 8; ++i) st[i] += v[i];
}
// SHA-512 over three concatenated pieces (R, A, M) without building the concatenation
void sha512_3(const uint8_t* a, size_t al, const uint8_t* b, size_t bl, const uint8_t* c, size_t cl, uint8_t out[64]) {
    uint64_t st[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                      0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    uint8_t buf[128];
    size_t fill = 0;
    const uint8_t* parts[3] = {a, b, c};
    const size_t lens[3] = {al, bl, cl};
    for (int p = 0; p < 3; ++p)
        for (size_t i = 0; i < lens[p];) {
            const size_t take = lens[p] - i < 128 - fill ? lens[p] - i : 128 - fill;
            memcpy(buf + fill, parts[p] + i, take);
            fill += take; i += take;
            if (fill == 128) { sha512_compress(st, buf); fill = 0; }
        }
    const uint64_t bits = (uint64_t)(al + bl + cl) * 8;
    buf[fill++] = 0x80;
    if (fill > 112) { memset(buf + fill, 0, 128 - fill); sha512_compress(st, buf); fill = 0; }
    memset(buf + fill, 0, 128 - fill);
    for (int i = 0; i < 8; ++i) buf[127 - i] = (uint8_t)(bits >> (8 * i));
    sha512_compress(st, buf);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(st[i] >> (56 - 8 * j));
}
// x (64 bytes little-endian) mod L -> 32 bytes little-endian (edwards25519 Scalar.SetUniformBytes)
void reduce_mod_l(const uint8_t in[64], uint8_t out[32]) {
    static const uint64_t Lm[4] = {0x5812631a5cf5d3edull, 0x14def9dea2f79cd6ull, 0, 0x1000000000000000ull};
    uint64_t acc[5] = {0, 0, 0, 0, 0};
    for (int bit = 511; bit >= 0; --bit) {
        uint64_t carry = (in[bit >> 3] >> (bit & 7)) & 1;
        for (int i = 0; i < 5; ++i) { const uint64_t nc = acc[i] >> 63; acc[i] = (acc[i] << 1) | carry; carry = nc; }
        uint64_t d[5], borrow = 0;
        for (int i = 0; i < 5; ++i) {
            const unsigned __int128 x = (unsigned __int128)acc[i] - (i < 4 ? Lm[i] : 0) - borrow;
            d[i] = (uint64_t)x; borrow = (uint64_t)(x >> 64) & 1;
        }
        if (!borrow) memcpy(acc, d, sizeof d);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(acc[i] >> (8 * j));
}

}  // namespace

extern "C" int sbv_ed25519_make_tuples(const uint8_t* sigs, const uint8_t* pks, const uint8_t* msgs, const uint64_t* offsets,
                                       size_t n, uint8_t* tuples_out) {
    if (n == 0) return SBV_OK;
    if (!sigs || !pks || !offsets || !tuples_out) return SBV_EINVAL;
    static const uint8_t empty = 0;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return SBV_EINVAL;
        uint8_t* t = tuples_out + 128 * i;
        memcpy(t, sigs + 64 * i, 64);
        memcpy(t + 64, pks + 32 * i, 32);
        uint8_t h[64];
        const size_t ml = (size_t)(offsets[i + 1] - offsets[i]);
        sha512_3(sigs + 64 * i, 32, pks + 32 * i, 32, ml ? msgs + offsets[i] : &empty, ml, h);
        reduce_mod_l(h, t + 96);
    }
    return SBV_OK;
}

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

extern "C" int sbv_sha256_uses_cpu_extensions(void) { return g_sha_ext ? 1 : 0; }

extern "C" int sbv_sha256_batch(const uint8_t* msgs, const uint64_t* offsets, size_t n, uint8_t* out_hashes) {
    if (n == 0) return SBV_OK;
    if (!msgs || !offsets || !out_hashes) return SBV_EINVAL;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return SBV_EINVAL;
        sha256(msgs + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), out_hashes + 32 * i);
    }
    return SBV_OK;
}
