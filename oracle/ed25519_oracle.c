/*
 * ed25519_oracle.c — CPU restatement of Ed25519 verification with Go crypto/ed25519 semantics
 * (BASELINE.json configs[4]: "Ed25519 Verifier variant").
 *
 * TEST INFRASTRUCTURE ONLY (same rules as oracle/p256_oracle.c): the checker for the HIP path;
 * nothing in the product links or calls it.
 *
 * PARITY STATUS: "parity unpinned" by the reference — SmartBFT-Go/consensus holds no signature
 * code or vectors.  The algorithm lives in Go's standard library
 * (crypto/ed25519.verify, crypto/internal/edwards25519 = filippo.io/edwards25519; toolchain
 * go 1.20, go.mod:3), absent here.  Restated from the published algorithm (RFC 8032 §5.1.7 with
 * Go's specific choices) and pinned on RFC 8032 §7.1 tests 1-3, the Python twin
 * (oracle/ed25519_py.py) and OpenSSL's Ed25519 on honest / bit-flipped signatures.
 *
 * Go rules followed:
 *   len(sig) == 64 and sig[63] & 0xE0 == 0, else false.
 *   A = Point.SetBytes(pk): y = low 255 bits reduced mod p (non-canonical accepted);
 *       x = SqrtRatio(y^2 - 1, d y^2 + 1), not square -> false; sign bit selects -x; "x = 0 with
 *       the sign bit set" is accepted.
 *   k = SHA-512(sig[:32] || pk || msg) mod L;  S = sig[32:] must be < L.
 *   R' = [S]B + [k](-A), cofactorless; accept iff encode(R') == sig[:32] byte-wise.
 *
 * Arithmetic: GF(2^255-19) in 4 x 64-bit limbs (weakly reduced < 2^256, frozen for compare),
 * extended twisted-Edwards coordinates, plain double-and-add — nothing shared with the device code.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;

static const fe FE_D = {{0x75eb4dca135978a3ull, 0x00700a4d4141d8abull, 0x8cc740797779e898ull, 0x52036cee2b6ffe73ull}};
static const fe FE_SQRTM1 = {{0xc4ee1b274a0ea0b0ull, 0x2f431806ad2fe478ull, 0x2b4d00993dfbd7a7ull, 0x2b8324804fc1df0bull}};
static const fe FE_ONE = {{1, 0, 0, 0}};
static const fe FE_ZERO = {{0, 0, 0, 0}};
static const fe FE_BX = {{0xc9562d608f25d51aull, 0x692cc7609525a7b2ull, 0xc0a4e231fdd6dc5cull, 0x216936d3cd6e53feull}};
static const fe FE_BY = {{0x6666666666666658ull, 0x6666666666666666ull, 0x6666666666666666ull, 0x6666666666666666ull}};
/* L = 2^252 + 27742317777372353535851937790883648493 */
static const uint64_t L_LIMBS[4] = {0x5812631a5cf5d3edull, 0x14def9dea2f79cd6ull, 0, 0x1000000000000000ull};

static void fe_add(fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a->v[i] + b->v[i]; r->v[i] = (uint64_t)c; c >>= 64; }
    /* fold 2^256 = 38 */
    u128 t = (u128)r->v[0] + (uint64_t)c * 38;
    r->v[0] = (uint64_t)t; t >>= 64;
    for (int i = 1; i < 4 && t; ++i) { t += r->v[i]; r->v[i] = (uint64_t)t; t >>= 64; }
    if (t) r->v[0] += 38;   /* cannot carry again */
}
static void fe_sub(fe *r, const fe *a, const fe *b) {
    /* a - b + 4p - (4p) handled by adding 2*(2^256 - 38) = multiple of p... use: a + (2^256-38)*2 - b with folding */
    uint64_t borrow = 0;
    fe t;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->v[i] - b->v[i] - borrow;
        t.v[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    /* borrow means we wrapped by 2^256 = 38 (mod p): subtract 38 */
    uint64_t sub = borrow * 38;
    borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)t.v[i] - (i == 0 ? sub : 0) - borrow;
        t.v[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) t.v[0] -= 38;   /* wrapped once more; result stays < 2^256 */
    *r = t;
}
static void fe_mul(fe *r, const fe *a, const fe *b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->v[i] * b->v[j] + t[i + j];
            t[i + j] = (uint64_t)c; c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)t[i + 4] * 38 + t[i]; r->v[i] = (uint64_t)c; c >>= 64; }
    u128 f = (u128)r->v[0] + (uint64_t)c * 38;
    r->v[0] = (uint64_t)f; f >>= 64;
    for (int i = 1; i < 4 && f; ++i) { f += r->v[i]; r->v[i] = (uint64_t)f; f >>= 64; }
    if (f) r->v[0] += 38;
}
static void fe_sqr(fe *r, const fe *a) { fe_mul(r, a, a); }
/* canonical representative in [0, p) */
static void fe_freeze(fe *r, const fe *a) {
    fe t = *a;
    /* reduce bit 255: t = (t mod 2^255) + 19 * (t >> 255), twice */
    for (int k = 0; k < 2; ++k) {
        uint64_t top = t.v[3] >> 63;
        t.v[3] &= 0x7fffffffffffffffull;
        u128 c = (u128)t.v[0] + top * 19;
        t.v[0] = (uint64_t)c; c >>= 64;
        for (int i = 1; i < 4; ++i) { c += t.v[i]; t.v[i] = (uint64_t)c; c >>= 64; }
    }
    /* now t < 2^255 + small; subtract p if t >= p */
    static const uint64_t Pm[4] = {0xffffffffffffffedull, 0xffffffffffffffffull, 0xffffffffffffffffull, 0x7fffffffffffffffull};
    fe d; uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 x = (u128)t.v[i] - Pm[i] - borrow;
        d.v[i] = (uint64_t)x; borrow = (uint64_t)(x >> 64) & 1;
    }
    *r = borrow ? t : d;
}
static int fe_eq(const fe *a, const fe *b) {
    fe x, y; fe_freeze(&x, a); fe_freeze(&y, b);
    return ((x.v[0] ^ y.v[0]) | (x.v[1] ^ y.v[1]) | (x.v[2] ^ y.v[2]) | (x.v[3] ^ y.v[3])) == 0;
}
static void fe_neg(fe *r, const fe *a) { fe_sub(r, &FE_ZERO, a); }
static int fe_is_negative(const fe *a) { fe t; fe_freeze(&t, a); return (int)(t.v[0] & 1); }
static void fe_pow_p58(fe *r, const fe *a) {   /* a^((p-5)/8) = a^(2^252 - 3) */
    fe acc = FE_ONE;
    /* exponent bits: 2^252 - 3 = 250 ones, then 0, 1 */
    for (int i = 251; i >= 0; --i) {
        fe_sqr(&acc, &acc);
        int bit = (i >= 2) ? 1 : (i == 1 ? 0 : 1);
        if (bit) fe_mul(&acc, &acc, a);
    }
    *r = acc;
}
static void fe_inv(fe *r, const fe *a) {       /* a^(p-2), p-2 = 2^255 - 21 */
    fe acc = FE_ONE;
    /* bits of 2^255-21: 250 ones then 0 1 0 1 1  (…11101011) */
    static const int low5[5] = {0, 1, 0, 1, 1};
    for (int i = 254; i >= 0; --i) {
        fe_sqr(&acc, &acc);
        int bit = (i >= 5) ? 1 : low5[4 - i];
        if (bit) fe_mul(&acc, &acc, a);
    }
    *r = acc;
}
static void fe_from_le(fe *r, const uint8_t b[32]) {
    for (int i = 0; i < 4; ++i) {
        uint64_t w = 0;
        for (int j = 7; j >= 0; --j) w = (w << 8) | b[8 * i + j];
        r->v[i] = w;
    }
}
static void fe_to_le(uint8_t b[32], const fe *a) {
    fe t; fe_freeze(&t, a);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(t.v[i] >> (8 * j));
}

typedef struct { fe X, Y, Z, T; } ept;

static void pt_add(ept *r, const ept *p, const ept *q) {
    fe a, b, c, d, e, f, g, h, t1, t2;
    fe_sub(&t1, &p->Y, &p->X); fe_sub(&t2, &q->Y, &q->X); fe_mul(&a, &t1, &t2);
    fe_add(&t1, &p->Y, &p->X); fe_add(&t2, &q->Y, &q->X); fe_mul(&b, &t1, &t2);
    fe_mul(&c, &p->T, &q->T); fe_mul(&c, &c, &FE_D); fe_add(&c, &c, &c);
    fe_mul(&d, &p->Z, &q->Z); fe_add(&d, &d, &d);
    fe_sub(&e, &b, &a); fe_sub(&f, &d, &c); fe_add(&g, &d, &c); fe_add(&h, &b, &a);
    fe_mul(&r->X, &e, &f); fe_mul(&r->Y, &g, &h); fe_mul(&r->Z, &f, &g); fe_mul(&r->T, &e, &h);
}
static void pt_ident(ept *r) { r->X = FE_ZERO; r->Y = FE_ONE; r->Z = FE_ONE; r->T = FE_ZERO; }
static void pt_mul(ept *r, const uint8_t k_le[32], const ept *p) {
    ept acc; pt_ident(&acc);
    for (int i = 255; i >= 0; --i) {
        pt_add(&acc, &acc, &acc);
        if ((k_le[i >> 3] >> (i & 7)) & 1) pt_add(&acc, &acc, p);
    }
    *r = acc;
}
static void pt_encode(uint8_t out[32], const ept *p) {
    fe zi, x, y;
    fe_inv(&zi, &p->Z); fe_mul(&x, &p->X, &zi); fe_mul(&y, &p->Y, &zi);
    fe_to_le(out, &y);
    out[31] |= (uint8_t)(fe_is_negative(&x) << 7);
}
/* edwards25519.Point.SetBytes */
static int pt_decompress(ept *r, const uint8_t b[32]) {
    uint8_t yb[32]; memcpy(yb, b, 32); yb[31] &= 0x7f;
    fe y, y2, u, v, v3, v7, t, rr, check, nu, nui;
    fe_from_le(&y, yb);
    fe_sqr(&y2, &y);
    fe_sub(&u, &y2, &FE_ONE);
    fe_mul(&v, &y2, &FE_D); fe_add(&v, &v, &FE_ONE);
    fe_sqr(&t, &v); fe_mul(&v3, &t, &v);
    fe_sqr(&t, &v3); fe_mul(&v7, &t, &v);
    fe_mul(&t, &u, &v7); fe_pow_p58(&t, &t);
    fe_mul(&rr, &u, &v3); fe_mul(&rr, &rr, &t);
    fe_sqr(&t, &rr); fe_mul(&check, &v, &t);
    fe_neg(&nu, &u); fe_mul(&nui, &nu, &FE_SQRTM1);
    int correct = fe_eq(&check, &u), flipped = fe_eq(&check, &nu), flipped_i = fe_eq(&check, &nui);
    if (flipped || flipped_i) fe_mul(&rr, &rr, &FE_SQRTM1);
    if (fe_is_negative(&rr)) fe_neg(&rr, &rr);
    if (!(correct || flipped)) return 0;
    if (b[31] >> 7) fe_neg(&rr, &rr);
    r->X = rr; r->Y = y; r->Z = FE_ONE; fe_mul(&r->T, &rr, &y);
    return 1;
}

/* ---- SHA-512 (FIPS 180-4) ----------------------------------------------------------------------- */
static const uint64_t K512[80] = {
    0x428a2f98d728ae22ull,0x7137449123ef65cdull,0xb5c0fbcfec4d3b2full,0xe9b5dba58189dbbcull,0x3956c25bf348b538ull,0x59f111f1b605d019ull,0x923f82a4af194f9bull,0xab1c5ed5da6d8118ull,
    0xd807aa98a3030242ull,0x12835b0145706fbeull,0x243185be4ee4b28cull,0x550c7dc3d5ffb4e2ull,0x72be5d74f27b896full,0x80deb1fe3b1696b1ull,0x9bdc06a725c71235ull,0xc19bf174cf692694ull,
    0xe49b69c19ef14ad2ull,0xefbe4786384f25e3ull,0x0fc19dc68b8cd5b5ull,0x240ca1cc77ac9c65ull,0x2de92c6f592b0275ull,0x4a7484aa6ea6e483ull,0x5cb0a9dcbd41fbd4ull,0x76f988da831153b5ull,
    0x983e5152ee66dfabull,0xa831c66d2db43210ull,0xb00327c898fb213full,0xbf597fc7beef0ee4ull,0xc6e00bf33da88fc2ull,0xd5a79147930aa725ull,0x06ca6351e003826full,0x142929670a0e6e70ull,
    0x27b70a8546d22ffcull,0x2e1b21385c26c926ull,0x4d2c6dfc5ac42aedull,0x53380d139d95b3dfull,0x650a73548baf63deull,0x766a0abb3c77b2a8ull,0x81c2c92e47edaee6ull,0x92722c851482353bull,
    0xa2bfe8a14cf10364ull,0xa81a664bbc423001ull,0xc24b8b70d0f89791ull,0xc76c51a30654be30ull,0xd192e819d6ef5218ull,0xd69906245565a910ull,0xf40e35855771202aull,0x106aa07032bbd1b8ull,
    0x19a4c116b8d2d0c8ull,0x1e376c085141ab53ull,0x2748774cdf8eeb99ull,0x34b0bcb5e19b48a8ull,0x391c0cb3c5c95a63ull,0x4ed8aa4ae3418acbull,0x5b9cca4f7763e373ull,0x682e6ff3d6b2b8a3ull,
    0x748f82ee5defb2fcull,0x78a5636f43172f60ull,0x84c87814a1f0ab72ull,0x8cc702081a6439ecull,0x90befffa23631e28ull,0xa4506cebde82bde9ull,0xbef9a3f7b2c67915ull,0xc67178f2e372532bull,
    0xca273eceea26619cull,0xd186b8c721c0c207ull,0xeada7dd6cde0eb1eull,0xf57d4f7fee6ed178ull,0x06f067aa72176fbaull,0x0a637dc5a2c898a6ull,0x113f9804bef90daeull,0x1b710b35131c471bull,
    0x28db77f523047d84ull,0x32caab7b40c72493ull,0x3c9ebe0a15c9bebcull,0x431d67c49c100d4cull,0x4cc5d4becb3e42b6ull,0x597f299cfc657e2aull,0x5fcb6fab3ad6faecull,0x6c44198c4a475817ull};
#define ROR64(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
static void sha512_block(uint64_t h[8], const uint8_t blk[128]) {
    uint64_t w[80];
    for (int i = 0; i < 16; ++i) { uint64_t x = 0; for (int j = 0; j < 8; ++j) x = (x << 8) | blk[8 * i + j]; w[i] = x; }
    for (int i = 16; i < 80; ++i) {
        uint64_t s0 = ROR64(w[i-15], 1) ^ ROR64(w[i-15], 8) ^ (w[i-15] >> 7);
        uint64_t s1 = ROR64(w[i-2], 19) ^ ROR64(w[i-2], 61) ^ (w[i-2] >> 6);
        w[i] = w[i-16] + s0 + w[i-7] + s1;
    }
    uint64_t a=h[0],b=h[1],c=h[2],d=h[3],e=h[4],f=h[5],g=h[6],hh=h[7];
    for (int i = 0; i < 80; ++i) {
        uint64_t S1 = ROR64(e,14) ^ ROR64(e,18) ^ ROR64(e,41), ch = (e & f) ^ (~e & g);
        uint64_t t1 = hh + S1 + ch + K512[i] + w[i];
        uint64_t S0 = ROR64(a,28) ^ ROR64(a,34) ^ ROR64(a,39), mj = (a & b) ^ (a & c) ^ (b & c);
        uint64_t t2 = S0 + mj;
        hh=g; g=f; f=e; e=d+t1; d=c; c=b; b=a; a=t1+t2;
    }
    h[0]+=a;h[1]+=b;h[2]+=c;h[3]+=d;h[4]+=e;h[5]+=f;h[6]+=g;h[7]+=hh;
}
void sbvo_sha512(const uint8_t *msg, size_t len, uint8_t out[64]) {
    uint64_t h[8] = {0x6a09e667f3bcc908ull,0xbb67ae8584caa73bull,0x3c6ef372fe94f82bull,0xa54ff53a5f1d36f1ull,
                     0x510e527fade682d1ull,0x9b05688c2b3e6c1full,0x1f83d9abfb41bd6bull,0x5be0cd19137e2179ull};
    size_t i = 0;
    for (; i + 128 <= len; i += 128) sha512_block(h, msg + i);
    uint8_t tail[256]; size_t rem = len - i;
    memset(tail, 0, sizeof tail); memcpy(tail, msg + i, rem);
    tail[rem] = 0x80;
    size_t tl = (rem < 112) ? 128 : 256;
    uint64_t bits = (uint64_t)len * 8;
    for (int j = 0; j < 8; ++j) tail[tl - 1 - j] = (uint8_t)(bits >> (8 * j));
    sha512_block(h, tail); if (tl == 256) sha512_block(h, tail + 128);
    for (int j = 0; j < 8; ++j) for (int k = 0; k < 8; ++k) out[8 * j + k] = (uint8_t)(h[j] >> (56 - 8 * k));
}

/* ---- scalars mod L --------------------------------------------------------------------------------- */
/* r (32 bytes LE) = x (64 bytes LE) mod L, by binary long reduction (simple, not fast) */
static void sc_reduce512(uint8_t out[32], const uint8_t in[64]) {
    uint64_t acc[5] = {0, 0, 0, 0, 0};   /* acc < 2L at every step */
    for (int bit = 511; bit >= 0; --bit) {
        /* acc = 2*acc + bit */
        uint64_t carry = (in[bit >> 3] >> (bit & 7)) & 1;
        for (int i = 0; i < 5; ++i) { uint64_t nc = acc[i] >> 63; acc[i] = (acc[i] << 1) | carry; carry = nc; }
        /* if acc >= L: acc -= L */
        uint64_t d[5]; uint64_t borrow = 0;
        for (int i = 0; i < 5; ++i) {
            u128 x = (u128)acc[i] - (i < 4 ? L_LIMBS[i] : 0) - borrow;
            d[i] = (uint64_t)x; borrow = (uint64_t)(x >> 64) & 1;
        }
        if (!borrow) memcpy(acc, d, sizeof d);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(acc[i] >> (8 * j));
}
static int sc_is_canonical(const uint8_t s[32]) {
    for (int i = 3; i >= 0; --i) {
        uint64_t w = 0;
        for (int j = 7; j >= 0; --j) w = (w << 8) | s[8 * i + j];
        if (w < L_LIMBS[i]) return 1;
        if (w > L_LIMBS[i]) return 0;
    }
    return 0;
}
/* r = (a + b*c) mod L, all 32-byte LE, via 512-bit accumulate + reduce */
static void sc_muladd(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]) {
    uint64_t B[4], C[4], A[4], t[8] = {0};
    for (int i = 0; i < 4; ++i) { B[i] = C[i] = A[i] = 0; for (int j = 7; j >= 0; --j) { B[i] = (B[i] << 8) | b[8*i+j]; C[i] = (C[i] << 8) | c[8*i+j]; A[i] = (A[i] << 8) | a[8*i+j]; } }
    for (int i = 0; i < 4; ++i) {
        u128 carry = 0;
        for (int j = 0; j < 4; ++j) { carry += (u128)B[i] * C[j] + t[i + j]; t[i + j] = (uint64_t)carry; carry >>= 64; }
        t[i + 4] = (uint64_t)carry;
    }
    u128 carry = 0;
    for (int i = 0; i < 8; ++i) { carry += (u128)t[i] + (i < 4 ? A[i] : 0); t[i] = (uint64_t)carry; carry >>= 64; }
    uint8_t wide[64];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) wide[8 * i + j] = (uint8_t)(t[i] >> (8 * j));
    sc_reduce512(out, wide);
}

/* ---- verification ----------------------------------------------------------------------------------- */
/* tuple = sig(64) | pk(32) | k(32 LE, reduced mod L by whoever hashed; k >= L rejects) */
int sbvo_ed25519_verify_tuple(const uint8_t t[128]) {
    const uint8_t *sig = t, *pk = t + 64;
    if (sig[63] & 0xE0) return 0;
    if (!sc_is_canonical(sig + 32) || !sc_is_canonical(t + 96)) return 0;
    ept A, nA, B, sB, kA, R;
    if (!pt_decompress(&A, pk)) return 0;
    fe_neg(&nA.X, &A.X); nA.Y = A.Y; nA.Z = A.Z; fe_neg(&nA.T, &A.T);
    const uint8_t *k = t + 96;
    B.X = FE_BX; B.Y = FE_BY; B.Z = FE_ONE; fe_mul(&B.T, &FE_BX, &FE_BY);
    pt_mul(&sB, sig + 32, &B);
    pt_mul(&kA, k, &nA);
    pt_add(&R, &sB, &kA);
    uint8_t enc[32];
    pt_encode(enc, &R);
    return memcmp(enc, sig, 32) == 0;
}
static void hram(uint8_t k[32], const uint8_t r[32], const uint8_t pk[32], const uint8_t *msg, size_t len) {
    uint8_t *buf = (uint8_t *)malloc(64 + len), h[64];
    memcpy(buf, r, 32); memcpy(buf + 32, pk, 32); if (len) memcpy(buf + 64, msg, len);
    sbvo_sha512(buf, 64 + len, h);
    free(buf);
    sc_reduce512(k, h);
}
int sbvo_ed25519_verify(const uint8_t pk[32], const uint8_t *msg, size_t len, const uint8_t *sig, size_t siglen) {
    if (siglen != 64) return 0;
    uint8_t t[128];
    memcpy(t, sig, 64); memcpy(t + 64, pk, 32);
    hram(t + 96, sig, pk, msg, len);
    return sbvo_ed25519_verify_tuple(t);
}
void sbvo_ed25519_make_tuple(const uint8_t pk[32], const uint8_t *msg, size_t len, const uint8_t sig[64], uint8_t t[128]) {
    memcpy(t, sig, 64); memcpy(t + 64, pk, 32);
    hram(t + 96, sig, pk, msg, len);
}

/* ---- signing (RFC 8032 §5.1.6) for synthetic data ----------------------------------------------------- */
static void expand(const uint8_t seed[32], uint8_t a[32], uint8_t prefix[32]) {
    uint8_t h[64]; sbvo_sha512(seed, 32, h);
    memcpy(a, h, 32); memcpy(prefix, h + 32, 32);
    a[0] &= 248; a[31] &= 127; a[31] |= 64;
}
void sbvo_ed25519_public_key(const uint8_t seed[32], uint8_t pk[32]) {
    uint8_t a[32], prefix[32]; expand(seed, a, prefix);
    ept B, P; B.X = FE_BX; B.Y = FE_BY; B.Z = FE_ONE; fe_mul(&B.T, &FE_BX, &FE_BY);
    pt_mul(&P, a, &B); pt_encode(pk, &P);
}
void sbvo_ed25519_sign(const uint8_t seed[32], const uint8_t *msg, size_t len, uint8_t sig[64]) {
    uint8_t a[32], prefix[32], pk[32], r[32], k[32], h[64];
    expand(seed, a, prefix);
    ept B, P; B.X = FE_BX; B.Y = FE_BY; B.Z = FE_ONE; fe_mul(&B.T, &FE_BX, &FE_BY);
    pt_mul(&P, a, &B); pt_encode(pk, &P);
    uint8_t *buf = (uint8_t *)malloc(32 + len);
    memcpy(buf, prefix, 32); if (len) memcpy(buf + 32, msg, len);
    sbvo_sha512(buf, 32 + len, h); free(buf);
    sc_reduce512(r, h);
    pt_mul(&P, r, &B); pt_encode(sig, &P);
    hram(k, sig, pk, msg, len);
    sc_muladd(sig + 32, r, k, a);
}

/* ---- synthetic batch: nkeys keys, tuple i = signature by key i % nkeys over a 32-byte message;
 * every invalid_every-th tuple has one bit flipped somewhere in sig|pk (k is recomputed from the
 * flipped bytes, as a verifier would).  expect = oracle verdicts. -------------------------------------- */
typedef struct { uint32_t seed; size_t lo, hi, nkeys; unsigned inv; const uint8_t *seeds, *pks; uint8_t *tuples, *expect; } egen;
static void *egen_worker(void *arg) {
    egen *j = (egen *)arg;
    for (size_t i = j->lo; i < j->hi; ++i) {
        uint8_t msg[32], sig[64];
        memset(msg, 0, 32); memcpy(msg, "sbv-ed-msg", 10);
        msg[12] = (uint8_t)(j->seed >> 24); msg[13] = (uint8_t)(j->seed >> 16); msg[14] = (uint8_t)(j->seed >> 8); msg[15] = (uint8_t)j->seed;
        for (int b = 0; b < 8; ++b) msg[24 + b] = (uint8_t)((uint64_t)i >> (56 - 8 * b));
        size_t key = i % j->nkeys;
        sbvo_ed25519_sign(j->seeds + 32 * key, msg, 32, sig);
        uint8_t pk[32]; memcpy(pk, j->pks + 32 * key, 32);
        int valid = 1;
        if (j->inv && (i % j->inv) == j->inv - 1) {
            uint8_t lbl[24], sel[64]; memcpy(lbl, "sbv-ed-flip", 11);
            for (int b = 0; b < 8; ++b) lbl[11 + b] = (uint8_t)((uint64_t)i >> (56 - 8 * b));
            lbl[19] = (uint8_t)(j->seed >> 24); lbl[20] = (uint8_t)(j->seed >> 16); lbl[21] = (uint8_t)(j->seed >> 8); lbl[22] = (uint8_t)j->seed;
            sbvo_sha512(lbl, 23, sel);
            unsigned bit = (((unsigned)sel[0] << 8) | sel[1]) % 768u;      /* 96 bytes: sig | pk */
            if (bit < 512) sig[bit >> 3] ^= (uint8_t)(1u << (bit & 7)); else pk[(bit - 512) >> 3] ^= (uint8_t)(1u << (bit & 7));
            valid = -1;
        }
        uint8_t *t = j->tuples + 128 * i;
        sbvo_ed25519_make_tuple(pk, msg, 32, sig, t);
        if (valid < 0) valid = sbvo_ed25519_verify_tuple(t);
        if (j->expect && valid) __atomic_fetch_or(&j->expect[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
    }
    return NULL;
}
void sbvo_ed25519_gen_batch(uint32_t seed, size_t n, size_t nkeys, unsigned invalid_every, uint8_t *tuples, uint8_t *expect, int threads) {
    uint8_t *seeds = (uint8_t *)malloc(32 * nkeys), *pks = (uint8_t *)malloc(32 * nkeys);
    for (size_t i = 0; i < nkeys; ++i) {
        uint8_t lbl[24], h[64]; memcpy(lbl, "sbv-ed-key", 10);
        lbl[10] = (uint8_t)(seed >> 24); lbl[11] = (uint8_t)(seed >> 16); lbl[12] = (uint8_t)(seed >> 8); lbl[13] = (uint8_t)seed;
        for (int b = 0; b < 8; ++b) lbl[14 + b] = (uint8_t)((uint64_t)i >> (56 - 8 * b));
        sbvo_sha512(lbl, 22, h); memcpy(seeds + 32 * i, h, 32);
        sbvo_ed25519_public_key(seeds + 32 * i, pks + 32 * i);
    }
    if (expect) memset(expect, 0, (n + 7) / 8);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; egen jobs[256];
    size_t per = ((n + threads - 1) / threads + 7) & ~(size_t)7;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo >= n) break; if (hi > n) hi = n;
        jobs[t] = (egen){seed, lo, hi, nkeys, invalid_every, seeds, pks, tuples, expect};
        pthread_create(&th[t], NULL, egen_worker, &jobs[t]); ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(seeds); free(pks);
}

typedef struct { const uint8_t *tuples; size_t lo, hi; uint8_t *bitmap; } ever;
static void *ever_worker(void *arg) {
    ever *j = (ever *)arg;
    for (size_t i = j->lo; i < j->hi; ++i)
        if (sbvo_ed25519_verify_tuple(j->tuples + 128 * i)) j->bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
    return NULL;
}
void sbvo_ed25519_verify_batch(const uint8_t *tuples, size_t n, uint8_t *bitmap, int threads) {
    memset(bitmap, 0, (n + 7) / 8);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; ever jobs[256];
    size_t per = ((n + threads - 1) / threads + 7) & ~(size_t)7;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo >= n) break; if (hi > n) hi = n;
        jobs[t] = (ever){tuples, lo, hi, bitmap};
        pthread_create(&th[t], NULL, ever_worker, &jobs[t]); ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
}
