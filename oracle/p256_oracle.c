/*
 * p256_oracle.c — CPU restatement of ECDSA P-256 verification with Go crypto/ecdsa semantics.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path in
 * consensus_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may link, load or execute anything under oracle/.  The product library (libsbv.so)
 * never calls into it and has no CPU fallback.
 *
 * PARITY STATUS: "parity unpinned" by the reference.  SmartBFT-Go/consensus contains no
 * signature arithmetic: api.Verifier (pkg/api/dependencies.go:54-71) is a plugin seam and
 * every Verifier in the tree is a no-op or mock (examples/naive_chain/node.go:86-96,
 * test/test_app.go:231-248, internal/bft/mocks/verifier_mock.go).  The algorithm the north
 * star names as ground truth is Go's standard library — crypto/ecdsa (verifyNISTEC,
 * parseSignature, hashToNat), crypto/internal/nistec (P256 point arithmetic),
 * crypto/internal/bigmod, golang.org/x/crypto/cryptobyte (vendored in std) — pinned only
 * by the toolchain (go.mod:3 "go 1.20"; CI 1.21.8, .github/workflows/build.yml:17).  It is
 * absent from /root/reference and cannot run here (no Go toolchain).  This file restates
 * the published algorithm (SEC 1 v2.0 §4.1.4 + the Go-specific input rules below) and is
 * pinned against RFC 6979 A.2.5 known-answer signatures (tests/golden/rfc6979_p256.json),
 * a Python big-int twin (oracle/p256_py.py) and OpenSSL 3.0 ECDSA_do_verify
 * (oracle/openssl_check.c) on every mathematically defined vector class.
 *
 * Go >= 1.20 rules followed (function names are Go's, for a maintainer to cross-read):
 *   parseSignature     : one DER SEQUENCE, no trailing bytes, exactly two INTEGERs, each
 *                        minimal and non-negative (cryptobyte ReadASN1Integer), zero-stripped.
 *   bigmod SetBytes    : r, s longer than 32 bytes or >= N  -> reject; zero -> reject.
 *   pointFromAffine    : Qx,Qy non-negative, <= 256 bits, < p, on curve ((0,0) is off curve).
 *   hashToNat          : leftmost 32 bytes of the hash, big-endian, reduced mod N; e = 0 allowed.
 *   verifyNISTEC       : w = s^-1; u1 = e*w; u2 = r*w; R = u1*G + u2*Q (exact group law,
 *                        incl. u1*G == +-u2*Q); R = infinity -> reject; accept iff R.x mod N == r.
 *   No low-S rule.
 *
 * Implementation is intentionally different from the device code (4x64-bit limbs,
 * generic CIOS Montgomery with unsigned __int128, unsigned 4-bit windows, separate scalar
 * multiplications) so that bugs do not correlate.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } u256;

typedef struct {
    u256 m;       /* modulus */
    u256 r2;      /* R^2 mod m, R = 2^256 */
    u256 one;     /* R mod m */
    uint64_t n0;  /* -m^-1 mod 2^64 */
} mctx;

static const mctx FP = {
    {{0xFFFFFFFFFFFFFFFFull, 0x00000000FFFFFFFFull, 0x0000000000000000ull, 0xFFFFFFFF00000001ull}},
    {{0x0000000000000003ull, 0xFFFFFFFBFFFFFFFFull, 0xFFFFFFFFFFFFFFFEull, 0x00000004FFFFFFFDull}},
    {{0x0000000000000001ull, 0xFFFFFFFF00000000ull, 0xFFFFFFFFFFFFFFFFull, 0x00000000FFFFFFFEull}},
    0x0000000000000001ull};
static const mctx FN = {
    {{0xF3B9CAC2FC632551ull, 0xBCE6FAADA7179E84ull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFF00000000ull}},
    {{0x83244C95BE79EEA2ull, 0x4699799C49BD6FA6ull, 0x2845B2392B6BEC59ull, 0x66E12D94F3D95620ull}},
    {{0x0C46353D039CDAAFull, 0x4319055258E8617Bull, 0x0000000000000000ull, 0x00000000FFFFFFFFull}},
    0xCCD1C8AAEE00BC4Full};

static const u256 CURVE_B = {{0x3BCE3C3E27D2604Bull, 0x651D06B0CC53B0F6ull, 0xB3EBBD55769886BCull, 0x5AC635D8AA3A93E7ull}};
static const u256 GX = {{0xF4A13945D898C296ull, 0x77037D812DEB33A0ull, 0xF8BCE6E563A440F2ull, 0x6B17D1F2E12C4247ull}};
static const u256 GY = {{0xCBB6406837BF51F5ull, 0x2BCE33576B315ECEull, 0x8EE7EB4A7C0F9E16ull, 0x4FE342E2FE1A7F9Bull}};

/* ---- 256-bit helpers ---------------------------------------------------------------- */
static int u256_is_zero(const u256 *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int u256_eq(const u256 *a, const u256 *b) {
    return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0;
}
/* a >= b ? */
static int u256_geq(const u256 *a, const u256 *b) {
    for (int i = 3; i >= 0; --i) {
        if (a->v[i] > b->v[i]) return 1;
        if (a->v[i] < b->v[i]) return 0;
    }
    return 1;
}
static uint64_t u256_add(u256 *r, const u256 *a, const u256 *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a->v[i] + b->v[i]; r->v[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t u256_sub(u256 *r, const u256 *a, const u256 *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->v[i] - b->v[i] - borrow;
        r->v[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static void u256_from_be(u256 *r, const uint8_t b[32]) {
    for (int i = 0; i < 4; ++i) {
        uint64_t w = 0;
        for (int j = 0; j < 8; ++j) w = (w << 8) | b[(3 - i) * 8 + j];
        r->v[i] = w;
    }
}
static void u256_to_be(uint8_t b[32], const u256 *a) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) b[(3 - i) * 8 + j] = (uint8_t)(a->v[i] >> (56 - 8 * j));
}

/* ---- modular arithmetic (values < m) ------------------------------------------------- */
static void mod_add(u256 *r, const u256 *a, const u256 *b, const mctx *c) {
    u256 t; uint64_t carry = u256_add(&t, a, b);
    if (carry || u256_geq(&t, &c->m)) u256_sub(&t, &t, &c->m);
    *r = t;
}
static void mod_sub(u256 *r, const u256 *a, const u256 *b, const mctx *c) {
    u256 t; if (u256_sub(&t, a, b)) u256_add(&t, &t, &c->m);
    *r = t;
}
/* Montgomery product a*b*R^-1 mod m (CIOS, Koc et al.) */
static void mont_mul(u256 *r, const u256 *a, const u256 *b, const mctx *c) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 acc = 0;
        for (int j = 0; j < 4; ++j) {
            acc += (u128)a->v[j] * b->v[i] + t[j];
            t[j] = (uint64_t)acc; acc >>= 64;
        }
        acc += t[4]; t[4] = (uint64_t)acc; t[5] = (uint64_t)(acc >> 64);
        uint64_t m = t[0] * c->n0;
        acc = (u128)m * c->m.v[0] + t[0];
        acc >>= 64;
        for (int j = 1; j < 4; ++j) {
            acc += (u128)m * c->m.v[j] + t[j];
            t[j - 1] = (uint64_t)acc; acc >>= 64;
        }
        acc += t[4]; t[3] = (uint64_t)acc; t[4] = t[5] + (uint64_t)(acc >> 64);
    }
    u256 res = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || u256_geq(&res, &c->m)) u256_sub(&res, &res, &c->m);
    *r = res;
}
static void to_mont(u256 *r, const u256 *a, const mctx *c) { mont_mul(r, a, &c->r2, c); }
static void from_mont(u256 *r, const u256 *a, const mctx *c) {
    u256 one = {{1, 0, 0, 0}}; mont_mul(r, a, &one, c);
}
/* a^(m-2) in the Montgomery domain (Fermat), plain square-and-multiply */
static void mont_inv(u256 *r, const u256 *a, const mctx *c) {
    u256 e = c->m; u256 two = {{2, 0, 0, 0}}; u256_sub(&e, &e, &two);
    u256 acc = c->one;
    for (int i = 255; i >= 0; --i) {
        mont_mul(&acc, &acc, &acc, c);
        if ((e.v[i >> 6] >> (i & 63)) & 1) mont_mul(&acc, &acc, a, c);
    }
    *r = acc;
}

/* ---- points: Jacobian (X,Y,Z) over Fp in Montgomery form; Z == 0 is infinity ---------- */
typedef struct { u256 X, Y, Z; } jpoint;

static void pt_set_inf(jpoint *p) { memset(p, 0, sizeof *p); }
static int pt_is_inf(const jpoint *p) { return u256_is_zero(&p->Z); }

/* dbl-2001-b (a = -3): exact for any finite point, infinity in -> infinity out;
 * a point of order 2 does not exist on P-256 (prime order) so Y != 0 for finite points. */
static void pt_dbl(jpoint *r, const jpoint *p) {
    if (pt_is_inf(p)) { pt_set_inf(r); return; }
    u256 delta, gamma, beta, alpha, t1, t2, X3, Y3, Z3;
    mont_mul(&delta, &p->Z, &p->Z, &FP);
    mont_mul(&gamma, &p->Y, &p->Y, &FP);
    mont_mul(&beta, &p->X, &gamma, &FP);
    mod_sub(&t1, &p->X, &delta, &FP);
    mod_add(&t2, &p->X, &delta, &FP);
    mont_mul(&alpha, &t1, &t2, &FP);
    mod_add(&t1, &alpha, &alpha, &FP); mod_add(&alpha, &t1, &alpha, &FP);   /* 3*(X-d)(X+d) */
    mont_mul(&X3, &alpha, &alpha, &FP);
    u256 b2, b4, b8;
    mod_add(&b2, &beta, &beta, &FP); mod_add(&b4, &b2, &b2, &FP); mod_add(&b8, &b4, &b4, &FP);
    mod_sub(&X3, &X3, &b8, &FP);
    mod_add(&t1, &p->Y, &p->Z, &FP);
    mont_mul(&Z3, &t1, &t1, &FP);
    mod_sub(&Z3, &Z3, &gamma, &FP); mod_sub(&Z3, &Z3, &delta, &FP);
    mod_sub(&t1, &b4, &X3, &FP);
    mont_mul(&Y3, &alpha, &t1, &FP);
    mont_mul(&t2, &gamma, &gamma, &FP);
    mod_add(&t2, &t2, &t2, &FP); mod_add(&t2, &t2, &t2, &FP); mod_add(&t2, &t2, &t2, &FP); /* 8*g^2 */
    mod_sub(&Y3, &Y3, &t2, &FP);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

/* Exact group law on Jacobian inputs: handles infinity, P == Q (doubles), P == -Q (infinity). */
static void pt_add(jpoint *r, const jpoint *p, const jpoint *q) {
    if (pt_is_inf(p)) { *r = *q; return; }
    if (pt_is_inf(q)) { *r = *p; return; }
    u256 z1z1, z2z2, u1, u2, s1, s2, h, rr, t;
    mont_mul(&z1z1, &p->Z, &p->Z, &FP);
    mont_mul(&z2z2, &q->Z, &q->Z, &FP);
    mont_mul(&u1, &p->X, &z2z2, &FP);
    mont_mul(&u2, &q->X, &z1z1, &FP);
    mont_mul(&t, &q->Z, &z2z2, &FP); mont_mul(&s1, &p->Y, &t, &FP);
    mont_mul(&t, &p->Z, &z1z1, &FP); mont_mul(&s2, &q->Y, &t, &FP);
    mod_sub(&h, &u2, &u1, &FP);
    mod_sub(&rr, &s2, &s1, &FP);
    if (u256_is_zero(&h)) {
        if (u256_is_zero(&rr)) { pt_dbl(r, p); return; }
        pt_set_inf(r); return;
    }
    u256 hh, hhh, v, X3, Y3, Z3;
    mont_mul(&hh, &h, &h, &FP);
    mont_mul(&hhh, &h, &hh, &FP);
    mont_mul(&v, &u1, &hh, &FP);
    mont_mul(&X3, &rr, &rr, &FP);
    mod_sub(&X3, &X3, &hhh, &FP); mod_sub(&X3, &X3, &v, &FP); mod_sub(&X3, &X3, &v, &FP);
    mod_sub(&t, &v, &X3, &FP);
    mont_mul(&Y3, &rr, &t, &FP);
    mont_mul(&t, &s1, &hhh, &FP);
    mod_sub(&Y3, &Y3, &t, &FP);
    mont_mul(&t, &p->Z, &q->Z, &FP);
    mont_mul(&Z3, &t, &h, &FP);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

/* k*P with unsigned 4-bit windows, k given as plain (non-Montgomery) integer */
static void pt_mul(jpoint *r, const u256 *k, const jpoint *p) {
    jpoint tbl[16];
    pt_set_inf(&tbl[0]); tbl[1] = *p;
    for (int i = 2; i < 16; ++i) pt_add(&tbl[i], &tbl[i - 1], p);
    jpoint acc; pt_set_inf(&acc);
    for (int i = 63; i >= 0; --i) {
        pt_dbl(&acc, &acc); pt_dbl(&acc, &acc); pt_dbl(&acc, &acc); pt_dbl(&acc, &acc);
        unsigned d = (unsigned)(k->v[i >> 4] >> ((i & 15) * 4)) & 15u;
        pt_add(&acc, &acc, &tbl[d]);
    }
    *r = acc;
}

/* fixed-base comb for G: 32 windows x 256 entries (lazy, thread-safe init) — used by the
 * synthetic-data signer and by verification's u1*G. */
static jpoint *g_comb;
static pthread_once_t g_comb_once = PTHREAD_ONCE_INIT;
static void g_comb_build(void) {
    g_comb = (jpoint *)malloc(sizeof(jpoint) * 32 * 256);
    jpoint base; to_mont(&base.X, &GX, &FP); to_mont(&base.Y, &GY, &FP); base.Z = FP.one;
    for (int w = 0; w < 32; ++w) {
        jpoint *t = g_comb + w * 256;
        pt_set_inf(&t[0]); t[1] = base;
        for (int i = 2; i < 256; ++i) pt_add(&t[i], &t[i - 1], &base);
        jpoint nb; pt_add(&nb, &t[255], &base); base = nb;   /* 2^(8(w+1)) G */
    }
}
static void pt_mul_base(jpoint *r, const u256 *k) {
    pthread_once(&g_comb_once, g_comb_build);
    jpoint acc; pt_set_inf(&acc);
    for (int w = 0; w < 32; ++w) {
        unsigned d = (unsigned)(k->v[w >> 3] >> ((w & 7) * 8)) & 255u;
        pt_add(&acc, &acc, &g_comb[w * 256 + d]);
    }
    *r = acc;
}

/* affine x (plain integer) of a finite point */
static void pt_affine(u256 *x, u256 *y, const jpoint *p) {
    u256 zi, zi2, zi3, t;
    mont_inv(&zi, &p->Z, &FP);
    mont_mul(&zi2, &zi, &zi, &FP);
    mont_mul(&t, &p->X, &zi2, &FP); from_mont(x, &t, &FP);
    if (y) { mont_mul(&zi3, &zi2, &zi, &FP); mont_mul(&t, &p->Y, &zi3, &FP); from_mont(y, &t, &FP); }
}

static int on_curve(const u256 *x, const u256 *y) {   /* plain integers < p */
    u256 xm, ym, lhs, rhs, t, bm;
    to_mont(&xm, x, &FP); to_mont(&ym, y, &FP); to_mont(&bm, &CURVE_B, &FP);
    mont_mul(&lhs, &ym, &ym, &FP);
    mont_mul(&t, &xm, &xm, &FP); mont_mul(&rhs, &t, &xm, &FP);
    mod_sub(&rhs, &rhs, &xm, &FP); mod_sub(&rhs, &rhs, &xm, &FP); mod_sub(&rhs, &rhs, &xm, &FP);
    mod_add(&rhs, &rhs, &bm, &FP);
    return u256_eq(&lhs, &rhs);
}

/* ---- verification (crypto/ecdsa.verifyNISTEC restated) -------------------------------- */
int sbvo_p256_verify_raw(const uint8_t r_be[32], const uint8_t s_be[32], const uint8_t e_be[32],
                         const uint8_t qx_be[32], const uint8_t qy_be[32]) {
    u256 r, s, e, qx, qy;
    u256_from_be(&r, r_be); u256_from_be(&s, s_be); u256_from_be(&e, e_be);
    u256_from_be(&qx, qx_be); u256_from_be(&qy, qy_be);
    /* pointFromAffine */
    if (u256_geq(&qx, &FP.m) || u256_geq(&qy, &FP.m)) return 0;
    if (!on_curve(&qx, &qy)) return 0;
    /* bigmod SetBytes + IsZero */
    if (u256_is_zero(&r) || u256_geq(&r, &FN.m)) return 0;
    if (u256_is_zero(&s) || u256_geq(&s, &FN.m)) return 0;
    /* hashToNat: SetOverflowingBytes = one conditional subtraction (e < 2^256 < 2N) */
    if (u256_geq(&e, &FN.m)) u256_sub(&e, &e, &FN.m);
    u256 sm, wm, em, rm, u1, u2;
    to_mont(&sm, &s, &FN); mont_inv(&wm, &sm, &FN);
    to_mont(&em, &e, &FN); to_mont(&rm, &r, &FN);
    mont_mul(&u1, &em, &wm, &FN); from_mont(&u1, &u1, &FN);
    mont_mul(&u2, &rm, &wm, &FN); from_mont(&u2, &u2, &FN);
    jpoint Q, p1, p2, R;
    to_mont(&Q.X, &qx, &FP); to_mont(&Q.Y, &qy, &FP); Q.Z = FP.one;
    pt_mul_base(&p1, &u1);
    pt_mul(&p2, &u2, &Q);
    pt_add(&R, &p1, &p2);
    if (pt_is_inf(&R)) return 0;
    u256 x; pt_affine(&x, NULL, &R);
    if (u256_geq(&x, &FN.m)) u256_sub(&x, &x, &FN.m);
    return u256_eq(&x, &r);
}

int sbvo_p256_verify_tuple(const uint8_t t[160]) {
    return sbvo_p256_verify_raw(t, t + 32, t + 64, t + 96, t + 128);
}

/* ---- DER (cryptobyte-strict) ----------------------------------------------------------
 * returns number of header+content bytes consumed, or 0 on reject; body/blen receive the contents */
static size_t read_asn1(const uint8_t *p, size_t n, uint8_t want, const uint8_t **body, size_t *blen) {
    if (n < 2) return 0;
    uint8_t tag = p[0], lb = p[1];
    if ((tag & 0x1f) == 0x1f) return 0;
    size_t len, hdr;
    if (!(lb & 0x80)) { len = lb; hdr = 2; }
    else {
        unsigned ll = lb & 0x7f;
        if (ll == 0 || ll > 4) return 0;
        if (n < 2 + (size_t)ll) return 0;
        len = 0;
        for (unsigned i = 0; i < ll; ++i) len = (len << 8) | p[2 + i];
        if (len < 128) return 0;
        if ((len >> ((ll - 1) * 8)) == 0) return 0;
        hdr = 2 + ll;
    }
    if (n < hdr + len) return 0;
    if (tag != want) return 0;
    *body = p + hdr; *blen = len;
    return hdr + len;
}
static int read_uint(const uint8_t **pp, size_t *pn, uint8_t out[32]) {
    const uint8_t *b; size_t bl;
    size_t used = read_asn1(*pp, *pn, 0x02, &b, &bl);
    if (!used) return -1;
    *pp += used; *pn -= used;
    if (bl == 0) return -1;
    if (bl > 1) {
        if (b[0] == 0x00 && !(b[1] & 0x80)) return -1;
        if (b[0] == 0xff && (b[1] & 0x80)) return -1;
    }
    if (b[0] & 0x80) return -1;                 /* negative */
    while (bl > 1 && b[0] == 0) { ++b; --bl; }
    if (bl > 32) return -2;                     /* bigmod: overflows the modulus size */
    memset(out, 0, 32);
    memcpy(out + 32 - bl, b, bl);
    return 0;
}
/* 0 = parsed into rs[64] (r|s big-endian, zero padded); <0 = reject */
int sbvo_p256_parse_der(const uint8_t *der, size_t len, uint8_t rs[64]) {
    const uint8_t *inner; size_t il;
    size_t used = read_asn1(der, len, 0x30, &inner, &il);
    if (!used || used != len) return -1;
    int rc;
    if ((rc = read_uint(&inner, &il, rs)) != 0) return rc;
    if ((rc = read_uint(&inner, &il, rs + 32)) != 0) return rc;
    if (il != 0) return -1;
    return 0;
}

/* crypto/ecdsa.VerifyASN1(pub, hash, sig) with pub given as 32-byte big-endian coordinates */
int sbvo_p256_verify_asn1(const uint8_t qx[32], const uint8_t qy[32], const uint8_t *hash, size_t hlen,
                          const uint8_t *sig, size_t slen) {
    uint8_t rs[64], e[32];
    if (sbvo_p256_parse_der(sig, slen, rs) != 0) return 0;
    memset(e, 0, 32);
    if (hlen >= 32) memcpy(e, hash, 32); else memcpy(e + 32 - hlen, hash, hlen);
    return sbvo_p256_verify_raw(rs, rs + 32, e, qx, qy);
}

/* ---- SHA-256 (FIPS 180-4), for the synthetic generator --------------------------------- */
static const uint32_t K256[64] = {
    0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
    0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
    0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
    0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
    0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
    0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
    0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
    0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t h[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
        w[i] = ((uint32_t)blk[4*i] << 24) | ((uint32_t)blk[4*i+1] << 16) | ((uint32_t)blk[4*i+2] << 8) | blk[4*i+3];
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = ROR(w[i-15], 7) ^ ROR(w[i-15], 18) ^ (w[i-15] >> 3);
        uint32_t s1 = ROR(w[i-2], 17) ^ ROR(w[i-2], 19) ^ (w[i-2] >> 10);
        w[i] = w[i-16] + s0 + w[i-7] + s1;
    }
    uint32_t a=h[0],b=h[1],c=h[2],d=h[3],e=h[4],f=h[5],g=h[6],hh=h[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t S1 = ROR(e,6) ^ ROR(e,11) ^ ROR(e,25), ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ROR(a,2) ^ ROR(a,13) ^ ROR(a,22), mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh=g; g=f; f=e; e=d+t1; d=c; c=b; b=a; a=t1+t2;
    }
    h[0]+=a;h[1]+=b;h[2]+=c;h[3]+=d;h[4]+=e;h[5]+=f;h[6]+=g;h[7]+=hh;
}
void sbvo_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
    size_t i = 0;
    for (; i + 64 <= len; i += 64) sha256_block(h, msg + i);
    uint8_t tail[128]; size_t rem = len - i;
    memset(tail, 0, sizeof tail); memcpy(tail, msg + i, rem);
    tail[rem] = 0x80;
    size_t tl = (rem < 56) ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int j = 0; j < 8; ++j) tail[tl - 1 - j] = (uint8_t)(bits >> (8 * j));
    sha256_block(h, tail); if (tl == 128) sha256_block(h, tail + 64);
    for (int j = 0; j < 8; ++j) { out[4*j]=(uint8_t)(h[j]>>24); out[4*j+1]=(uint8_t)(h[j]>>16); out[4*j+2]=(uint8_t)(h[j]>>8); out[4*j+3]=(uint8_t)h[j]; }
}

/* ---- signing / key derivation for synthetic data -------------------------------------- */
void sbvo_p256_pubkey(const uint8_t d_be[32], uint8_t q[64]) {
    u256 d, x, y; u256_from_be(&d, d_be);
    jpoint p; pt_mul_base(&p, &d);
    pt_affine(&x, &y, &p);
    u256_to_be(q, &x); u256_to_be(q + 32, &y);
}
/* textbook ECDSA sign with explicit nonce; returns 0 on success, -1 if r or s would be 0 */
int sbvo_p256_sign(const uint8_t d_be[32], const uint8_t k_be[32], const uint8_t h32[32], uint8_t rs[64]) {
    u256 d, k, e, x; u256_from_be(&d, d_be); u256_from_be(&k, k_be); u256_from_be(&e, h32);
    if (u256_geq(&e, &FN.m)) u256_sub(&e, &e, &FN.m);
    jpoint p; pt_mul_base(&p, &k);
    if (pt_is_inf(&p)) return -1;
    pt_affine(&x, NULL, &p);
    if (u256_geq(&x, &FN.m)) u256_sub(&x, &x, &FN.m);
    if (u256_is_zero(&x)) return -1;
    u256 km, ki, rm, dm, em, t, s;
    to_mont(&km, &k, &FN); mont_inv(&ki, &km, &FN);
    to_mont(&rm, &x, &FN); to_mont(&dm, &d, &FN); to_mont(&em, &e, &FN);
    mont_mul(&t, &rm, &dm, &FN); mod_add(&t, &t, &em, &FN);
    mont_mul(&s, &ki, &t, &FN); from_mont(&s, &s, &FN);
    if (u256_is_zero(&s)) return -1;
    u256_to_be(rs, &x); u256_to_be(rs + 32, &s);
    return 0;
}

/* scalar = SHA-256(label || seed_be32 || index_be64) mod (N-1) + 1    */
static void derive_scalar(const char *label, uint32_t seed, uint64_t idx, uint8_t out[32]) {
    uint8_t buf[64]; size_t ll = strlen(label);
    memcpy(buf, label, ll);
    buf[ll] = (uint8_t)(seed >> 24); buf[ll+1] = (uint8_t)(seed >> 16); buf[ll+2] = (uint8_t)(seed >> 8); buf[ll+3] = (uint8_t)seed;
    for (int j = 0; j < 8; ++j) buf[ll + 4 + j] = (uint8_t)(idx >> (56 - 8 * j));
    uint8_t h[32]; sbvo_sha256(buf, ll + 12, h);
    u256 v, nm1, one = {{1, 0, 0, 0}}; u256_from_be(&v, h);
    u256_sub(&nm1, &FN.m, &one);
    while (u256_geq(&v, &nm1)) u256_sub(&v, &v, &nm1);
    u256_add(&v, &v, &one);
    u256_to_be(out, &v);
}

typedef struct {
    uint32_t seed; size_t lo, hi; size_t nkeys; unsigned invalid_every;
    const uint8_t *sk, *pk; uint8_t *tuples; uint8_t *expect;
} gen_job;

static void *gen_worker(void *arg) {
    gen_job *j = (gen_job *)arg;
    for (size_t i = j->lo; i < j->hi; ++i) {
        uint8_t msg[32], h[32], k[32], rs[64];
        memset(msg, 0, 32); memcpy(msg, "sbv-msg", 7);
        for (int b = 0; b < 8; ++b) msg[24 + b] = (uint8_t)((uint64_t)i >> (56 - 8 * b));
        msg[8] = (uint8_t)(j->seed >> 24); msg[9] = (uint8_t)(j->seed >> 16); msg[10] = (uint8_t)(j->seed >> 8); msg[11] = (uint8_t)j->seed;
        sbvo_sha256(msg, 32, h);
        size_t key = i % j->nkeys;
        uint64_t ctr = i;
        do { derive_scalar("sbv-nonce", j->seed, ctr, k); ctr += 0x9E3779B97F4A7C15ull; }
        while (sbvo_p256_sign(j->sk + 32 * key, k, h, rs) != 0);
        uint8_t *t = j->tuples + 160 * i;
        memcpy(t, rs, 64); memcpy(t + 64, h, 32); memcpy(t + 96, j->pk + 64 * key, 64);
        int valid = 1;
        if (j->invalid_every && (i % j->invalid_every) == j->invalid_every - 1) {
            /* flip one pseudo-random bit of the 1280-bit tuple */
            uint8_t sel[32], lbl[40]; memcpy(lbl, "sbv-flip", 8);
            for (int b = 0; b < 8; ++b) lbl[8 + b] = (uint8_t)((uint64_t)i >> (56 - 8 * b));
            lbl[16] = (uint8_t)(j->seed >> 24); lbl[17] = (uint8_t)(j->seed >> 16); lbl[18] = (uint8_t)(j->seed >> 8); lbl[19] = (uint8_t)j->seed;
            sbvo_sha256(lbl, 20, sel);
            unsigned bit = (((unsigned)sel[0] << 8) | sel[1]) % 1280u;
            t[bit >> 3] ^= (uint8_t)(1u << (bit & 7));
            valid = sbvo_p256_verify_tuple(t);   /* a flipped bit is (essentially) always fatal; be exact */
        }
        if (j->expect && valid) __atomic_fetch_or(&j->expect[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
    }
    return NULL;
}

/* Synthetic batch per SURVEY.md §8d: nkeys key pairs, tuple i signs SHA-256(msg_i) with key
 * i % nkeys and a derived nonce; every `invalid_every`-th tuple has one bit flipped
 * (0 = all valid).  expect (ceil(n/8) bytes, LSB-first) receives the oracle's verdicts. */
void sbvo_gen_batch(uint32_t seed, size_t n, size_t nkeys, unsigned invalid_every,
                    uint8_t *tuples, uint8_t *expect, int threads) {
    uint8_t *sk = (uint8_t *)malloc(32 * nkeys), *pk = (uint8_t *)malloc(64 * nkeys);
    for (size_t i = 0; i < nkeys; ++i) { derive_scalar("sbv-key", seed, i, sk + 32 * i); sbvo_p256_pubkey(sk + 32 * i, pk + 64 * i); }
    if (expect) memset(expect, 0, (n + 7) / 8);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; gen_job jobs[256];
    size_t per = ((n + threads - 1) / threads + 7) & ~(size_t)7;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo >= n) break; if (hi > n) hi = n;
        jobs[t] = (gen_job){seed, lo, hi, nkeys, invalid_every, sk, pk, tuples, expect};
        pthread_create(&th[t], NULL, gen_worker, &jobs[t]); ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(sk); free(pk);
}

/* ---- batch verify (multi-threaded), the cpu_baseline leg ------------------------------- */
typedef struct { const uint8_t *tuples; size_t lo, hi; uint8_t *bitmap; } ver_job;
static void *ver_worker(void *arg) {
    ver_job *j = (ver_job *)arg;
    for (size_t i = j->lo; i < j->hi; ++i)
        if (sbvo_p256_verify_tuple(j->tuples + 160 * i))
            j->bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));     /* ranges are byte aligned */
    return NULL;
}
void sbvo_p256_verify_batch(const uint8_t *tuples, size_t n, uint8_t *bitmap, int threads) {
    memset(bitmap, 0, (n + 7) / 8);
    pthread_once(&g_comb_once, g_comb_build);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; ver_job jobs[256];
    size_t per = ((n + threads - 1) / threads + 7) & ~(size_t)7;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo >= n) break; if (hi > n) hi = n;
        jobs[t] = (ver_job){tuples, lo, hi, bitmap};
        pthread_create(&th[t], NULL, ver_worker, &jobs[t]); ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
}
