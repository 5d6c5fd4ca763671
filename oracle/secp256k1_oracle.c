/*
 * secp256k1_oracle.c — CPU restatement of ECDSA verification over secp256k1 for the "other curves" variant of the
 * hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (consensus_amd/, bench.py's timed GPU leg) may link, load or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * Parity status: **parity unpinned by the reference** — SmartBFT-Go/consensus holds no signature arithmetic at all
 * (every api.Verifier in the tree is a no-op or a mock: examples/naive_chain/node.go:86-96, test/test_app.go:231-248),
 * and secp256k1 is not even in Go's standard library.  The seam this serves is curve-agnostic (api.Verifier /
 * api.Signer, pkg/api/dependencies.go:46-71).  This file restates the *published* algorithm — SEC 1 v2.0 §4.1.4 with
 * SEC 2 v2.0 §2.4.1 parameters, and the same input rules as the P-256 oracle (p256_oracle.c: 1 <= r, s <= n - 1, public
 * key coordinates < p and on the curve, hash = leftmost 32 bytes reduced mod n, exact group law, R = infinity rejected,
 * accept iff R.x mod n == r) — and is pinned against (i) the community RFC 6979 secp256k1 known answers
 * (tests/golden/rfc6979_k256.json: d, message, nonce, signature; the signer below reproduces r and s from the published
 * nonce, the verifier accepts them), (ii) OpenSSL 3.0 ECDSA_do_verify with NID_secp256k1 (oracle/openssl_check.c,
 * tests/test_k256_cpu.py) on every vector class and on whole seeded batches, (iii) a Python big-int twin (oracle/k256_py.py).
 *
 * Deliberately simple: 4 x 64-bit words, plain (non-Montgomery) residues, the pseudo-Mersenne folds 2^256 = 2^32 + 977
 * (mod p) and 2^256 = c (mod n), Fermat inversions, Jacobian double-and-add.  It shares no arithmetic with the device code
 * (29-bit signed limbs, division-step inversions, comb tables).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } k256;      /* little-endian words */

static const k256 KP = {{0xFFFFFFFEFFFFFC2Full, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull}};
static const k256 KN = {{0xBFD25E8CD0364141ull, 0xBAAEDCE6AF48A03Bull, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull}};
static const k256 KGX = {{0x59F2815B16F81798ull, 0x029BFCDB2DCE28D9ull, 0x55A06295CE870B07ull, 0x79BE667EF9DCBBACull}};
static const k256 KGY = {{0x9C47D08FFB10D4B8ull, 0xFD17B448A6855419ull, 0x5DA4FBFC0E1108A8ull, 0x483ADA7726A3C465ull}};
/* 2^256 mod p and 2^256 mod n, as (up to) three words */
static const uint64_t FOLD_P[3] = {0x00000001000003D1ull, 0, 0};
static const uint64_t FOLD_N[3] = {0x402DA1732FC9BEBFull, 0x4551231950B75FC4ull, 0x1ull};

static int k_is_zero(const k256 *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int k_eq(const k256 *a, const k256 *b) { return a->v[0] == b->v[0] && a->v[1] == b->v[1] && a->v[2] == b->v[2] && a->v[3] == b->v[3]; }
static int k_geq(const k256 *a, const k256 *b) {
    for (int i = 3; i >= 0; --i) { if (a->v[i] > b->v[i]) return 1; if (a->v[i] < b->v[i]) return 0; }
    return 1;
}
static uint64_t k_add(k256 *r, const k256 *a, const k256 *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a->v[i] + b->v[i]; r->v[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t k_sub(k256 *r, const k256 *a, const k256 *b) {
    uint64_t bw = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->v[i] - b->v[i] - bw;
        r->v[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1;
    }
    return bw;
}
static void k_from_be(k256 *r, const uint8_t b[32]) {
    for (int i = 0; i < 4; ++i) { uint64_t w = 0; for (int j = 0; j < 8; ++j) w = (w << 8) | b[(3 - i) * 8 + j]; r->v[i] = w; }
}
static void k_to_be(uint8_t b[32], const k256 *a) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[(3 - i) * 8 + j] = (uint8_t)(a->v[i] >> (56 - 8 * j));
}

/* x (up to 8 words) mod m, where 2^256 = fold (mod m): repeat x = lo + hi * fold until hi is 0, then subtract m */
static void k_reduce(k256 *r, const uint64_t x_in[8], const k256 *m, const uint64_t fold[3]) {
    uint64_t x[8]; memcpy(x, x_in, sizeof x);
    for (;;) {
        if ((x[4] | x[5] | x[6] | x[7]) == 0) break;
        uint64_t y[8] = {x[0], x[1], x[2], x[3], 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {           /* y += x[4 + i] * fold << (64 i) */
            u128 c = 0;
            for (int j = 0; j < 3; ++j) {
                c += (u128)x[4 + i] * fold[j] + y[i + j];
                y[i + j] = (uint64_t)c; c >>= 64;
            }
            for (int k = i + 3; k < 8 && c; ++k) { c += y[k]; y[k] = (uint64_t)c; c >>= 64; }
        }
        memcpy(x, y, sizeof x);
    }
    k256 v = {{x[0], x[1], x[2], x[3]}};
    while (k_geq(&v, m)) k_sub(&v, &v, m);
    *r = v;
}
static void k_mulmod(k256 *r, const k256 *a, const k256 *b, const k256 *m, const uint64_t fold[3]) {
    uint64_t x[8] = {0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a->v[i] * b->v[j] + x[i + j]; x[i + j] = (uint64_t)c; c >>= 64; }
        x[i + 4] = (uint64_t)c;
    }
    k_reduce(r, x, m, fold);
}
static void k_addmod(k256 *r, const k256 *a, const k256 *b, const k256 *m) {
    uint64_t c = k_add(r, a, b);
    if (c || k_geq(r, m)) k_sub(r, r, m);
}
static void k_submod(k256 *r, const k256 *a, const k256 *b, const k256 *m) {
    if (k_sub(r, a, b)) k_add(r, r, m);
}
/* a^(m-2) mod m */
static void k_invmod(k256 *r, const k256 *a, const k256 *m, const uint64_t fold[3]) {
    k256 e, two = {{2, 0, 0, 0}}, acc = {{1, 0, 0, 0}};
    k_sub(&e, m, &two);
    for (int bit = 255; bit >= 0; --bit) {
        k_mulmod(&acc, &acc, &acc, m, fold);
        if ((e.v[bit >> 6] >> (bit & 63)) & 1) k_mulmod(&acc, &acc, a, m, fold);
    }
    *r = acc;
}
#define FMUL(r, a, b) k_mulmod((r), (a), (b), &KP, FOLD_P)
#define FADD(r, a, b) k_addmod((r), (a), (b), &KP)
#define FSUB(r, a, b) k_submod((r), (a), (b), &KP)

typedef struct { k256 X, Y, Z; } kpoint;      /* Jacobian; Z = 0 is the point at infinity */
static void kp_set_inf(kpoint *p) { memset(p, 0, sizeof *p); }
static int kp_is_inf(const kpoint *p) { return k_is_zero(&p->Z); }

/* 2P on y^2 = x^3 + 7 (a = 0):  S = 4 X Y^2, M = 3 X^2, X' = M^2 - 2 S, Y' = M (S - X') - 8 Y^4, Z' = 2 Y Z */
static void kp_dbl(kpoint *r, const kpoint *p) {
    if (kp_is_inf(p) || k_is_zero(&p->Y)) { kp_set_inf(r); return; }
    k256 yy, s, m, xx, t, y4, x3, y3, z3;
    FMUL(&yy, &p->Y, &p->Y);
    FMUL(&s, &p->X, &yy); FADD(&s, &s, &s); FADD(&s, &s, &s);
    FMUL(&xx, &p->X, &p->X); FADD(&m, &xx, &xx); FADD(&m, &m, &xx);
    FMUL(&x3, &m, &m); FSUB(&x3, &x3, &s); FSUB(&x3, &x3, &s);
    FMUL(&y4, &yy, &yy); FADD(&y4, &y4, &y4); FADD(&y4, &y4, &y4); FADD(&y4, &y4, &y4);
    FSUB(&t, &s, &x3); FMUL(&y3, &m, &t); FSUB(&y3, &y3, &y4);
    FMUL(&z3, &p->Y, &p->Z); FADD(&z3, &z3, &z3);
    r->X = x3; r->Y = y3; r->Z = z3;
}
/* P + Q, general Jacobian addition with every special case spelled out */
static void kp_add(kpoint *r, const kpoint *p, const kpoint *q) {
    if (kp_is_inf(p)) { *r = *q; return; }
    if (kp_is_inf(q)) { *r = *p; return; }
    k256 z1z1, z2z2, u1, u2, s1, s2, h, rr, t;
    FMUL(&z1z1, &p->Z, &p->Z); FMUL(&z2z2, &q->Z, &q->Z);
    FMUL(&u1, &p->X, &z2z2); FMUL(&u2, &q->X, &z1z1);
    FMUL(&t, &q->Z, &z2z2); FMUL(&s1, &p->Y, &t);
    FMUL(&t, &p->Z, &z1z1); FMUL(&s2, &q->Y, &t);
    FSUB(&h, &u2, &u1); FSUB(&rr, &s2, &s1);
    if (k_is_zero(&h)) {
        if (k_is_zero(&rr)) { kp_dbl(r, p); return; }
        kp_set_inf(r); return;
    }
    k256 hh, hhh, v, x3, y3, z3;
    FMUL(&hh, &h, &h); FMUL(&hhh, &h, &hh); FMUL(&v, &u1, &hh);
    FMUL(&x3, &rr, &rr); FSUB(&x3, &x3, &hhh); FSUB(&x3, &x3, &v); FSUB(&x3, &x3, &v);
    FSUB(&t, &v, &x3); FMUL(&y3, &rr, &t); FMUL(&t, &s1, &hhh); FSUB(&y3, &y3, &t);
    FMUL(&z3, &p->Z, &q->Z); FMUL(&z3, &z3, &h);
    r->X = x3; r->Y = y3; r->Z = z3;
}
static void kp_mul(kpoint *r, const k256 *k, const kpoint *p) {
    kpoint acc; kp_set_inf(&acc);
    for (int bit = 255; bit >= 0; --bit) {
        kp_dbl(&acc, &acc);
        if ((k->v[bit >> 6] >> (bit & 63)) & 1) kp_add(&acc, &acc, p);
    }
    *r = acc;
}
/* u1 * G + u2 * Q with one shared doubling chain (Shamir's trick) */
static void kp_mul2(kpoint *r, const k256 *u1, const kpoint *g, const k256 *u2, const kpoint *q) {
    kpoint gq, acc; kp_add(&gq, g, q); kp_set_inf(&acc);
    for (int bit = 255; bit >= 0; --bit) {
        kp_dbl(&acc, &acc);
        const int a = (int)((u1->v[bit >> 6] >> (bit & 63)) & 1), b = (int)((u2->v[bit >> 6] >> (bit & 63)) & 1);
        if (a && b) kp_add(&acc, &acc, &gq); else if (a) kp_add(&acc, &acc, g); else if (b) kp_add(&acc, &acc, q);
    }
    *r = acc;
}
static void kp_affine(k256 *x, k256 *y, const kpoint *p) {
    k256 zi, zi2, zi3;
    k_invmod(&zi, &p->Z, &KP, FOLD_P);
    FMUL(&zi2, &zi, &zi); FMUL(&zi3, &zi2, &zi);
    FMUL(x, &p->X, &zi2);
    if (y) FMUL(y, &p->Y, &zi3);
}
static int k_on_curve(const k256 *x, const k256 *y) {      /* plain integers < p */
    k256 l, r, seven = {{7, 0, 0, 0}};
    FMUL(&l, y, y);
    FMUL(&r, x, x); FMUL(&r, &r, x); FADD(&r, &r, &seven);
    return k_eq(&l, &r);
}
static kpoint k_generator(void) { kpoint g; g.X = KGX; g.Y = KGY; memset(&g.Z, 0, sizeof g.Z); g.Z.v[0] = 1; return g; }

/* SEC 1 v2.0 §4.1.4 on raw big-endian fields; e_be = the leftmost 32 bytes of the hash */
int sbvo_k256_verify_raw(const uint8_t r_be[32], const uint8_t s_be[32], const uint8_t e_be[32],
                         const uint8_t qx_be[32], const uint8_t qy_be[32]) {
    k256 r, s, e, qx, qy;
    k_from_be(&r, r_be); k_from_be(&s, s_be); k_from_be(&e, e_be); k_from_be(&qx, qx_be); k_from_be(&qy, qy_be);
    if (k_is_zero(&r) || k_geq(&r, &KN) || k_is_zero(&s) || k_geq(&s, &KN)) return 0;
    if (k_geq(&qx, &KP) || k_geq(&qy, &KP) || !k_on_curve(&qx, &qy)) return 0;
    if (k_geq(&e, &KN)) k_sub(&e, &e, &KN);
    k256 w, u1, u2;
    k_invmod(&w, &s, &KN, FOLD_N);
    k_mulmod(&u1, &e, &w, &KN, FOLD_N);
    k_mulmod(&u2, &r, &w, &KN, FOLD_N);
    kpoint g = k_generator(), q, R;
    q.X = qx; q.Y = qy; memset(&q.Z, 0, sizeof q.Z); q.Z.v[0] = 1;
    kp_mul2(&R, &u1, &g, &u2, &q);
    if (kp_is_inf(&R)) return 0;
    k256 x;
    kp_affine(&x, NULL, &R);
    if (k_geq(&x, &KN)) k_sub(&x, &x, &KN);
    return k_eq(&x, &r);
}
int sbvo_k256_verify_tuple(const uint8_t t[160]) { return sbvo_k256_verify_raw(t, t + 32, t + 64, t + 96, t + 128); }

void sbvo_k256_pubkey(const uint8_t d_be[32], uint8_t q[64]) {
    k256 d, x, y; k_from_be(&d, d_be);
    kpoint g = k_generator(), p; kp_mul(&p, &d, &g);
    kp_affine(&x, &y, &p);
    k_to_be(q, &x); k_to_be(q + 32, &y);
}
/* textbook ECDSA sign with an explicit nonce; 0 on success, -1 if r or s would be 0 */
int sbvo_k256_sign(const uint8_t d_be[32], const uint8_t k_be[32], const uint8_t h32[32], uint8_t rs[64]) {
    k256 d, k, e, x; k_from_be(&d, d_be); k_from_be(&k, k_be); k_from_be(&e, h32);
    if (k_is_zero(&k) || k_geq(&k, &KN)) return -1;
    if (k_geq(&e, &KN)) k_sub(&e, &e, &KN);
    kpoint g = k_generator(), p; kp_mul(&p, &k, &g);
    if (kp_is_inf(&p)) return -1;
    kp_affine(&x, NULL, &p);
    if (k_geq(&x, &KN)) k_sub(&x, &x, &KN);
    if (k_is_zero(&x)) return -1;
    k256 ki, t, s;
    k_invmod(&ki, &k, &KN, FOLD_N);
    k_mulmod(&t, &x, &d, &KN, FOLD_N); k_addmod(&t, &t, &e, &KN);
    k_mulmod(&s, &ki, &t, &KN, FOLD_N);
    if (k_is_zero(&s)) return -1;
    k_to_be(rs, &x); k_to_be(rs + 32, &s);
    return 0;
}

/* ---- synthetic batches: same scheme as sbvo_gen_batch (p256_oracle.c), labels prefixed so that the two curves differ --------- */
void sbvo_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);      /* p256_oracle.c */
static void k_derive_scalar(const char *label, uint32_t seed, uint64_t idx, uint8_t out[32]) {
    uint8_t buf[64]; size_t ll = strlen(label);
    memcpy(buf, label, ll);
    buf[ll] = (uint8_t)(seed >> 24); buf[ll+1] = (uint8_t)(seed >> 16); buf[ll+2] = (uint8_t)(seed >> 8); buf[ll+3] = (uint8_t)seed;
    for (int j = 0; j < 8; ++j) buf[ll + 4 + j] = (uint8_t)(idx >> (56 - 8 * j));
    uint8_t h[32]; sbvo_sha256(buf, ll + 12, h);
    k256 v, nm1, one = {{1, 0, 0, 0}}; k_from_be(&v, h);
    k_sub(&nm1, &KN, &one);
    while (k_geq(&v, &nm1)) k_sub(&v, &v, &nm1);
    k_add(&v, &v, &one);
    k_to_be(out, &v);
}
typedef struct {
    uint32_t seed; size_t lo, hi; size_t nkeys; unsigned invalid_every;
    const uint8_t *sk, *pk; uint8_t *tuples; uint8_t *expect;
} kgen_job;
static void *kgen_worker(void *arg) {
    kgen_job *j = (kgen_job *)arg;
    for (size_t i = j->lo; i < j->hi; ++i) {
        uint8_t msg[32], h[32], k[32], rs[64];
        memset(msg, 0, 32); memcpy(msg, "k256-msg", 8);
        for (int b = 0; b < 8; ++b) msg[24 + b] = (uint8_t)((uint64_t)i >> (56 - 8 * b));
        msg[8] = (uint8_t)(j->seed >> 24); msg[9] = (uint8_t)(j->seed >> 16); msg[10] = (uint8_t)(j->seed >> 8); msg[11] = (uint8_t)j->seed;
        sbvo_sha256(msg, 32, h);
        size_t key = i % j->nkeys;
        uint64_t ctr = i;
        do { k_derive_scalar("k256-nonce", j->seed, ctr, k); ctr += 0x9E3779B97F4A7C15ull; }
        while (sbvo_k256_sign(j->sk + 32 * key, k, h, rs) != 0);
        uint8_t *t = j->tuples + 160 * i;
        memcpy(t, rs, 64); memcpy(t + 64, h, 32); memcpy(t + 96, j->pk + 64 * key, 64);
        int valid = 1;
        if (j->invalid_every && (i % j->invalid_every) == j->invalid_every - 1) {
            uint8_t sel[32], lbl[40]; memcpy(lbl, "k256flip", 8);
            for (int b = 0; b < 8; ++b) lbl[8 + b] = (uint8_t)((uint64_t)i >> (56 - 8 * b));
            lbl[16] = (uint8_t)(j->seed >> 24); lbl[17] = (uint8_t)(j->seed >> 16); lbl[18] = (uint8_t)(j->seed >> 8); lbl[19] = (uint8_t)j->seed;
            sbvo_sha256(lbl, 20, sel);
            unsigned bit = (((unsigned)sel[0] << 8) | sel[1]) % 1280u;
            t[bit >> 3] ^= (uint8_t)(1u << (bit & 7));
            valid = sbvo_k256_verify_tuple(t);
        }
        if (j->expect && valid) __atomic_fetch_or(&j->expect[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
    }
    return NULL;
}
void sbvo_k256_gen_batch(uint32_t seed, size_t n, size_t nkeys, unsigned invalid_every,
                         uint8_t *tuples, uint8_t *expect, int threads) {
    uint8_t *sk = (uint8_t *)malloc(32 * nkeys), *pk = (uint8_t *)malloc(64 * nkeys);
    for (size_t i = 0; i < nkeys; ++i) { k_derive_scalar("k256-key", seed, i, sk + 32 * i); sbvo_k256_pubkey(sk + 32 * i, pk + 64 * i); }
    if (expect) memset(expect, 0, (n + 7) / 8);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; kgen_job jobs[256];
    size_t per = ((n + threads - 1) / threads + 7) & ~(size_t)7;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo >= n) break; if (hi > n) hi = n;
        jobs[t] = (kgen_job){seed, lo, hi, nkeys, invalid_every, sk, pk, tuples, expect};
        pthread_create(&th[t], NULL, kgen_worker, &jobs[t]); ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(sk); free(pk);
}

typedef struct { const uint8_t *tuples; size_t lo, hi; uint8_t *bitmap; } kver_job;
static void *kver_worker(void *arg) {
    kver_job *j = (kver_job *)arg;
    for (size_t i = j->lo; i < j->hi; ++i)
        if (sbvo_k256_verify_tuple(j->tuples + 160 * i)) j->bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
    return NULL;
}
void sbvo_k256_verify_batch(const uint8_t *tuples, size_t n, uint8_t *bitmap, int threads) {
    memset(bitmap, 0, (n + 7) / 8);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; kver_job jobs[256];
    size_t per = ((n + threads - 1) / threads + 7) & ~(size_t)7;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo >= n) break; if (hi > n) hi = n;
        jobs[t] = (kver_job){tuples, lo, hi, bitmap};
        pthread_create(&th[t], NULL, kver_worker, &jobs[t]); ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
}
