/*
 * datagen.c — synthetic (sk, pk, msg) signature batches for bench.py, built on OpenSSL's
 * libcrypto (EC_POINT_mul / BN), deliberately independent of oracle/ (which only the
 * checker legs may touch) and of the product.  Workload = SURVEY.md §8d / BASELINE.json
 * configs[1]: nkeys key pairs sk_i = SHA-256("sbv-key" || seed || i) mod (N-1) + 1, tuple j
 * signs SHA-256(msg_j) with key j % nkeys and nonce k_j = SHA-256("sbv-nonce" || seed || j)
 * mod (N-1) + 1; every `invalid_every`-th tuple has one pseudo-random bit flipped.
 * The derivations are byte-identical to oracle/p256_oracle.c:sbvo_gen_batch, so the two
 * generators cross-check each other (tests/test_datagen.py).
 */
#define OPENSSL_SUPPRESS_DEPRECATED 1
#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/obj_mac.h>
#include <openssl/sha.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t seed; size_t lo, hi, nkeys; unsigned invalid_every;
    const uint8_t *sk, *pk; uint8_t *tuples, *valid;
} job_t;

static void derive_scalar(const char *label, uint32_t seed, uint64_t idx, const BIGNUM *nm1, BIGNUM *out, BN_CTX *ctx) {
    uint8_t buf[64], h[32]; size_t ll = strlen(label);
    memcpy(buf, label, ll);
    buf[ll] = (uint8_t)(seed >> 24); buf[ll+1] = (uint8_t)(seed >> 16); buf[ll+2] = (uint8_t)(seed >> 8); buf[ll+3] = (uint8_t)seed;
    for (int j = 0; j < 8; ++j) buf[ll + 4 + j] = (uint8_t)(idx >> (56 - 8 * j));
    SHA256(buf, ll + 12, h);
    BN_bin2bn(h, 32, out);
    BN_mod(out, out, nm1, ctx);
    BN_add_word(out, 1);
}

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    EC_GROUP *grp = EC_GROUP_new_by_curve_name(NID_X9_62_prime256v1);
    BN_CTX *ctx = BN_CTX_new();
    BIGNUM *n = BN_new(), *nm1 = BN_new(), *k = BN_new(), *x = BN_new(), *r = BN_new(), *s = BN_new(),
           *e = BN_new(), *d = BN_new(), *t = BN_new();
    EC_POINT *R = EC_POINT_new(grp);
    EC_GROUP_get_order(grp, n, ctx);
    BN_copy(nm1, n); BN_sub_word(nm1, 1);
    for (size_t i = j->lo; i < j->hi; ++i) {
        uint8_t msg[32], h[32];
        memset(msg, 0, 32); memcpy(msg, "sbv-msg", 7);
        msg[8] = (uint8_t)(j->seed >> 24); msg[9] = (uint8_t)(j->seed >> 16); msg[10] = (uint8_t)(j->seed >> 8); msg[11] = (uint8_t)j->seed;
        for (int b = 0; b < 8; ++b) msg[24 + b] = (uint8_t)((uint64_t)i >> (56 - 8 * b));
        SHA256(msg, 32, h);
        const size_t key = i % j->nkeys;
        BN_bin2bn(j->sk + 32 * key, 32, d);
        BN_bin2bn(h, 32, e); BN_mod(e, e, n, ctx);
        uint64_t ctr = i;
        for (;;) {
            derive_scalar("sbv-nonce", j->seed, ctr, nm1, k, ctx);
            ctr += 0x9E3779B97F4A7C15ull;
            EC_POINT_mul(grp, R, k, NULL, NULL, ctx);
            EC_POINT_get_affine_coordinates(grp, R, x, NULL, ctx);
            BN_mod(r, x, n, ctx);
            if (BN_is_zero(r)) continue;
            BN_mod_mul(t, r, d, n, ctx);
            BN_mod_add(t, t, e, n, ctx);
            BN_mod_inverse(s, k, n, ctx);
            BN_mod_mul(s, s, t, n, ctx);
            if (BN_is_zero(s)) continue;
            break;
        }
        uint8_t *tp = j->tuples + 160 * i;
        BN_bn2binpad(r, tp, 32); BN_bn2binpad(s, tp + 32, 32);
        memcpy(tp + 64, h, 32); memcpy(tp + 96, j->pk + 64 * key, 64);
        int valid = 1;
        if (j->invalid_every && (i % j->invalid_every) == j->invalid_every - 1) {
            uint8_t lbl[20], sel[32]; memcpy(lbl, "sbv-flip", 8);
            for (int b = 0; b < 8; ++b) lbl[8 + b] = (uint8_t)((uint64_t)i >> (56 - 8 * b));
            lbl[16] = (uint8_t)(j->seed >> 24); lbl[17] = (uint8_t)(j->seed >> 16); lbl[18] = (uint8_t)(j->seed >> 8); lbl[19] = (uint8_t)j->seed;
            SHA256(lbl, 20, sel);
            unsigned bit = (((unsigned)sel[0] << 8) | sel[1]) % 1280u;
            tp[bit >> 3] ^= (uint8_t)(1u << (bit & 7));
            valid = 0;   /* by construction (a flipped bit surviving verification has probability ~2^-128) */
        }
        if (j->valid && valid) __atomic_fetch_or(&j->valid[i >> 3], (uint8_t)(1u << (i & 7)), __ATOMIC_RELAXED);
    }
    EC_POINT_free(R);
    BN_free(n); BN_free(nm1); BN_free(k); BN_free(x); BN_free(r); BN_free(s); BN_free(e); BN_free(d); BN_free(t);
    BN_CTX_free(ctx); EC_GROUP_free(grp);
    return NULL;
}

/* valid: ceil(n/8) bytes, bit i = 1 iff tuple i was left uncorrupted */
int sbvd_gen_batch(uint32_t seed, size_t n, size_t nkeys, unsigned invalid_every, uint8_t *tuples, uint8_t *valid, int threads) {
    EC_GROUP *grp = EC_GROUP_new_by_curve_name(NID_X9_62_prime256v1);
    if (!grp) return -1;
    BN_CTX *ctx = BN_CTX_new();
    BIGNUM *nn = BN_new(), *nm1 = BN_new(), *d = BN_new(), *x = BN_new(), *y = BN_new();
    EC_POINT *Q = EC_POINT_new(grp);
    EC_GROUP_get_order(grp, nn, ctx); BN_copy(nm1, nn); BN_sub_word(nm1, 1);
    uint8_t *sk = (uint8_t *)malloc(32 * nkeys), *pk = (uint8_t *)malloc(64 * nkeys);
    for (size_t i = 0; i < nkeys; ++i) {
        derive_scalar("sbv-key", seed, i, nm1, d, ctx);
        BN_bn2binpad(d, sk + 32 * i, 32);
        EC_POINT_mul(grp, Q, d, NULL, NULL, ctx);
        EC_POINT_get_affine_coordinates(grp, Q, x, y, ctx);
        BN_bn2binpad(x, pk + 64 * i, 32); BN_bn2binpad(y, pk + 64 * i + 32, 32);
    }
    EC_POINT_free(Q); BN_free(nn); BN_free(nm1); BN_free(d); BN_free(x); BN_free(y); BN_CTX_free(ctx); EC_GROUP_free(grp);
    if (valid) memset(valid, 0, (n + 7) / 8);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; job_t jobs[256];
    size_t per = ((n + threads - 1) / threads + 7) & ~(size_t)7;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo >= n) break; if (hi > n) hi = n;
        jobs[t] = (job_t){seed, lo, hi, nkeys, invalid_every, sk, pk, tuples, valid};
        pthread_create(&th[t], NULL, worker, &jobs[t]); ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(sk); free(pk);
    return 0;
}
