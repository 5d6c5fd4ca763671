/*
 * openssl_check.c — independent third opinion for the oracle: OpenSSL libcrypto's
 * ECDSA_do_verify on the raw 160-byte ABI tuple (r|s|hash|Qx|Qy, big-endian).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/p256_oracle.c header).  OpenSSL is NOT the
 * reference (that is Go crypto/ecdsa, absent here); it agrees with Go on every
 * mathematically defined case (range, on-curve, group law) and is used to pin the
 * restatement on those classes, and as a second CPU timing ("openssl") beside the port.
 */
#define OPENSSL_SUPPRESS_DEPRECATED 1
#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/ecdsa.h>
#include <openssl/obj_mac.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

int sbvssl_p256_verify_tuple(const uint8_t t[160]) {
    int ok = 0;
    EC_KEY *key = EC_KEY_new_by_curve_name(NID_X9_62_prime256v1);
    BIGNUM *x = BN_bin2bn(t + 96, 32, NULL), *y = BN_bin2bn(t + 128, 32, NULL);
    BIGNUM *r = BN_bin2bn(t, 32, NULL), *s = BN_bin2bn(t + 32, 32, NULL);
    ECDSA_SIG *sig = ECDSA_SIG_new();
    if (key && x && y && r && s && sig && EC_KEY_set_public_key_affine_coordinates(key, x, y) == 1) {
        ECDSA_SIG_set0(sig, r, s); r = s = NULL;   /* ownership moved */
        ok = ECDSA_do_verify(t + 64, 32, sig, key) == 1;
    }
    BN_free(x); BN_free(y); BN_free(r); BN_free(s);
    ECDSA_SIG_free(sig); EC_KEY_free(key);
    return ok;
}

/* the same on secp256k1 (the "other curves" variant: oracle/secp256k1_oracle.c) */
int sbvssl_k256_verify_tuple(const uint8_t t[160]) {
    int ok = 0;
    EC_KEY *key = EC_KEY_new_by_curve_name(NID_secp256k1);
    BIGNUM *x = BN_bin2bn(t + 96, 32, NULL), *y = BN_bin2bn(t + 128, 32, NULL);
    BIGNUM *r = BN_bin2bn(t, 32, NULL), *s = BN_bin2bn(t + 32, 32, NULL);
    ECDSA_SIG *sig = ECDSA_SIG_new();
    if (key && x && y && r && s && sig && EC_KEY_set_public_key_affine_coordinates(key, x, y) == 1) {
        ECDSA_SIG_set0(sig, r, s); r = s = NULL;
        ok = ECDSA_do_verify(t + 64, 32, sig, key) == 1;
    }
    BN_free(x); BN_free(y); BN_free(r); BN_free(s);
    ECDSA_SIG_free(sig); EC_KEY_free(key);
    return ok;
}

typedef struct { const uint8_t *tuples; size_t lo, hi; uint8_t *bitmap; int k256; } job_t;
static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    for (size_t i = j->lo; i < j->hi; ++i)
        if (j->k256 ? sbvssl_k256_verify_tuple(j->tuples + 160 * i) : sbvssl_p256_verify_tuple(j->tuples + 160 * i))
            j->bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
    return NULL;
}
static void verify_batch_curve(const uint8_t *tuples, size_t n, uint8_t *bitmap, int threads, int k256);
void sbvssl_p256_verify_batch(const uint8_t *tuples, size_t n, uint8_t *bitmap, int threads) { verify_batch_curve(tuples, n, bitmap, threads, 0); }
void sbvssl_k256_verify_batch(const uint8_t *tuples, size_t n, uint8_t *bitmap, int threads) { verify_batch_curve(tuples, n, bitmap, threads, 1); }
static void verify_batch_curve(const uint8_t *tuples, size_t n, uint8_t *bitmap, int threads, int k256) {
    memset(bitmap, 0, (n + 7) / 8);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; job_t jobs[256];
    size_t per = ((n + threads - 1) / threads + 7) & ~(size_t)7;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo >= n) break; if (hi > n) hi = n;
        jobs[t] = (job_t){tuples, lo, hi, bitmap, k256};
        pthread_create(&th[t], NULL, worker, &jobs[t]); ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
}

/* Ed25519 third opinion (EVP_DigestVerify).  OpenSSL follows RFC 8032 with its own choices on
 * non-canonical / small-order inputs, so tests consult it only on honest and bit-flipped
 * signatures, where every correct implementation must agree. */
#include <openssl/evp.h>
int sbvssl_ed25519_verify(const uint8_t pk[32], const uint8_t *msg, size_t len, const uint8_t sig[64]) {
    int ok = 0;
    EVP_PKEY *key = EVP_PKEY_new_raw_public_key(EVP_PKEY_ED25519, NULL, pk, 32);
    EVP_MD_CTX *ctx = EVP_MD_CTX_new();
    if (key && ctx && EVP_DigestVerifyInit(ctx, NULL, NULL, NULL, key) == 1)
        ok = EVP_DigestVerify(ctx, sig, 64, msg, len) == 1;
    EVP_MD_CTX_free(ctx); EVP_PKEY_free(key);
    return ok;
}

/* A whole synthetic Ed25519 batch (oracle/ed25519_oracle.c: sbvo_ed25519_gen_batch) through EVP_DigestVerify: tuple i is
 * R|S|A|k (128 bytes); OpenSSL needs the message, which the generator derives from (seed, i) alone — "sbv-ed-msg", the seed
 * at bytes 12..15, i big-endian at bytes 24..31 (egen_worker) — and takes A and R|S from the tuple, i.e. the bytes a verifier
 * would see (the generator's bit flips are already in them; k is ignored: OpenSSL hashes R|A|M itself).  The flips are random
 * single bits, which no two correct implementations judge differently (see the note above), so the bitmap must equal the
 * oracle's and the device's on every tuple.  Used by tests/test_gpu_ed25519.py and as bench.py's Ed25519 CPU baseline. */
typedef struct { uint32_t seed; const uint8_t *tuples; size_t first, lo, hi; uint8_t *bitmap; } edjob_t;
static void *ed_worker(void *arg) {
    edjob_t *j = (edjob_t *)arg;
    for (size_t l = j->lo; l < j->hi; ++l) {                 /* l: index inside the slice; i: index in the generated batch */
        const uint64_t i = (uint64_t)j->first + l;
        uint8_t msg[32];
        memset(msg, 0, 32); memcpy(msg, "sbv-ed-msg", 10);
        msg[12] = (uint8_t)(j->seed >> 24); msg[13] = (uint8_t)(j->seed >> 16); msg[14] = (uint8_t)(j->seed >> 8); msg[15] = (uint8_t)j->seed;
        for (int b = 0; b < 8; ++b) msg[24 + b] = (uint8_t)(i >> (56 - 8 * b));
        const uint8_t *t = j->tuples + 128 * l;
        if (sbvssl_ed25519_verify(t + 64, msg, 32, t)) j->bitmap[l >> 3] |= (uint8_t)(1u << (l & 7));
    }
    return NULL;
}
/* tuples = n tuples starting at tuple `first` of the batch generated from `seed` (the message depends on the global index) */
void sbvssl_ed25519_verify_gen_batch(uint32_t seed, const uint8_t *tuples, size_t first, size_t n, uint8_t *bitmap, int threads) {
    memset(bitmap, 0, (n + 7) / 8);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; edjob_t jobs[256];
    size_t per = ((n + threads - 1) / threads + 7) & ~(size_t)7;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per; if (lo >= n) break; if (hi > n) hi = n;
        jobs[t] = (edjob_t){seed, tuples, first, lo, hi, bitmap};
        pthread_create(&th[t], NULL, ed_worker, &jobs[t]); ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
}

/* ---- a strict-DER judge for VerifyASN1, built only from OpenSSL primitives ---------------------------------------
 * The DER / encoding classes of the golden vectors used to be confirmed by the builder's C restatement and the builder's Python
 * twin only — one author, one reading of Go's crypto/ecdsa.parseSignature (x/crypto/cryptobyte).  This is a third opinion that
 * shares no parsing code with either:
 *     d2i_ECDSA_SIG            OpenSSL's own ASN.1 decoder (BER-tolerant)
 *     all input consumed       cryptobyte: !input.Empty() after the SEQUENCE is an error
 *     r, s not negative        cryptobyte ReadASN1Integer into []byte refuses a set sign bit
 *     i2d_ECDSA_SIG == input   byte for byte: whatever BER liberty the decoder took (non-minimal lengths, padded or non-minimal
 *                              integers, indefinite forms) cannot survive a canonical DER re-encoding
 *     ECDSA_do_verify          on the hash AS GIVEN (any length: OpenSSL truncates to the leftmost 32 bytes and takes a shorter
 *                              digest as the integer it spells, which is hashToNat's rule)
 * Returns 1 = accept, 0 = reject.  The key goes through EC_KEY_set_public_key_affine_coordinates (coordinates < p, on the curve). */
int sbvssl_p256_verify_asn1(const uint8_t qx[32], const uint8_t qy[32], const uint8_t *hash, size_t hlen, const uint8_t *der, size_t dlen) {
    int ok = 0;
    const unsigned char *p = der;
    ECDSA_SIG *sig = dlen ? d2i_ECDSA_SIG(NULL, &p, (long)dlen) : NULL;
    unsigned char *re = NULL;
    EC_KEY *key = NULL;
    BIGNUM *x = NULL, *y = NULL;
    if (!sig || (size_t)(p - der) != dlen) goto done;
    {
        const BIGNUM *r = NULL, *s = NULL;
        ECDSA_SIG_get0(sig, &r, &s);
        if (!r || !s || BN_is_negative(r) || BN_is_negative(s)) goto done;
    }
    {
        const int rl = i2d_ECDSA_SIG(sig, &re);
        if (rl < 0 || (size_t)rl != dlen || memcmp(re, der, dlen) != 0) goto done;
    }
    key = EC_KEY_new_by_curve_name(NID_X9_62_prime256v1);
    x = BN_bin2bn(qx, 32, NULL);
    y = BN_bin2bn(qy, 32, NULL);
    if (!key || !x || !y || EC_KEY_set_public_key_affine_coordinates(key, x, y) != 1) goto done;
    {
        static const unsigned char none = 0;
        ok = ECDSA_do_verify(hlen ? hash : &none, (int)hlen, sig, key) == 1;
    }
done:
    OPENSSL_free(re);
    BN_free(x); BN_free(y);
    EC_KEY_free(key);
    ECDSA_SIG_free(sig);
    return ok;
}
/* the parse half alone: 1 = the judge above would hand (r, s) to the verification */
int sbvssl_p256_der_is_strict(const uint8_t *der, size_t dlen) {
    const unsigned char *p = der;
    ECDSA_SIG *sig = dlen ? d2i_ECDSA_SIG(NULL, &p, (long)dlen) : NULL;
    unsigned char *re = NULL;
    int ok = 0;
    if (sig && (size_t)(p - der) == dlen) {
        const BIGNUM *r = NULL, *s = NULL;
        ECDSA_SIG_get0(sig, &r, &s);
        if (r && s && !BN_is_negative(r) && !BN_is_negative(s)) {
            const int rl = i2d_ECDSA_SIG(sig, &re);
            ok = rl >= 0 && (size_t)rl == dlen && memcmp(re, der, dlen) == 0;
        }
    }
    OPENSSL_free(re);
    ECDSA_SIG_free(sig);
    return ok;
}
