"""bench.py's synthetic batches come from tools/datagen.c (OpenSSL EC_POINT_mul / BN), kept
independent of oracle/.  The oracle has its own signer; the two must produce byte-identical
batches from the same seed — which pins the oracle's scalar arithmetic and fixed-base
multiplication against OpenSSL on thousands of signatures."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_openssl_datagen_equals_oracle_generator(oracle):
    import synth
    n = 3000
    tuples, valid = synth.gen_batch(0x5B7F2026, n, nkeys=64, invalid_every=8, threads=4, cache=False)
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x5B7F2026, n, 64, 8, tup, exp, 4)
    assert tuples.tobytes() == tup.raw
    assert valid.tobytes() == exp.raw[:(n + 7) // 8]     # "corrupted => invalid" agrees with the oracle's verdicts
    bm = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_p256_verify_batch(tup, n, bm, 4)
    assert bm.raw == exp.raw


def test_host_ed25519_generator_equals_oracle_generator(oracle):
    """tools/bench_ed25519.py's batch comes from the host library's RFC 8032 signer (consensus_amd/host/ed25519_host.cc,
    the api.Signer half of the product); the oracle has its own signer.  Same seed -> byte-identical tuples (which pins
    both signers, both SHA-512s and both mod-L reductions against each other on thousands of signatures), and the host
    generator's "corrupted => invalid" flags agree with the oracle's verdicts."""
    import hostlib
    h = hostlib.load()
    oracle.sbvo_ed25519_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int]
    n = 4000
    t1, e1 = ctypes.create_string_buffer(128 * n), ctypes.create_string_buffer((n + 7) // 8)
    t2, e2 = ctypes.create_string_buffer(128 * n), ctypes.create_string_buffer((n + 7) // 8)
    h.sbvh_ed25519_gen_batch(0x5B7F2026, n, 37, 8, ctypes.addressof(t1), ctypes.addressof(e1), 4)
    oracle.sbvo_ed25519_gen_batch(0x5B7F2026, n, 37, 8, t2, e2, 4)
    assert t1.raw == t2.raw
    assert e1.raw == e2.raw


def test_host_secp256k1_generator_is_what_the_oracle_says(oracle):
    """bench.py's secp256k1 leg takes its batch from the host library's RFC 6979 signer (consensus_amd/host/k256_host.cc, the
    api.Signer half of the product; no oracle code outside tests / smoke / cpu_baseline).  The oracle verifies every tuple of
    it and agrees with the generator's "untouched => valid, one bit flipped => invalid" flags; public keys match the oracle's
    key derivation from the same scalars' signatures (a valid signature under a wrong key would not verify)."""
    import hostlib
    h = hostlib.load()
    oracle.sbvo_k256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    n = 4096
    t1, e1 = ctypes.create_string_buffer(160 * n), ctypes.create_string_buffer(n // 8)
    h.sbvh_k256_gen_batch(0x5B7F2026, n, 37, 8, ctypes.addressof(t1), ctypes.addressof(e1), 4)
    bm = ctypes.create_string_buffer(n // 8)
    oracle.sbvo_k256_verify_batch(t1, n, bm, 4)
    assert bm.raw == e1.raw
    assert sum(bin(b).count("1") for b in e1.raw) == n - n // 8
    # deterministic, and the thread count does not change a byte
    t2, e2 = ctypes.create_string_buffer(160 * n), ctypes.create_string_buffer(n // 8)
    h.sbvh_k256_gen_batch(0x5B7F2026, n, 37, 8, ctypes.addressof(t2), ctypes.addressof(e2), 1)
    assert t1.raw == t2.raw and e1.raw == e2.raw
