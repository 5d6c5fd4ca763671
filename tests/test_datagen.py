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
