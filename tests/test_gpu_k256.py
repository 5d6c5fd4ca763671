"""GPU tier for the secp256k1 variant: the HIP path through the C-ABI (sbv_secp256k1_verify_batch[_dev])
against the golden vectors, the oracle and OpenSSL (NID_secp256k1), bit for bit."""
import ctypes
import json
import os
import random

import pytest

import consensus_amd as sbv

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
THREADS = os.cpu_count() or 1


@pytest.fixture(scope="module")
def gpu():
    sbv.init(0)
    yield sbv


@pytest.fixture(scope="module")
def koracle(oracle):
    oracle.sbvo_k256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    oracle.sbvo_k256_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_int]
    return oracle


def _gen(koracle, seed, n, nkeys, inv):
    tup = ctypes.create_string_buffer(160 * max(n, 1))
    exp = ctypes.create_string_buffer((n + 7) // 8 or 1)
    koracle.sbvo_k256_gen_batch(seed, n, nkeys, inv, tup, exp, THREADS)
    return tup, exp


def test_golden_vectors(gpu):
    vs = json.load(open(os.path.join(GOLDEN, "k256_vectors.json")))["vectors"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    got = sbv.bitmap_to_list(gpu.secp256k1_verify_batch(blob), len(vs))
    bad = [v["name"] for v, g in zip(vs, got) if g != v["accept"]]
    assert not bad, bad


def test_external_known_answers_rfc6979_secp256k1(gpu):
    """tests/golden/rfc6979_k256.json (community RFC 6979 secp256k1 vectors): the published (r, s) and (r, n - s) verify under
    d * G and not under a neighbouring key."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import k256_py as kc
    tuples, want = [], []
    for v in json.load(open(os.path.join(GOLDEN, "rfc6979_k256.json")))["vectors"]:
        h = hashlib.sha256(v["msg"].encode()).digest()
        r, s_low = int(v["sig"][:64], 16), int(v["sig"][64:], 16)
        Q = kc.pt_mul(int(v["d"], 16), kc.G)
        for s_ in (s_low, kc.N - s_low):
            tuples.append(kc.make_tuple(r, s_, h, Q)); want.append(True)
            tuples.append(kc.make_tuple(r, s_, h, kc.pt_add(Q, kc.pt_mul(2, kc.G)))); want.append(False)
    assert sbv.bitmap_to_list(gpu.secp256k1_verify_batch(b"".join(tuples)), len(tuples)) == want


@pytest.mark.parametrize("n", [0, 1, 7, 63, 64, 65, 257, 1000, 20000])
def test_ragged_sizes_match_oracle(gpu, koracle, n):
    tup, exp = _gen(koracle, 0x6B00 + n, n, 13, 3)
    assert gpu.secp256k1_verify_batch(tup.raw[:160 * n], n) == exp.raw[:(n + 7) // 8]


def test_garbage_and_p256_signatures_are_rejected(gpu, oracle, koracle):
    rng = random.Random(256)
    n = 2000
    junk = bytes(rng.getrandbits(8) for _ in range(160 * n))
    want = ctypes.create_string_buffer((n + 7) // 8)
    koracle.sbvo_k256_verify_batch(junk, n, want, THREADS)
    assert gpu.secp256k1_verify_batch(junk, n) == want.raw
    # honest P-256 signatures are not secp256k1 signatures (their keys are not even on this curve)
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x256, n, 9, 0, tup, exp, THREADS)
    assert gpu.secp256k1_verify_batch(tup.raw, n) == bytes((n + 7) // 8)


def test_batch_2_16_vs_oracle_and_openssl(gpu, koracle, openssl_check):
    n = 1 << 16
    tup, exp = _gen(koracle, 0x5B7F2026, n, 1024, 8)
    want = ctypes.create_string_buffer(n // 8)
    koracle.sbvo_k256_verify_batch(tup, n, want, THREADS)
    assert want.raw == exp.raw
    openssl_check.sbvssl_k256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    ssl = ctypes.create_string_buffer(n // 8)
    openssl_check.sbvssl_k256_verify_batch(tup, n, ssl, THREADS)
    got = gpu.secp256k1_verify_batch(tup.raw, n)
    assert got == want.raw == ssl.raw
    assert sum(sbv.bitmap_to_list(got, n)) == n - n // 8


def test_full_batch_2_20_device_resident_vs_openssl(gpu, koracle, openssl_check):
    """The headline batch shape (2^20 tuples, 1024 keys, 7/8 valid + 1/8 single-bit-corrupted) on this curve, device-resident,
    diffed whole against OpenSSL and the generator's oracle verdicts; idempotent."""
    import numpy as np
    import torch
    n = 1 << 20
    tup, exp = _gen(koracle, 0x5B7F2026, n, 1024, 8)
    openssl_check.sbvssl_k256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    ssl = ctypes.create_string_buffer(n // 8)
    openssl_check.sbvssl_k256_verify_batch(tup, n, ssl, THREADS)
    assert ssl.raw == exp.raw
    d_t = torch.frombuffer(tup, dtype=torch.uint8).cuda()
    d_b = torch.zeros(n // 8, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    for _ in range(2):
        d_b.zero_()
        gpu.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        assert d_b.cpu().numpy().tobytes() == exp.raw
    # timing for the record (one more pass)
    import time
    t0 = time.perf_counter()
    gpu.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"\nsecp256k1: 2^20 tuples in {1e3 * dt:.2f} ms = {n / dt / 1e6:.1f} M verifies/s")


def test_grouped_step_equals_the_one_lane_kernel_and_openssl(gpu, koracle, openssl_check):
    """The grouped step of this curve (k256_group.h: per-batch key combs, key-sorted Q phase) on 2^18 tuples with 300 keys and on a
    ragged 2^17 + 777: verdicts equal OpenSSL's, the oracle's and the one-lane kernel's (grouping off); group statistics say
    the comb path really ran."""
    import numpy as np
    openssl_check.sbvssl_k256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    for n, nkeys in ((1 << 18, 300), ((1 << 17) + 777, 40)):
        tup, exp = _gen(koracle, 0x6B00 + nkeys, n, nkeys, 6)
        ssl = ctypes.create_string_buffer((n + 7) // 8)
        openssl_check.sbvssl_k256_verify_batch(tup, n, ssl, THREADS)
        want = exp.raw[:(n + 7) // 8]
        assert ssl.raw == want
        got = gpu.secp256k1_verify_batch(tup.raw, n)
        assert got == want, [i for i in range(len(want)) if got[i] != want[i]][:8]
        groups, grouped, generic, rejected = gpu.last_group_stats()
        assert groups == nkeys and grouped > n * 0.8 and grouped + generic + rejected == n, (groups, grouped, generic, rejected)
        gpu.set_grouping(False)
        try:
            assert gpu.secp256k1_verify_batch(tup.raw, n) == want
        finally:
            gpu.set_grouping(True)


def test_key_table_cache_of_this_curve_on_gpu(gpu, oracle, koracle, openssl_check):
    """sbv_key_cache(SBV_SCHEME_SECP256K1) (round 4): the curve's own comb pool keeps the tables a grouped batch built.  Cold batch,
    warm batch (other signatures of the same 300 keys: all hits, no table kernel has work), a capacity below the key set
    (overflow rebuilt per batch), a warm batch of 4096 tuples (grouped although every key is far below the count threshold),
    cache off; and the two ECDSA curves cannot meet in a table: a P-256 batch verified on P-256 leaves this cache empty, the
    same bytes offered to THIS curve are all rejected (their keys are no points here) and are cached as invalid, after which
    both curves still give their own verdicts.  Every bitmap equals the oracle's and OpenSSL's."""
    openssl_check.sbvssl_k256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    K = sbv.SCHEME_SECP256K1
    n = 1 << 17

    def run(tup, exp, m=n):
        ssl = ctypes.create_string_buffer((m + 7) // 8)
        openssl_check.sbvssl_k256_verify_batch(tup, m, ssl, THREADS)
        want = exp.raw[:(m + 7) // 8]
        assert ssl.raw == want
        got = gpu.secp256k1_verify_batch(tup.raw[:160 * m], m)
        assert got == want, [i for i in range(len(want)) if got[i] != want[i]][:8]
        return gpu.key_cache_stats(K)

    a, ea = _gen(koracle, 0x4B01, n, 300, 7)
    b, eb = _gen(koracle, 0x4B01, n + 512, 300, 5)        # same seed: the same 300 keys, other signatures and corruptions
    c, ec = _gen(koracle, 0x4B02, n, 300, 7)
    try:
        gpu.key_cache(False, 0, K)
        gpu.key_cache(True, 1024, K)
        entries, hits, misses, cap = run(a, ea)
        assert hits == 0 and misses >= 300 and entries == misses and cap == 1024, (entries, hits, misses, cap)
        first = entries
        entries, hits, misses, cap = run(b, eb, n + 512)
        assert misses == 0 and hits == first and entries == first, (entries, hits, misses)
        small, es = _gen(koracle, 0x4B01, 4096, 300, 7)     # ~13 signatures per key: only the cache makes them groups
        entries, hits, misses, cap = run(small, es, 4096)
        groups, grouped, generic, rejected = gpu.last_group_stats()
        assert misses == 0 and hits >= 290 and grouped > 3500 and generic == 0, (hits, misses, groups, grouped, generic, rejected)
        entries, hits, misses, cap = run(c, ec)
        assert hits == 0 and misses >= 300 and entries == first + misses
        gpu.key_cache(False, 0, K)
        gpu.key_cache(True, 256, K)                         # smaller than one batch's key set
        entries, hits, misses, cap = run(a, ea)
        assert entries == 256 and cap == 256
        entries, hits, misses, cap = run(a, ea)
        assert hits == 256 and misses == first - 256
        # the two curves: P-256 tuples (valid there) through both entries
        gpu.key_cache(False, 0, K)
        gpu.key_cache(True, 1024, K)
        gpu.key_cache(False)
        gpu.key_cache(True, 4096)
        oracle.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        p, ep = ctypes.create_string_buffer(160 * n), ctypes.create_string_buffer(n // 8)
        oracle.sbvo_gen_batch(0x4B03, n, 200, 7, p, ep, THREADS)
        for rep in range(2):
            assert gpu.verify_batch(p.raw, n) == ep.raw
            ke = gpu.key_cache_stats(K)[0]                   # the P-256 entry never touches this curve's cache: empty before this
            assert ke == 0 if rep == 0 else ke >= 200        # curve has seen those bytes, their keys cached as invalid afterwards
            pe = gpu.key_cache_stats()[0]
            assert pe >= 200
            got = gpu.secp256k1_verify_batch(p.raw, n)
            assert got == bytes(n // 8)                      # not one key of that batch is a point of this curve
            assert gpu.key_cache_stats()[0] == pe
        assert gpu.key_cache_stats(K)[0] >= 200              # cached here as invalid keys
        entries, hits, misses, cap = run(a, ea)               # and this curve's honest batch is untouched by them
        gpu.key_cache(False, 0, K)
        entries, hits, misses, cap = run(a, ea)
        assert (entries, hits, misses) == (0, 0, 0)
    finally:
        gpu.key_cache(False, 0, K)
        gpu.key_cache(True, 1024, K)
        gpu.key_cache(False)
        gpu.key_cache(True, 4096)
