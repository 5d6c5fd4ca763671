"""CPU tier for the secp256k1 variant: the oracle against its pins (golden vectors from the Python
twin, OpenSSL NID_secp256k1), and the device algorithm (consensus_amd/csrc/k256_*.h compiled by g++ into tests/emul, with
the field's contract checks on) against big ints, the golden vectors and the oracle."""
import ctypes
import json
import os
import random
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import k256_py as ec  # noqa: E402
from test_emul_device_algo import emul  # noqa: E402,F401  (fixture: builds tests/emul/libsbv_emul.so)

P, N = ec.P, ec.N


@pytest.fixture(scope="module")
def k256_vectors():
    with open(os.path.join(HERE, "golden", "k256_vectors.json")) as f:
        return json.load(f)["vectors"]


@pytest.fixture(scope="module")
def koracle(oracle):
    oracle.sbvo_k256_verify_tuple.argtypes = [ctypes.c_char_p]
    oracle.sbvo_k256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    oracle.sbvo_k256_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_int]
    oracle.sbvo_k256_pubkey.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    oracle.sbvo_k256_sign.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    return oracle


def words(x):
    return (ctypes.c_uint32 * 8)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)])


def val(arr, n=8):
    return sum(int(arr[i]) << (32 * i) for i in range(n))


def bits(bm, n):
    return [bool((bm[i >> 3] >> (i & 7)) & 1) for i in range(n)]


def edge_values(m):
    return [0, 1, 2, m - 1, m - 2, (m - 1) // 2, 2**32 - 1, 2**32, 2**64 - 1, 2**128 - 1, 2**255 % m, (2**256 - 1) % m, 2**256 - 1,
            m, m + 1, 977, 2**32 + 977, 2**29 - 1, 2**232, 2**232 - 1, 2**261 % m,
            0xFFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000, 0x1FFFFFFF << 203]


# ---- the oracle against its pins ------------------------------------------------------------------------------------
def test_oracle_matches_the_golden_vectors(koracle, k256_vectors):
    for v in k256_vectors:
        assert bool(koracle.sbvo_k256_verify_tuple(bytes.fromhex(v["tuple"]))) == v["accept"], v["name"]


def test_golden_vectors_are_what_the_python_twin_says(k256_vectors):
    for v in k256_vectors[::3]:
        assert ec.verify_tuple(bytes.fromhex(v["tuple"])) == v["accept"], v["name"]
    assert sum(v["accept"] for v in k256_vectors) >= 50 and sum(not v["accept"] for v in k256_vectors) >= 50


def test_openssl_agrees_on_the_golden_vectors_and_a_seeded_batch(koracle, openssl_check, k256_vectors):
    openssl_check.sbvssl_k256_verify_tuple.argtypes = [ctypes.c_char_p]
    openssl_check.sbvssl_k256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    for v in k256_vectors:
        assert bool(openssl_check.sbvssl_k256_verify_tuple(bytes.fromhex(v["tuple"]))) == v["accept"], v["name"]
    n = 4096
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer(n // 8)
    koracle.sbvo_k256_gen_batch(0x6B32, n, 37, 4, tup, exp, 8)
    got = ctypes.create_string_buffer(n // 8)
    openssl_check.sbvssl_k256_verify_batch(tup, n, got, 8)
    assert got.raw == exp.raw
    assert sum(bits(exp.raw, n)) == n - n // 4


def test_oracle_signer_and_keys_match_the_python_twin(koracle):
    rng = random.Random(5)
    for _ in range(6):
        d, k = rng.randrange(1, N), rng.randrange(1, N)
        h = rng.randbytes(32)
        q = ctypes.create_string_buffer(64)
        koracle.sbvo_k256_pubkey(d.to_bytes(32, "big"), q)
        Q = ec.pt_mul(d, ec.G)
        assert q.raw == Q[0].to_bytes(32, "big") + Q[1].to_bytes(32, "big")
        rs = ctypes.create_string_buffer(64)
        assert koracle.sbvo_k256_sign(d.to_bytes(32, "big"), k.to_bytes(32, "big"), h, rs) == 0
        r, s = ec.sign(d, k, h)
        assert rs.raw == r.to_bytes(32, "big") + s.to_bytes(32, "big")


def test_external_known_answers_rfc6979_secp256k1(koracle, openssl_check, emul, k256_vectors):
    """The one pin that comes from outside this repository: the community RFC 6979 secp256k1 vectors (tests/golden/
    rfc6979_k256.json: provenance there).  The oracle's signer reproduces r and s (up to the vectors' low-S normalisation)
    from d and the published nonce; the oracle, its Python twin, OpenSSL and the emulated device path accept both the published
    (r, s) and (r, n - s), and reject them under a neighbouring key."""
    import hashlib
    kats = json.load(open(os.path.join(HERE, "golden", "rfc6979_k256.json")))["vectors"]
    openssl_check.sbvssl_k256_verify_tuple.argtypes = [ctypes.c_char_p]
    tuples, want = [], []
    for v in kats:
        d, k = bytes.fromhex(v["d"]), bytes.fromhex(v["k"])
        h = hashlib.sha256(v["msg"].encode()).digest()
        r, s_low = int(v["sig"][:64], 16), int(v["sig"][64:], 16)
        rs = ctypes.create_string_buffer(64)
        assert koracle.sbvo_k256_sign(d, k, h, rs) == 0
        s_raw = int.from_bytes(rs.raw[32:], "big")
        assert int.from_bytes(rs.raw[:32], "big") == r and min(s_raw, N - s_raw) == s_low and (s_raw != s_low) == v["s_was_high"]
        assert ec.sign(int(v["d"], 16), int(v["k"], 16), h) == (r, s_raw)
        q = ctypes.create_string_buffer(64)
        koracle.sbvo_k256_pubkey(d, q)
        Q = ec.pt_mul(int(v["d"], 16), ec.G)
        assert q.raw == Q[0].to_bytes(32, "big") + Q[1].to_bytes(32, "big")
        Q2 = ec.pt_add(Q, ec.pt_mul(2, ec.G))           # d = n - 1 makes Q = -G: Q + G would be infinity
        for s_ in (s_low, N - s_low):
            tuples.append(ec.make_tuple(r, s_, h, Q)); want.append(True)
            tuples.append(ec.make_tuple(r, s_, h, Q2)); want.append(False)
    for t, w in zip(tuples, want):
        assert bool(koracle.sbvo_k256_verify_tuple(t)) == w and ec.verify_tuple(t) == w
        assert bool(openssl_check.sbvssl_k256_verify_tuple(t)) == w
    bm = ctypes.create_string_buffer((len(tuples) + 7) // 8)
    emul.sbve_k256_verify_batch(b"".join(tuples), ctypes.c_size_t(len(tuples)), bm)
    assert bits(bm.raw, len(tuples)) == want


# ---- the device arithmetic against big ints ------------------------------------------------------------------------------
def test_field_operations_match_bigint(emul):
    rng = random.Random(11)
    vals = edge_values(P) + [rng.randrange(2**256) for _ in range(60)]
    out = (ctypes.c_uint32 * 8)()
    for a in vals:
        emul.sbve_kfe_op(1, words(a), words(0), out)
        assert val(out) == a * a % P
        emul.sbve_kfe_op(4, words(a), words(0), out)
        assert val(out) == (pow(a, -1, P) if a % P else 0)
        emul.sbve_kfe_op(6, words(a), words(0), out)
        assert val(out) == -a % P
        for b in vals[::5] + [a]:
            for op, want in [(0, a * b % P), (2, (a + b) % P), (3, (a - b) % P), (5, (3 * a - 8 * b) % P)]:
                emul.sbve_kfe_op(op, words(a), words(b), out)
                assert val(out) == want, (op, hex(a), hex(b))
            assert emul.sbve_kfe_chain_is_zero(words(a), words(b)) == 1


def test_field_products_at_the_extremes_of_the_operand_contract(emul):
    """kfe_mul / kfe_sqr on raw limb patterns up to the product bound 9 * max|a| * max|b| < 2^62.9: all limbs at +-(2^29 - 1)
    (uncarried differences), one operand at 3 x reduced, alternating signs, large top limbs — against big ints.  The emulator
    build asserts the bound itself; the 64-bit intermediates of the reduction must not wrap below it."""
    rng = random.Random(14)
    M = 2**29 - 1
    out = (ctypes.c_uint32 * 8)()

    def value(l):
        return sum(v << (29 * i) for i, v in enumerate(l))

    pats = [[M] * 8 + [2**24 + 2**20], [-M] * 8 + [-(2**24 + 2**20)], [M, -M] * 4 + [2**24], [-M, M] * 4 + [-(2**20)],
            [0] * 8 + [2**24 + 2**20], [M] * 9, [-M] * 9, [1] + [0] * 8, [0] * 9]
    pats += [[rng.randrange(-M, M + 1) for _ in range(8)] + [rng.randrange(-(2**24), 2**24)] for _ in range(40)]
    wide = [[3 * M] * 8 + [3 * 2**24], [-3 * M, 3 * M] * 4 + [2**25]]          # 3 x reduced against a reduced / difference operand
    for a in pats:
        arr_a = (ctypes.c_int32 * 9)(*a)
        emul.sbve_kfe_mul_raw(arr_a, None, out)
        assert val(out) == value(a) ** 2 % P, a
        for b in pats[::3] + wide:
            arr_b = (ctypes.c_int32 * 9)(*b)
            emul.sbve_kfe_mul_raw(arr_a, arr_b, out)
            assert val(out) == value(a) * value(b) % P, (a, b)


def test_scalar_operations_match_bigint(emul):
    rng = random.Random(12)
    vals = [v % N for v in edge_values(N)] + [rng.randrange(N) for _ in range(60)]
    out = (ctypes.c_uint32 * 8)()
    for a in vals:
        emul.sbve_ksc_inv(words(a), out)
        assert val(out) == (pow(a, -1, N) if a else 0)
        for b in vals[::4]:
            emul.sbve_ksc_mul(words(a), words(b), out)
            assert val(out) == a * b % N
    for x in [0, 1, N, N - 1, 2**256, 2**512 - 1, (N - 1) * (N - 1), 2**511, 2**385 - 1] + [rng.randrange(2**512) for _ in range(200)]:
        x16 = (ctypes.c_uint32 * 16)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(16)])
        emul.sbve_ksc_reduce512(x16, out)
        assert val(out) == x % N, hex(x)


def test_glv_decomposition_of_scalars(emul):
    """Round 6, k256_sc.h: ksc_split_lambda — k = (+-k1) + (+-k2) lambda (mod n) with k1, k2 < 2^128, for 200 000 random scalars and the
    edge scalars (0, 1, n - 1, n (reduced first), 2^256 - 1, lambda, n - lambda, the lattice vectors' neighbourhoods); the constants
    lambda, beta, g1, g2, -b1, -b2 are re-derived here from (n, lambda) by the extended Euclidean algorithm rather than trusted."""
    import math
    import random
    n, p = ec.N, ec.P
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    beta = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE
    assert pow(lam, 3, n) == 1 and lam != 1 and pow(beta, 3, p) == 1 and beta != 1
    lg = ec.pt_mul(lam, ec.G)
    assert lg == (beta * ec.G[0] % p, ec.G[1])                     # phi(G) = lambda G
    # the lattice basis from the extended Euclidean algorithm on (n, lambda)
    r0, r1, t0, t1, rows = n, lam, 0, 1, []
    while r1:
        q = r0 // r1
        r0, r1, t0, t1 = r1, r0 - q * r1, t1, t0 - q * t1
        rows.append((r0, t0))
    i = next(i for i, (r, _) in enumerate(rows) if r < math.isqrt(n))
    a1, b1 = rows[i][0], -rows[i][1]
    cands = [(rows[i - 1][0], -rows[i - 1][1]), (rows[i + 1][0], -rows[i + 1][1])]
    a2, b2 = min(cands, key=lambda v: v[0] ** 2 + v[1] ** 2)
    assert (a1 + b1 * lam) % n == 0 and (a2 + b2 * lam) % n == 0
    assert -b1 == 0xE4437ED6010E88286F547FA90ABFE4C3 and (-b2) % n == 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFE8A280AC50774346DD765CDA83DB1562C
    assert (2 ** 384 * b2 + n // 2) // n == 0x3086D221A7D46BCDE86C90E49284EB153DAA8A1471E8CA7FE893209A45DBB031
    assert (2 ** 384 * (-b1) + n // 2) // n == 0xE4437ED6010E88286F547FA90ABFE4C4221208AC9DF506C61571B4AE8AC47F71
    emul.sbve_k256_split_lambda.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    rng = random.Random(0x6157)
    ks = [0, 1, 2, n - 1, n, n + 1, 2 ** 256 - 1, lam, n - lam, lam + 1, (n + 1) // 2, n // 2, a1, a2, abs(b1), abs(b2), 2 ** 128, 2 ** 128 - 1, 2 ** 255]
    ks += [rng.randrange(n) for _ in range(200000)]
    kin = (ctypes.c_uint32 * 8)()
    out = (ctypes.c_uint32 * 18)()
    worst = 0
    for k in ks:
        for j in range(8):
            kin[j] = (k >> (32 * j)) & 0xFFFFFFFF
        emul.sbve_k256_split_lambda(kin, out)
        k1 = sum(out[j] << (32 * j) for j in range(8))
        k2 = sum(out[8 + j] << (32 * j) for j in range(8))
        s1, s2 = (-k1 if out[16] else k1), (-k2 if out[17] else k2)
        assert (s1 + s2 * lam - k) % n == 0, hex(k)
        assert k1 < 2 ** 128 and k2 < 2 ** 128, (hex(k), hex(k1), hex(k2))
        worst = max(worst, k1.bit_length(), k2.bit_length())
    assert worst == 128


def test_point_layer_and_comb_of_G_match_the_python_twin(emul):
    rng = random.Random(13)
    out = (ctypes.c_uint32 * 16)()
    for j, m in [(0, 1), (0, 2), (0, 32768), (1, 1), (7, 12345), (16, 1), (16, 32768), (15, 32767)]:
        emul.sbve_k256_g_entry(j, m, out)
        want = ec.pt_mul(m * 2**(16 * j) % N, ec.G)
        assert (val(out), val(out[8:])) == want, (j, m)
    Q = ec.pt_mul(rng.randrange(1, N), ec.G)
    cases = [(rng.randrange(N), rng.randrange(N)) for _ in range(6)] + [(0, 0), (1, 0), (0, 1), (2, 0), (N - 1, 0), (5, N - 5 * 1)]
    for k, l in cases:
        ok = emul.sbve_k256_mul2(words(k), words(Q[0]), words(Q[1]), words(l), out)
        want = ec.pt_add(ec.pt_mul(k, Q), ec.pt_mul(l, ec.G))
        assert (ok == 0) == (want is None)
        if want is not None:
            assert (val(out), val(out[8:])) == want, (k, l)
    # k Q + l G with Q = G: every addition of G onto a multiple of G, including G + G and G + (-G)
    for k, l in [(1, 1), (1, N - 1), (3, N - 3), (2, 2), (N - 2, 1)]:
        ok = emul.sbve_k256_mul2(words(k), words(ec.GX), words(ec.GY), words(l), out)
        want = ec.pt_mul((k + l) % N, ec.G)
        assert (ok == 0) == (want is None)
        if want is not None:
            assert (val(out), val(out[8:])) == want


# ---- the whole device path, emulated ----------------------------------------------------------------------------------------
def test_emulated_device_path_on_the_golden_vectors(emul, k256_vectors):
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in k256_vectors)
    n = len(k256_vectors)
    bm = ctypes.create_string_buffer((n + 7) // 8)
    emul.sbve_k256_verify_batch(blob, ctypes.c_size_t(n), bm)
    got = bits(bm.raw, n)
    bad = [v["name"] for v, g in zip(k256_vectors, got) if g != v["accept"]]
    assert not bad, bad


def test_emulated_device_path_equals_the_oracle_on_a_seeded_batch(emul, koracle):
    n = 1536
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer(n // 8)
    koracle.sbvo_k256_gen_batch(0xC0DE, n, 11, 3, tup, exp, 8)
    bm = ctypes.create_string_buffer(n // 8)
    emul.sbve_k256_verify_batch(tup.raw, ctypes.c_size_t(n), bm)
    assert bm.raw == exp.raw
    assert sum(bits(exp.raw, n)) == n - n // 3


def test_emulated_grouped_step_equals_the_oracle_and_the_one_lane_path(emul, koracle, k256_vectors):
    """k256_group.h (per-batch key combs built by the quad chain / rows on the isomorphic curve / affine fill, G phase and Q
    phase over the key-sorted list): the golden vectors replicated so that their keys take the table path, spliced into a seeded
    batch, in 1, 2 and 3 chunks of windows — verdicts equal the oracle's and the pinned ones, with the field's contract checks
    on; every key of the batch is grouped (threshold 2) except the corrupted ones."""
    emul.sbve_k256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    n = 360
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    koracle.sbvo_k256_gen_batch(0x256B, n, 12, 5, tup, exp, 4)
    blob = b"".join(bytes.fromhex(v["tuple"]) * 3 for v in k256_vectors)
    allt = tup.raw + blob
    total = len(allt) // 160
    want = bits(exp.raw, n) + [v["accept"] for v in k256_vectors for _ in range(3)]
    ob = ctypes.create_string_buffer((total + 7) // 8)
    koracle.sbvo_k256_verify_batch(allt, total, ob, 4)
    assert bits(ob.raw, total) == want
    stats = (ctypes.c_uint32 * 4)()
    for chunks in (2, 1, 3):
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_k256_verify_batch_grouped(allt, total, bm, 2, 512, 12, chunks, stats)
        got = bits(bm.raw, total)
        bad = [i for i in range(total) if got[i] != want[i]]
        assert not bad, (chunks, bad[:10])
        assert stats[0] >= 12 and stats[1] > 300 and stats[1] + stats[2] + stats[3] == total, list(stats)
    # threshold above every count: nothing is grouped, everything takes the one-lane kernel, same verdicts
    bm = ctypes.create_string_buffer((total + 7) // 8)
    emul.sbve_k256_verify_batch_grouped(allt, total, bm, 1000, 512, 12, 2, stats)
    assert bits(bm.raw, total) == want and stats[0] == 0
    # stage A with T tuples per inversion (k256_prep_chunk, GroupSync::k256_prep_t): golden vectors with r, s out of range sit
    # in the same product chains as valid ones; a ragged last workgroup (total is no multiple of 64 T)
    try:
        for T in (2, 4, 8):
            emul.sbve_set_k256_prep_t(T)
            bm = ctypes.create_string_buffer((total + 7) // 8)
            emul.sbve_k256_verify_batch_grouped(allt, total, bm, 2, 512, 12, 2, stats)
            got = bits(bm.raw, total)
            bad = [i for i in range(total) if got[i] != want[i]]
            assert not bad, (T, bad[:10])
    finally:
        emul.sbve_set_k256_prep_t(1)


def test_persistent_key_table_cache_of_this_curve_never_changes_verdicts(emul, koracle, k256_vectors):
    """sbv_key_cache(SBV_SCHEME_SECP256K1): the curve's own comb pool keeps the tables a grouped batch built.  Batch 1 is cold
    (every group misses, builds and stays), batch 2 over the same keys is warm (no chain / rows / fill lane runs for a hit), a
    cache too small for all keys overflows into the per-batch area, a key cached as INVALID (off the curve) stays rejected, a
    cached key is grouped however few of its signatures a batch carries, and switching the cache off forgets everything.
    Verdicts always equal the oracle's.  The P-256 cache is a different object: filling this one leaves that one empty."""
    emul.sbve_k256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    emul.sbve_scheme_key_cache.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
    off = next(bytes.fromhex(v["tuple"]) for v in k256_vectors if not v["accept"] and "off" in v["name"] and "curve" in v["name"])
    stats = (ctypes.c_uint32 * 4)()
    cstats = (ctypes.c_uint32 * 3)()

    def batch(seed, n, nkeys):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer((n + 7) // 8)
        koracle.sbvo_k256_gen_batch(seed, n, nkeys, 5, tup, exp, 4)
        return tup.raw, bits(exp.raw, n)

    def run(blob, want, chunks=2):
        total = len(blob) // 160
        ob = ctypes.create_string_buffer((total + 7) // 8)
        koracle.sbvo_k256_verify_batch(blob, total, ob, 4)
        assert bits(ob.raw, total) == want
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_k256_verify_batch_grouped(blob, total, bm, 8, 64, 12, chunks, stats)
        got = bits(bm.raw, total)
        assert got == want, [i for i in range(total) if got[i] != want[i]][:8]
        emul.sbve_scheme_key_cache_stats(1, cstats)
        return cstats[0], cstats[1], cstats[2]

    try:
        emul.sbve_scheme_key_cache(1, 1, 16)
        a, wa = batch(0x71, 400, 6)
        assert run(a + off * 20, wa + [False] * 20) == (7, 0, 7)          # 6 signer keys + the repeated off-curve key, all cold
        b, wb = batch(0x71, 500, 6)                                       # same seed -> same keys, other signatures
        assert run(b + off * 20, wb + [False] * 20, chunks=3) == (7, 7, 0)      # everything warm; the invalid key still rejected
        c, wc = batch(0x72, 600, 12)                                      # 12 new keys: 7 + 12 > 16 -> three overflow per batch
        assert run(c + a, wc + wa) == (16, 6, 12)
        assert run(c + a, wc + wa, chunks=1) == (16, 15, 3)               # the 9 that fitted are warm now, 3 stay cold every time
        d, wd = batch(0x71, 30, 6)                                        # 5 signatures per key: far below the threshold of 8
        entries, hits, misses = run(d, wd)
        assert hits >= 4 and misses == 0, (hits, misses)
        assert stats[1] >= 20 and stats[2] == 0 and stats[1] + stats[3] == 30, list(stats)
        emul.sbve_key_cache_stats(cstats)
        assert cstats[0] == 0                                             # the P-256 cache saw none of it
        emul.sbve_scheme_key_cache(1, 0, 16)                              # off: same verdicts, nothing cached, the small batch is generic again
        assert run(c + a, wc + wa)[1:] == (0, 0)
        run(d, wd)
        assert stats[1] == 0 and stats[2] >= 20
    finally:
        emul.sbve_scheme_key_cache(1, 0, 0)


def test_wide_comb_of_G_walk_matches_the_python_twin(emul):
    """The grouped step's G phase walks a `bits`-wide signed comb (20 bits on the device: 13 additions from 436 MB); the same
    recoding and table builder with 10-, 13- and 16-bit windows against u1 * G from big integers, edge scalars included."""
    emul.sbve_k256_gcomb_mul.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rng = random.Random(0x6C0)
    scalars = [0, 1, 2, N - 1, N - 2, (1 << 255), (1 << 256) - 1 - (1 << 32), 0x8000800080008000800080008000800080008000800080008000800080008000 % N,
               0x7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF] + [rng.randrange(N) for _ in range(24)]
    out = (ctypes.c_uint32 * 16)()
    for bits in (10, 13, 16):
        for u in scalars:
            got = emul.sbve_k256_gcomb_mul(words(u), bits, out)
            want = ec.pt_mul(u % N, ec.G) if u % N else None
            if want is None:
                assert got == 0
            else:
                assert got == 1 and (val(out[:8]), val(out[8:])) == want, (bits, hex(u))
