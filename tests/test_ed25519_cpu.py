"""CPU tier for the Ed25519 variant (BASELINE.json configs[4]): oracle vs RFC 8032 / golden vectors /
twin / OpenSSL, and the device algorithm (consensus_amd/csrc/ed25519_*.h) emulated lane by lane."""
import ctypes
import json
import os
import random

import pytest

import ed25519_py as ed
from test_emul_device_algo import emul, limbs, val  # noqa: F401  (fixture + helpers)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = ed.P


@pytest.fixture(scope="module")
def ed_vectors():
    return json.load(open(os.path.join(GOLDEN, "ed25519_vectors.json")))["vectors"]


def _tuples(vs):
    return b"".join(ed.pack_tuple(bytes.fromhex(v["pk"]), bytes.fromhex(v["msg"]), bytes.fromhex(v["sig"])) for v in vs)


def _bits(bm, n):
    return [bool((bm[i >> 3] >> (i & 7)) & 1) for i in range(n)]


def test_oracle_matches_vectors_and_openssl(oracle, openssl_check, ed_vectors):
    oracle.sbvo_ed25519_verify.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    openssl_check.sbvssl_ed25519_verify.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    assert len(ed_vectors) >= 80 and sum(v["class"] == "rfc8032" for v in ed_vectors) == 4
    n_ssl = 0
    for v in ed_vectors:
        pk, msg, sig = bytes.fromhex(v["pk"]), bytes.fromhex(v["msg"]), bytes.fromhex(v["sig"])
        assert bool(oracle.sbvo_ed25519_verify(pk, msg, len(msg), sig, len(sig))) == v["accept"], v["name"]
        if v["class"] in ("rfc8032", "honest", "bitflip"):
            assert bool(openssl_check.sbvssl_ed25519_verify(pk, msg, len(msg), sig)) == v["accept"], v["name"]
            n_ssl += 1
    assert n_ssl >= 40


POS = [(51 * i + 1) // 2 for i in range(10)]                 # limb i sits at bit ceil(25.5 i)
WID = [25 if i & 1 else 26 for i in range(10)]


def fval(limbs10):
    return sum(int(v) << POS[i] for i, v in enumerate(limbs10))


def flimbs(vals):
    return (ctypes.c_int32 * 10)(*vals)


def tight(out):
    return all(abs(int(out[i])) <= (1 << (WID[i] - 1)) + (1 << 19) for i in range(10))


def rand_limbs(rng, mult, corner=False):
    """limbs up to mult x tight (tight = 2^(w-1)), signed; corner: every limb at +- the bound"""
    out = []
    for i in range(10):
        b = mult * (1 << (WID[i] - 1))
        out.append(rng.choice((-b, b)) if corner else rng.randint(-b, b))
    return out


def test_fe25_ops_match_bigint(emul):
    """consensus_amd/csrc/ed25519_fe.h (10 signed limbs, radix 2^25.5) against Python integers, at and inside the operand contracts
    (the emulator build aborts on a contract violation: -DSBV_F25_CHECK)."""
    rng = random.Random(25519)
    out = (ctypes.c_int32 * 10)()
    w8 = (ctypes.c_uint32 * 8)()
    for it in range(400):
        corner = it < 40
        a, b, c = rand_limbs(rng, 8, corner), rand_limbs(rng, 3, corner), rand_limbs(rng, 3, corner)
        emul.sbve_fe25_mul(flimbs(a), flimbs(b), out)
        assert fval(out) % P == fval(a) * fval(b) % P and tight(out)
        emul.sbve_fe25_sqr(flimbs(c), out)
        assert fval(out) % P == fval(c) ** 2 % P and tight(out)
        emul.sbve_fe25_carry(flimbs(a), out)
        assert fval(out) % P == fval(a) % P and tight(out)
        emul.sbve_fe25_freeze(flimbs(a), w8)
        assert val(w8) == fval(a) % P
    # values around the modulus, through words -> limbs -> freeze
    for x in [0, 1, 18, 19, P - 1, P, P + 1, 2**255 - 1, 2**255 - 20, 2**254, (1 << 255) - (1 << 230)] + [rng.randrange(1 << 255) for _ in range(100)]:
        emul.sbve_fe25_from_words(limbs(x), out)
        assert fval(out) == x and all(0 <= int(out[i]) < (1 << WID[i]) for i in range(10))
        emul.sbve_fe25_freeze(out, w8)
        assert val(w8) == x % P
        neg = flimbs([-int(v) for v in out])
        emul.sbve_fe25_freeze(neg, w8)
        assert val(w8) == (-x) % P
    for x in [1, 2, P - 1, 2**255 - 1] + [rng.randrange(1, 1 << 255) for _ in range(10)]:
        if x % P == 0:
            continue
        emul.sbve_fe25_from_words(limbs(x), out)
        t = (ctypes.c_int32 * 10)()
        emul.sbve_fe25_carry(out, t)
        emul.sbve_fe25_inv(t, out)
        assert fval(out) % P == pow(x % P, -1, P)
        emul.sbve_fe25_inv_gcd(t, out)
        assert fval(out) % P == pow(x % P, -1, P)
    consts = (ctypes.c_int32 * 30)()
    emul.sbve_fe25_consts(consts)
    d = (-121665 * pow(121666, -1, P)) % P
    assert fval(consts[0:10]) == d and fval(consts[10:20]) == 2 * d % P and fval(consts[20:30]) == pow(2, (P - 1) // 4, P)


def test_device_algorithm_on_golden_vectors_and_random_batch(emul, oracle, ed_vectors):
    emul.sbve_ed25519_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    blob = _tuples(ed_vectors)
    n = len(ed_vectors)
    bm = ctypes.create_string_buffer((n + 7) // 8)
    emul.sbve_ed25519_verify_batch(blob, n, bm)
    bad = [v["name"] for v, g in zip(ed_vectors, _bits(bm.raw, n)) if g != v["accept"]]
    assert not bad, bad
    # tuples with k >= L or S >= L are rejected
    t = bytearray(blob[:128])
    t[96:128] = ed.L.to_bytes(32, "little")
    emul.sbve_ed25519_verify_batch(bytes(t), 1, bm)
    assert not _bits(bm.raw, 1)[0]
    # seeded batch vs the C oracle
    oracle.sbvo_ed25519_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int]
    m = 700
    tup = ctypes.create_string_buffer(128 * m)
    exp = ctypes.create_string_buffer((m + 7) // 8)
    oracle.sbvo_ed25519_gen_batch(0xED, m, 11, 3, tup, exp, 4)
    bm = ctypes.create_string_buffer((m + 7) // 8)
    emul.sbve_ed25519_verify_batch(tup.raw, m, bm)
    assert bm.raw == exp.raw
    assert sum(_bits(exp.raw, m)) == m - m // 3


def test_grouped_by_key_inside_the_batch_matches_plain_kernel(emul, oracle, ed_vectors):
    """ed25519_group.h: hash grouping by A + per-batch combs of -A + [S]B / [k](-A) phases == the one-lane verdicts,
    on the golden vectors (non-canonical and small-order keys, S >= L, ...), a seeded batch, an undecompressable key
    that repeats, and across chunk counts / lanes per window / thresholds (sampled and exact counts)."""
    emul.sbve_ed25519_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                                                       ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    oracle.sbvo_ed25519_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int]
    blob = _tuples(ed_vectors)
    m = 900
    tup = ctypes.create_string_buffer(128 * m)
    exp = ctypes.create_string_buffer((m + 7) // 8)
    oracle.sbvo_ed25519_gen_batch(0xED25, m, 7, 5, tup, exp, 4)
    # a key that is not a point (y = 2: x^2 = 3 / (4d + 1) is a non-residue?) -- find one, then repeat it 40 times
    bad_key = None
    for y in range(2, 60):
        cand = y.to_bytes(32, "little")
        if ed.decompress(cand) is None:
            bad_key = cand
            break
    assert bad_key is not None
    t0 = bytearray(tup.raw[:128])
    t0[64:96] = bad_key
    allt = blob + tup.raw + bytes(t0) * 40
    total = len(allt) // 128
    want = [v["accept"] for v in ed_vectors] + _bits(exp.raw, m) + [False] * 40
    stats = (ctypes.c_uint32 * 4)()
    emul.sbve_group_sort_violations.restype = ctypes.c_ulong
    emul.sbve_ed_chain_mismatches.restype = ctypes.c_ulong
    emul.sbve_ed_quad_mismatches.restype = ctypes.c_ulong
    violations = emul.sbve_group_sort_violations()
    for min_count, max_groups, ht_bits, chunks, parts in [(8, 64, 12, 2, 4), (8, 64, 12, 1, 8), (64, 64, 12, 4, 16), (1, 4096, 12, 3, 2),
                                                          (8, 3, 12, 2, 4), (2, 64, 11, 2, 4), (10**6, 64, 12, 2, 4)]:
        # the key-sorted grouped list (tuple-major accumulator records, runs of equal keys) and the split's compaction order
        res = []
        for sort in (0, 1):
            emul.sbve_set_group_sort(sort)
            bm = ctypes.create_string_buffer((total + 7) // 8)
            emul.sbve_ed25519_verify_batch_grouped(allt, total, bm, min_count, max_groups, ht_bits, chunks, parts, stats)
            res.append((bm.raw, tuple(stats)))
        # same verdicts; the sorted step also rejects the ungrouped candidates whose key is no point before the one-lane kernel
        # (ed_group_keycheck_lane, round 4), so its "ungrouped" and "rejected for the key" counts differ from the split's by exactly those
        assert res[0][0] == res[1][0], (min_count, max_groups, ht_bits, chunks, parts)
        (_, g0, u0, r0), (_, g1, u1, r1) = res[0][1], res[1][1]
        assert g0 == g1 and u0 + r0 == u1 + r1 == total - g1 and r0 == 0 and u1 <= u0, (res[0][1], res[1][1])
        if min_count == 64:
            assert r1 >= 40                   # the 40 copies of the undecompressable key are below this threshold: candidates, then rejected
        assert emul.sbve_group_sort_violations() == violations
        assert emul.sbve_ed_chain_mismatches() == 0      # round 5: the quad-lane base chain records byte for byte what the one-lane chain does
        assert emul.sbve_ed_quad_mismatches() == 0       # round 6: the quad form of the one-lane kernel (key-sorted step) gives the one-lane verdicts
        got = _bits(bm.raw, total)
        bad = [i for i in range(total) if got[i] != want[i]]
        assert not bad, (min_count, max_groups, ht_bits, chunks, parts, bad[:8])
        assert stats[1] + stats[2] + stats[3] == total
        if min_count == 8 and max_groups == 64:
            assert stats[0] >= 8 and stats[1] > 800
        if max_groups == 3:
            assert stats[0] == 3
        if min_count == 10**6:
            # nothing grouped: every tuple a candidate, the keys that are no points rejected before the one-lane kernel (sorted step: stats of the last run)
            assert stats[0] == 0 and stats[2] + stats[3] == total and stats[3] >= 40


def test_persistent_key_table_cache_of_this_scheme_never_changes_verdicts(emul, oracle, ed_vectors):
    """sbv_key_cache(SBV_SCHEME_ED25519): the combs of -A a grouped batch builds are kept in the scheme's own pool (slots found
    by the 32 key bytes).  Cold batch, warm batch over the same keys (key-sorted and compaction order), overflow of a small
    cache, an undecompressable key cached as invalid, a warm key grouped below the count threshold, the golden vectors' odd keys
    (non-canonical, small order) through cached tables; verdicts always equal the one-lane kernel's and the generator's."""
    emul.sbve_ed25519_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                                                       ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    emul.sbve_scheme_key_cache.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
    oracle.sbvo_ed25519_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int]
    bad_key = next(y.to_bytes(32, "little") for y in range(2, 60) if ed.decompress(y.to_bytes(32, "little")) is None)
    stats = (ctypes.c_uint32 * 4)()
    cstats = (ctypes.c_uint32 * 3)()

    def batch(seed, m, nkeys):
        tup = ctypes.create_string_buffer(128 * m)
        exp = ctypes.create_string_buffer((m + 7) // 8)
        oracle.sbvo_ed25519_gen_batch(seed, m, nkeys, 5, tup, exp, 4)
        return tup.raw, _bits(exp.raw, m)

    def run(blob, want, chunks=2, sort=1):
        total = len(blob) // 128
        plain = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_ed25519_verify_batch(blob, ctypes.c_size_t(total), plain)
        assert _bits(plain.raw, total) == want
        emul.sbve_set_group_sort(sort)
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_ed25519_verify_batch_grouped(blob, total, bm, 8, 64, 12, chunks, 4, stats)
        emul.sbve_set_group_sort(1)
        got = _bits(bm.raw, total)
        assert got == want, [i for i in range(total) if got[i] != want[i]][:8]
        emul.sbve_scheme_key_cache_stats(2, cstats)
        return cstats[0], cstats[1], cstats[2]

    try:
        emul.sbve_scheme_key_cache(2, 1, 16)
        a, wa = batch(0xE1, 300, 6)
        t0 = bytearray(a[:128])
        t0[64:96] = bad_key
        junk = bytes(t0) * 20
        assert run(a + junk, wa + [False] * 20) == (7, 0, 7)
        b, wb = batch(0xE1, 350, 6)
        assert run(b + junk, wb + [False] * 20, chunks=4) == (7, 7, 0)
        assert run(b + junk, wb + [False] * 20, chunks=1, sort=0) == (7, 7, 0)
        c, wc = batch(0xE2, 500, 12)
        assert run(c + a, wc + wa) == (16, 6, 12)
        assert run(c + a, wc + wa) == (16, 15, 3)
        d, wd = batch(0xE1, 30, 6)
        entries, hits, misses = run(d, wd)
        assert hits >= 4 and misses == 0 and stats[1] >= 20, (hits, misses, list(stats))
        emul.sbve_scheme_key_cache(2, 1, 64)                  # a fresh cache: the golden vectors' keys, each 9 times, twice
        blob = _tuples(ed_vectors) * 9
        want = [v["accept"] for v in ed_vectors] * 9
        e1 = run(blob, want)
        e2 = run(blob, want)
        assert e1[1] == 0 and e1[2] > 5 and e2[2] == 0 and e2[1] == e1[2], (e1, e2)
        emul.sbve_scheme_key_cache(2, 0, 64)
        assert run(blob, want)[1:] == (0, 0)
    finally:
        emul.sbve_scheme_key_cache(2, 0, 0)


def test_hot_keys_of_this_scheme_never_change_verdicts(emul, oracle, ed_vectors):
    """ed25519_group.h "hot keys" (round 6): a cache slot whose count passes the threshold gets a 16-bit comb of -A, built
    lane by lane (ed_widetab_lane) from base points its 8-bit comb holds; later batches serve the wavefronts whose lanes are all
    hot from it (ed_qphase_wide_lane).  Every promoted comb equals the host builder's entry by entry; verdicts equal the one-lane
    kernel's and the generator's before, during and after promotion, also for the golden vectors' odd keys (non-canonical
    encodings, points of small order — whose multiples pass through the identity); a full pool hands a comb to a hotter key
    (the P-256 step's eviction rule on this scheme's arrays); switching the pool off leaves the verdicts where they were."""
    emul.sbve_ed25519_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                                                       ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    emul.sbve_scheme_key_cache.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
    emul.sbve_ed_hot_keys.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    emul.sbve_ed_hot_stats.argtypes = [ctypes.c_void_p]
    emul.sbve_ed_hot_comb_mismatches.argtypes = [ctypes.c_uint32]
    emul.sbve_ed_hot_comb_mismatches.restype = ctypes.c_size_t
    emul.sbve_ed_hot_wide_of_key.argtypes = [ctypes.c_char_p]
    emul.sbve_ed_hot_wide_of_key.restype = ctypes.c_uint32
    emul.sbve_ed_hot_hits_of_key.argtypes = [ctypes.c_char_p]
    emul.sbve_ed_hot_hits_of_key.restype = ctypes.c_uint32
    oracle.sbvo_ed25519_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int]
    NONE = 0xFFFFFFFF
    hs = (ctypes.c_uint32 * 6)()

    def batch(seed, m, nkeys):
        tup = ctypes.create_string_buffer(128 * m)
        exp = ctypes.create_string_buffer((m + 7) // 8)
        oracle.sbvo_ed25519_gen_batch(seed, m, nkeys, 5, tup, exp, 4)
        return tup.raw, _bits(exp.raw, m)

    def keys_of(blob):                                     # the signers (a corrupted key is a key of its own, seen once)
        seen = {}
        for i in range(len(blob) // 128):
            seen[blob[128 * i + 64:128 * i + 96]] = seen.get(blob[128 * i + 64:128 * i + 96], 0) + 1
        return sorted(k for k, c in seen.items() if c >= 8)

    def run(blob, want, chunks=2, plain_check=False):
        total = len(blob) // 128
        if plain_check:
            plain = ctypes.create_string_buffer((total + 7) // 8)
            emul.sbve_ed25519_verify_batch(blob, ctypes.c_size_t(total), plain)
            assert _bits(plain.raw, total) == want
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_ed25519_verify_batch_grouped(blob, total, bm, 8, 64, 12, chunks, 4, None)
        got = _bits(bm.raw, total)
        assert got == want, [i for i in range(total) if got[i] != want[i]][:8]
        emul.sbve_ed_hot_stats(hs)
        return list(hs)

    try:
        emul.sbve_scheme_key_cache(2, 1, 16)
        emul.sbve_ed_hot_keys(2, 250)                     # a pool of two combs, promotion from 250 grouped tuples on
        a, wa = batch(0xA7, 900, 3)                       # three keys, 300 tuples each, every 5th corrupted
        ka = keys_of(a)
        assert len(ka) == 3
        st = run(a, wa, plain_check=True)                 # cold: all three pass the threshold, two combs exist
        assert st[0] == 2 and st[2] == 0 and st[4] == 0, st
        assert emul.sbve_ed_hot_comb_mismatches(0) == 0 and emul.sbve_ed_hot_comb_mismatches(1) == 0
        assert emul.sbve_ed_hot_comb_mismatches(2) == ctypes.c_size_t(-1).value
        owners = [k for k in ka if emul.sbve_ed_hot_wide_of_key(k) != NONE]
        assert len(owners) == 2
        b, wb = batch(0xA7, 960, 3)
        st = run(b, wb, chunks=4, plain_check=True)       # warm: the two owners' wavefronts take the wide pass
        assert 256 <= st[2] <= 640 and st[0] == 2 and st[4] == 0, st      # a similar key never takes a comb away (hysteresis)
        # the third key alone, hot: twice its rivals' count is enough for the comb of the colder owner
        loser = next(k for k in ka if k not in owners)
        only = b"".join(b[128 * i:128 * i + 128] for i in range(960) if b[128 * i + 64:128 * i + 96] == loser)
        wonly = [wb[i] for i in range(960) if b[128 * i + 64:128 * i + 96] == loser]
        for _ in range(3):
            st = run(only * 2, wonly * 2)
            if st[4]:
                break
        assert st[4] == 1 and st[0] == 2, st
        w = emul.sbve_ed_hot_wide_of_key(loser)
        assert w in (0, 1) and emul.sbve_ed_hot_comb_mismatches(w) == 0
        assert sum(1 for k in ka if emul.sbve_ed_hot_wide_of_key(k) != NONE) == 2
        st = run(b, wb)                                   # everybody again: the victim is served from its 8-bit comb
        assert st[2] >= 256, st
        # the golden vectors' keys, 40 tuples each, promoted from 30 on
        emul.sbve_scheme_key_cache(2, 1, 64)
        emul.sbve_ed_hot_keys(6, 30)
        blob = _tuples(ed_vectors) * 40
        want = [v["accept"] for v in ed_vectors] * 40
        s1 = run(blob, want, plain_check=True)
        s2 = run(blob, want)
        assert 1 <= s1[0] <= 6 and s2[2] >= 64, (s1, s2)
        for i in range(s2[0]):
            assert emul.sbve_ed_hot_comb_mismatches(i) == 0, i
        emul.sbve_ed_hot_keys(0, 30)                      # off: same verdicts, nothing served wide
        assert run(blob, want)[2] == 0
    finally:
        emul.sbve_ed_hot_keys(0, 4096)
        emul.sbve_scheme_key_cache(2, 0, 0)


def test_device_message_front_end_sha512_and_mod_l(emul):
    """sha512_dev.h: the 512-bit reduction mod L against big ints (edge values: 0, L-1, L, 2L, 2^252 multiples, 2^512-1)
    and the whole lane — SHA-512(R|A|M) mod L — against hashlib for message lengths around the block boundaries."""
    import hashlib
    rng = random.Random(99)
    L = ed.L
    out = (ctypes.c_uint32 * 8)()
    edge = [0, 1, L - 1, L, L + 1, 2 * L, 2**252, 2**252 - 1, 2**253, 2**256 - 1, 2**256, 2**511, 2**512 - 1, (2**512 - 1) // L * L,
            (2**512 - 1) // L * L - 1, L << 160, (L << 160) - 1, L << 64, 2**385, 2**414 - 1]
    for x in edge + [rng.getrandbits(512) for _ in range(400)] + [rng.getrandbits(k) for k in (253, 260, 300, 386, 450) for _ in range(20)]:
        arr = (ctypes.c_uint32 * 16)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(16)])
        emul.sbve_mod_l_512(arr, out)
        assert val(out) == x % L, hex(x)
    emul.sbve_ed_msg_frontend.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    t = ctypes.create_string_buffer(128)
    for mlen in [0, 1, 31, 46, 47, 48, 63, 64, 111, 112, 113, 127, 128, 129, 175, 176, 300, 1000]:
        sig = bytes(rng.getrandbits(8) for _ in range(64))
        pk = bytes(rng.getrandbits(8) for _ in range(32))
        msg = bytes(rng.getrandbits(8) for _ in range(mlen))
        emul.sbve_ed_msg_frontend(sig, pk, msg, mlen, t)
        k = int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % L
        assert t.raw == sig + pk + k.to_bytes(32, "little"), mlen


def test_openssl_batch_check_agrees_with_the_generator_and_the_oracle_verifier(oracle, openssl_check):
    """oracle/openssl_check.c: sbvssl_ed25519_verify_gen_batch rebuilds the generator's messages from (seed, index) and runs
    EVP_DigestVerify on what the tuples carry — the third opinion the GPU tier holds the full-size configs[4] bitmap against.
    Here: whole batch and an offset slice, against the generator's flags and the oracle's verifier."""
    n, seed = 3000, 0x5B7F2026
    oracle.sbvo_ed25519_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int]
    oracle.sbvo_ed25519_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    openssl_check.sbvssl_ed25519_verify_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t,
                                                              ctypes.c_void_p, ctypes.c_int]
    tup = ctypes.create_string_buffer(128 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_ed25519_gen_batch(seed, n, 37, 4, tup, exp, 4)
    want = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_ed25519_verify_batch(tup.raw, n, want, 4)
    ssl = ctypes.create_string_buffer((n + 7) // 8)
    openssl_check.sbvssl_ed25519_verify_gen_batch(seed, tup.raw, 0, n, ssl, 4)
    assert ssl.raw == want.raw == exp.raw
    assert sum(bin(b).count("1") for b in ssl.raw) == n - n // 4
    part = ctypes.create_string_buffer(100)
    openssl_check.sbvssl_ed25519_verify_gen_batch(seed, tup.raw[128 * 1600:128 * 2400], 1600, 800, part, 3)
    assert part.raw == exp.raw[200:300]
    wrong_seed = ctypes.create_string_buffer((n + 7) // 8)
    openssl_check.sbvssl_ed25519_verify_gen_batch(seed + 1, tup.raw, 0, n, wrong_seed, 4)
    assert wrong_seed.raw == bytes((n + 7) // 8)                     # other messages: nothing verifies


def test_comb_of_B_at_other_widths(emul, oracle, ed_vectors):
    """Round 5: the grouped step's comb of B is `bits` wide (ed25519_group.h: edcomb; libsbv default 20 bits = 13 additions instead of
    16).  [S]B through the comb walk == the big-int twin for edge scalars (0, 1, L - 1, 2^252, every digit at its extreme, the top window
    alone) and random ones at 12, 13, 19 and 20 bits; the grouped emulation over the golden vectors + a seeded batch gives the same
    bitmap at each width."""
    import random
    emul.sbve_set_ed_b_bits.argtypes = [ctypes.c_int]
    emul.sbve_ed_comb_mul.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    emul.sbve_ed25519_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                                                       ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = random.Random(0xB175)
    blob = _tuples(ed_vectors)
    m = 500
    tup = ctypes.create_string_buffer(128 * m)
    exp = ctypes.create_string_buffer((m + 7) // 8)
    oracle.sbvo_ed25519_gen_batch(0xED26, m, 5, 4, tup, exp, 4)
    allt = blob + tup.raw
    total = len(allt) // 128
    want = [v["accept"] for v in ed_vectors] + _bits(exp.raw, m)
    try:
        for bits in (12, 13, 19, 20, 16):
            emul.sbve_set_ed_b_bits(bits)
            half = 1 << (bits - 1)
            windows = -(-254 // bits)
            scalars = [0, 1, 2, ed.L - 1, ed.L - 2, 1 << 252, (1 << 252) - 1, (1 << 253) - 1,
                       half, half - 1, half + 1, (1 << bits) - 1, 1 << bits,
                       sum(half << (bits * j) for j in range(windows)) % (1 << 253),          # every digit -> 0 after the offset: skip everywhere
                       sum((half - 1) << (bits * j) for j in range(windows)) % (1 << 253),     # every digit -1
                       sum(((1 << bits) - 1) << (bits * j) for j in range(windows)) % (1 << 253),
                       1 << (bits * (windows - 1)), (1 << 253) - (1 << (bits * (windows - 1)))]
            scalars += [rng.randrange(1 << 253) for _ in range(24 if bits < 19 else 8)]
            out = ctypes.create_string_buffer(32)
            for sc in scalars:
                emul.sbve_ed_comb_mul((ctypes.c_uint32 * 8)(*[(sc >> (32 * i)) & 0xFFFFFFFF for i in range(8)]), out)
                assert out.raw == ed.encode(ed.pt_mul(sc, ed.B)), (bits, hex(sc))
            bm = ctypes.create_string_buffer((total + 7) // 8)
            emul.sbve_ed25519_verify_batch_grouped(allt, total, bm, 8, 64, 12, 2, 4, None)
            got = _bits(bm.raw, total)
            assert got == want, (bits, [i for i in range(total) if got[i] != want[i]][:8])
    finally:
        emul.sbve_set_ed_b_bits(16)
