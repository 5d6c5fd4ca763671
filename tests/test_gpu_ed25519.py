"""GPU tier for the Ed25519 variant (BASELINE.json configs[4]): HIP path through the C-ABI vs the oracle."""
import ctypes
import json
import os

import pytest

import consensus_amd as sbv
import ed25519_py as ed

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gpu():
    sbv.init(0)
    yield sbv


def _gen(oracle, seed, n, nkeys, inv):
    oracle.sbvo_ed25519_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int]
    tup = ctypes.create_string_buffer(128 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_ed25519_gen_batch(seed, n, nkeys, inv, tup, exp, os.cpu_count() or 1)
    return tup, exp


def _oracle_verdicts(oracle, tup, n):
    """The oracle's VERIFIER (oracle/ed25519_oracle.c: sbvo_ed25519_verify_batch) on the tuples themselves."""
    oracle.sbvo_ed25519_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    want = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_ed25519_verify_batch(tup, n, want, os.cpu_count() or 1)
    return want.raw


def _openssl_verdicts(openssl_check, seed, tup, n, first=0):
    """OpenSSL EVP_DigestVerify on (A, message, R|S) of every tuple (oracle/openssl_check.c: the generator's messages are
    rebuilt from the seed and the tuple's index)."""
    openssl_check.sbvssl_ed25519_verify_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t,
                                                              ctypes.c_void_p, ctypes.c_int]
    want = ctypes.create_string_buffer((n + 7) // 8)
    openssl_check.sbvssl_ed25519_verify_gen_batch(seed, tup, first, n, want, os.cpu_count() or 1)
    return want.raw


def test_golden_vectors_via_host_tuple_builder(gpu):
    vs = json.load(open(os.path.join(GOLDEN, "ed25519_vectors.json")))["vectors"]
    tuples = gpu.ed25519_make_tuples([bytes.fromhex(v["sig"]) for v in vs], [bytes.fromhex(v["pk"]) for v in vs],
                                     [bytes.fromhex(v["msg"]) for v in vs])
    assert tuples == b"".join(ed.pack_tuple(bytes.fromhex(v["pk"]), bytes.fromhex(v["msg"]), bytes.fromhex(v["sig"])) for v in vs)
    got = sbv.bitmap_to_list(gpu.ed25519_verify_batch(tuples), len(vs))
    bad = [v["name"] for v, g in zip(vs, got) if g != v["accept"]]
    assert not bad, bad
    # non-canonical scalars are rejected on the device
    t = bytearray(tuples[:128]); t[96:128] = ed.L.to_bytes(32, "little")
    assert gpu.ed25519_verify_batch(bytes(t)) == b"\x00"


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 1000, 20000])
def test_ragged_sizes_match_oracle(gpu, oracle, openssl_check, n):
    """Ragged batch sizes against three opinions: the generator's by-construction flags, the oracle's verifier run on the
    tuples, and OpenSSL on (key, message, signature)."""
    tup, exp = _gen(oracle, 0xE000 + n, n, 21, 3)
    got = gpu.ed25519_verify_batch(tup.raw, n)
    assert got == exp.raw[:(n + 7) // 8]
    assert got == _oracle_verdicts(oracle, tup.raw, n)
    assert got == _openssl_verdicts(openssl_check, 0xE000 + n, tup.raw, n)


def test_garbage(gpu, oracle):
    import random
    rng = random.Random(5)
    n = 2000
    junk = bytes(rng.getrandbits(8) for _ in range(128 * n))
    oracle.sbvo_ed25519_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    want = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_ed25519_verify_batch(junk, n, want, os.cpu_count() or 1)
    assert gpu.ed25519_verify_batch(junk, n) == want.raw == bytes((n + 7) // 8)


def test_full_batch_2_20(gpu, oracle, openssl_check):
    """configs[4]: 2^20 Ed25519 signatures, 1024 keys, 7/8 valid.  The WHOLE device bitmap is diffed against (i) the generator's
    by-construction flags, (ii) the oracle's verifier run over all 2^20 tuples and (iii) OpenSSL EVP_DigestVerify over all 2^20
    (key, message, signature) triples (until round 4 only (i) saw the full batch)."""
    n = 1 << 20
    tup, exp = _gen(oracle, 0x5B7F2026, n, 1024, 8)
    got = ctypes.create_string_buffer(n // 8)
    sbv._check(sbv.load().sbv_ed25519_verify_batch(ctypes.addressof(tup), n, ctypes.addressof(got)))
    assert got.raw == exp.raw
    want = _oracle_verdicts(oracle, tup.raw, n)
    assert got.raw == want, [i for i in range(n // 8) if got.raw[i] != want[i]][:8]
    ssl = _openssl_verdicts(openssl_check, 0x5B7F2026, tup.raw, n)
    assert got.raw == ssl, [i for i in range(n // 8) if got.raw[i] != ssl[i]][:8]
    assert sum(bin(b).count("1") for b in got.raw) == n - n // 8
    tm = gpu.last_timing()
    print(f"\n[ed25519 2^20] h2d {tm.h2d_us:.0f} us  verify {tm.verify_us:.0f} us -> {n / tm.verify_us:.1f} M verifies/s (kernel); "
          f"roofline: {128.125 * n / (tm.verify_us * 1e-6) / 1e9:.2f} GB/s algorithmic of 8000")


def test_in_step_key_grouping_equals_plain_kernel(gpu, oracle):
    """ed25519_group.h through the C-ABI: grouping by A, per-batch combs of -A, [S]B / [k](-A) phases — verdicts must
    equal the one-lane kernel's on the golden vectors (small-order / non-canonical keys, S >= L), a seeded batch and a
    key that is not a point repeated 40 times, in every configuration of thresholds and table slots."""
    vs = json.load(open(os.path.join(GOLDEN, "ed25519_vectors.json")))["vectors"]
    blob = b"".join(ed.pack_tuple(bytes.fromhex(v["pk"]), bytes.fromhex(v["msg"]), bytes.fromhex(v["sig"])) for v in vs)
    m = 3000
    tup, exp = _gen(oracle, 0xED25, m, 9, 5)
    bad_key = next(y.to_bytes(32, "little") for y in range(2, 60) if ed.decompress(y.to_bytes(32, "little")) is None)
    t0 = bytearray(tup.raw[:128]); t0[64:96] = bad_key
    allt = blob + tup.raw + bytes(t0) * 40
    total = len(allt) // 128
    want = [v["accept"] for v in vs] + sbv.bitmap_to_list(exp.raw, m) + [False] * 40
    try:
        gpu.set_grouping(False)
        plain = sbv.bitmap_to_list(gpu.ed25519_verify_batch(allt, total), total)
        assert plain == want
        for min_count, max_groups in [(8, 64), (1, 4096), (8, 3), (64, 64), (1000000, 64)]:
            gpu.set_grouping(True, 1, min_count, max_groups)
            got = sbv.bitmap_to_list(gpu.ed25519_verify_batch(allt, total), total)
            bad = [i for i in range(total) if got[i] != want[i]]
            assert not bad, (min_count, max_groups, bad[:8])
    finally:
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)


def test_grouped_vs_plain_full_batch(gpu, oracle):
    """2^20 signatures, 1024 keys: the grouped step (default at this size) and the one-lane kernel give the generator's
    verdicts; prints both kernel times."""
    n = 1 << 20
    tup, exp = _gen(oracle, 0x5B7F2026, n, 1024, 8)
    times = {}
    try:
        for mode in (True, False):
            gpu.set_grouping(mode, 131072, 64, 2048)
            got = ctypes.create_string_buffer(n // 8)
            sbv._check(sbv.load().sbv_ed25519_verify_batch(ctypes.addressof(tup), n, ctypes.addressof(got)))
            assert got.raw == exp.raw, mode
            times[mode] = gpu.last_timing().verify_us
    finally:
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)
    print(f"\n[ed25519 2^20] grouped {times[True]:.0f} us ({n / times[True]:.1f} M/s) | one lane per signature {times[False]:.0f} us "
          f"({n / times[False]:.1f} M/s)")


def test_hot_keys_of_this_scheme_on_the_device(gpu, oracle):
    """sbv_ed25519_hot_keys (ed25519_group.h "hot keys", round 6): cache slots that keep signing get 16-bit combs of -A, built on the
    device behind a batch's verdicts (64 promotions per batch at most); the wavefronts whose lanes are all hot take the wide pass.
    Verdicts equal the generator's in every batch — cold, during the promotions, served wide, with the pool full and with the
    feature switched off again — and promoted combs equal the host builder's, entry by entry.  Also the golden vectors' odd keys
    (non-canonical encodings, small order) promoted from 30 uses on."""
    n, nkeys = 1 << 18, 96
    tup, exp = _gen(oracle, 0x60D5EED, n, nkeys, 8)
    lib = sbv.load()

    def run(t=tup, m=n, want=exp.raw):
        got = ctypes.create_string_buffer((m + 7) // 8)
        sbv._check(lib.sbv_ed25519_verify_batch(ctypes.addressof(t), m, ctypes.addressof(got)))
        assert got.raw == want
        return gpu.ed_hot_key_stats(), gpu.last_timing().verify_us

    try:
        gpu.ed_hot_keys(64, 1024)
        (promoted, cap, wide, min_hits), t_cold = run()          # cold: the keys' 8-bit combs are built, nobody is served wide yet
        if cap == 0:
            pytest.skip("no room for the pool on this device")
        assert cap == 64 and min_hits == 1024 and wide == 0 and promoted == 64, (promoted, cap, wide)
        (promoted, _, wide, _), t_hot = run()                    # 64 of the 96 signers are served from their combs
        assert promoted == 64 and n // 2 < wide < n * 3 // 4, (promoted, wide)
        for i in (0, 31, 63):
            assert gpu.ed_hot_selfcheck(i), i
        for _ in range(3):                                       # the pool stays full: similar keys never trade combs; verdicts stay
            (promoted, _, wide2, _), _ = run()
        assert promoted == 64 and n // 2 < wide2 < n * 3 // 4
        assert gpu.ed_hot_selfcheck(5)
        gpu.ed_hot_keys(0, 1024)
        (promoted, cap, wide, _), t_warm = run()                 # off: the pool is gone (and with it this scheme's cache: a cold batch)
        assert (promoted, cap, wide) == (0, 0, 0)
        (_, _, _, _), t_warm = run()
        print(f"\n[ed25519 2^18 over {nkeys} signers] cold {t_cold:.0f} us | 64 hot signers {t_hot:.0f} us | cache only {t_warm:.0f} us")
        # the golden vectors' keys through combs of their own
        vs = json.load(open(os.path.join(GOLDEN, "ed25519_vectors.json")))["vectors"]
        blob = b"".join(ed.pack_tuple(bytes.fromhex(v["pk"]), bytes.fromhex(v["msg"]), bytes.fromhex(v["sig"])) for v in vs) * 40
        want = [v["accept"] for v in vs] * 40
        gpu.set_grouping(True, 1, 8, 64)
        gpu.ed_hot_keys(64, 30)
        for rnd in range(3):
            got = sbv.bitmap_to_list(gpu.ed25519_verify_batch(blob, len(want)), len(want))
            assert got == want, (rnd, [i for i in range(len(want)) if got[i] != want[i]][:8])
        promoted, cap, wide, _ = gpu.ed_hot_key_stats()
        assert 4 <= promoted <= 64, (promoted, cap, wide)
        print(f"[ed25519 golden keys x 40] {promoted} promoted, {wide} of {len(want)} tuples served wide")
        for i in range(promoted):
            assert gpu.ed_hot_selfcheck(i), i
    finally:
        gpu.ed_hot_keys(1024, 4096)                              # the library's default
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)


def test_hot_key_pool_of_this_scheme_follows_a_changing_signer_set(gpu, oracle):
    """The life cycle of the hot keys (p256_group.h: decay, eviction with hysteresis — the shared kernels k_hot_decay / k_promote_select /
    k_promote_evict / k_promote_publish) on this scheme's pool, on the device: 8 combs, three disjoint sets of 8 signers one after
    another.  The set that signs now takes the pool over from the set that stopped; every comb a new owner got equals the host
    builder's comb of -A for ITS key (sbv_ed25519_hot_selfcheck); every verdict of every batch is the generator's."""
    n = 8 * 8192
    sets = [_gen(oracle, 0xE71 + k, n, 8, 9) for k in range(3)]
    lib = sbv.load()

    def run(tup, exp):
        got = ctypes.create_string_buffer((n + 7) // 8)
        sbv._check(lib.sbv_ed25519_verify_batch(ctypes.addressof(tup), n, ctypes.addressof(got)))
        assert got.raw == exp.raw
        return gpu.ed_hot_key_stats()

    try:
        gpu.set_grouping(True, 1, 8, 64)
        gpu.ed_hot_keys(8, 4096)
        for k, (tup, exp) in enumerate(sets):
            settled_at = None
            for call in range(48):
                h = run(tup, exp)
                if h[2] >= n * 0.80:                           # (nearly) the whole batch through the wide pass: this set owns the pool
                    settled_at = call
                    break
            assert settled_at is not None and (settled_at <= 2 if k == 0 else 3 <= settled_at), (k, settled_at)
            for _ in range(2):
                h = run(tup, exp)
            assert h[0] == 8 and h[1] == 8 and h[2] >= n * 0.80, (k, h)
            for i in range(8):
                assert gpu.ed_hot_selfcheck(i), (k, i)
    finally:
        gpu.ed_hot_keys(1024, 4096)
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)


def test_quad_form_of_the_one_lane_kernel_gives_the_same_verdicts():
    """SBV_ED_UNGROUPED_QUAD=1 (round 6): the ungrouped list of the key-sorted step on four lanes per tuple (ed25519_group.h:
    ed25519_verify_quad — doublings and additions two multiplications deep, DPP broadcasts inside the quad, the key as the key check
    decompressed it).  Built to shorten the chain that gates a warm step; measured: the step is bound by the ALU work the ungrouped
    tuples and the Q phase share, not by the chain's length, and the quad form adds work (profiles/r06/ab_ed_ungrouped_quad_*_r06ac.jsonl:
    warm 2.69 -> 2.78 ms) — it stays opt-in.  Here its verdicts in a process of its own (the switch is read once): the golden
    vectors (small-order and non-canonical keys, S >= L) and a batch of 40 000 tuples over 30 000 keys, all ungrouped."""
    import subprocess
    import sys
    code = r"""
import ctypes, json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import consensus_amd as sbv
import ed25519_py as ed
oracle = ctypes.CDLL(os.path.join("oracle", "libsbv_oracle.so"))
oracle.sbvo_ed25519_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
vs = json.load(open(os.path.join("tests", "golden", "ed25519_vectors.json")))["vectors"]
blob = b"".join(ed.pack_tuple(bytes.fromhex(v["pk"]), bytes.fromhex(v["msg"]), bytes.fromhex(v["sig"])) for v in vs)
want = [v["accept"] for v in vs]
n = 40000
tup = ctypes.create_string_buffer(128 * n); exp = ctypes.create_string_buffer((n + 7) // 8)
oracle.sbvo_ed25519_gen_batch(0x0AD5, n, 30000, 4, tup, exp, os.cpu_count() or 1)
allt = blob * 3 + tup.raw
want = want * 3 + sbv.bitmap_to_list(exp.raw, n)
sbv.init(0)
sbv.set_grouping(True, 1, 8, 64)
for rnd in range(2):
    got = sbv.bitmap_to_list(sbv.ed25519_verify_batch(allt, len(want)), len(want))
    bad = [i for i in range(len(want)) if got[i] != want[i]]
    assert not bad, (rnd, bad[:8])
groups, grouped, ungrouped, rejected = sbv.last_group_stats()
assert ungrouped > 15000, (groups, grouped, ungrouped, rejected)
print("quad ok", groups, grouped, ungrouped, rejected)
"""
    env = dict(os.environ, SBV_ED_UNGROUPED_QUAD="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "quad ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_device_message_front_end_equals_host_tuple_builder(gpu, oracle):
    """sbv_ed25519_verify_msgs (SHA-512 + mod L on the device, sha512_dev.h) == sbv_ed25519_make_tuples + verify_batch on the
    golden vectors and on signed messages of every length around the SHA-512 block boundaries, some tampered."""
    import random
    vs = json.load(open(os.path.join(GOLDEN, "ed25519_vectors.json")))["vectors"]
    sigs = [bytes.fromhex(v["sig"]) for v in vs]
    pks = [bytes.fromhex(v["pk"]) for v in vs]
    msgs = [bytes.fromhex(v["msg"]) for v in vs]
    keep = [i for i, s in enumerate(sigs) if len(s) == 64]           # wrong-length signatures are the caller's reject
    sigs, pks, msgs = [sigs[i] for i in keep], [pks[i] for i in keep], [msgs[i] for i in keep]
    want = [vs[i]["accept"] for i in keep]
    rng = random.Random(512)
    oracle.sbvo_ed25519_public_key.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    oracle.sbvo_ed25519_sign.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    for j, mlen in enumerate([0, 1, 46, 47, 48, 63, 64, 65, 111, 112, 113, 127, 128, 129, 175, 176, 177, 500, 3000]):
        seed = bytes([j]) * 32
        pk, sig = ctypes.create_string_buffer(32), ctypes.create_string_buffer(64)
        msg = bytes(rng.getrandbits(8) for _ in range(mlen))
        oracle.sbvo_ed25519_public_key(seed, pk)
        oracle.sbvo_ed25519_sign(seed, msg, mlen, sig)
        sigs.append(sig.raw); pks.append(pk.raw); msgs.append(msg); want.append(True)
        if mlen:
            bad = bytearray(msg); bad[mlen // 2] ^= 1
            sigs.append(sig.raw); pks.append(pk.raw); msgs.append(bytes(bad)); want.append(False)
    n = len(sigs)
    got = sbv.bitmap_to_list(gpu.ed25519_verify_msgs(sigs, pks, msgs), n)
    via_tuples = sbv.bitmap_to_list(gpu.ed25519_verify_batch(gpu.ed25519_make_tuples(sigs, pks, msgs), n), n)
    assert got == via_tuples == want


def test_key_table_cache_of_this_scheme_on_gpu(gpu, oracle, openssl_check):
    """sbv_key_cache(SBV_SCHEME_ED25519) (round 4): the combs of -A a grouped batch builds stay in the scheme's own pool.  Cold,
    warm (same 300 keys, other signatures: every group hits), a warm batch of 4096 tuples (grouped by the cache alone), a
    capacity below the key set, the golden vectors' odd keys (non-canonical, small order, undecompressable) cached and reused,
    cache off.  Every bitmap equals the oracle's verifier and OpenSSL."""
    E = sbv.SCHEME_ED25519
    n = 1 << 17

    def run(seed, tup, exp, m):
        want = exp.raw[:(m + 7) // 8]
        assert _oracle_verdicts(oracle, tup.raw, m) == want
        assert _openssl_verdicts(openssl_check, seed, tup.raw, m) == want
        got = gpu.ed25519_verify_batch(tup.raw[:128 * m], m)
        assert got == want, [i for i in range(len(want)) if got[i] != want[i]][:8]
        return gpu.key_cache_stats(E)

    a, ea = _gen(oracle, 0xEDC1, n, 300, 7)
    b, eb = _gen(oracle, 0xEDC1, n + 384, 300, 5)
    small, es = _gen(oracle, 0xEDC1, 4096, 300, 7)
    try:
        gpu.set_grouping(True, 64, 64, 2048)
        gpu.key_cache(False, 0, E)
        gpu.key_cache(True, 1024, E)
        entries, hits, misses, cap = run(0xEDC1, a, ea, n)
        assert hits == 0 and misses >= 300 and entries == misses and cap == 1024, (entries, hits, misses, cap)
        first = entries
        entries, hits, misses, cap = run(0xEDC1, b, eb, n + 384)
        assert misses == 0 and hits == first and entries == first, (entries, hits, misses)
        entries, hits, misses, cap = run(0xEDC1, small, es, 4096)
        groups, grouped, generic, rejected = gpu.last_group_stats()
        assert misses == 0 and hits >= 290 and grouped > 3500, (hits, misses, groups, grouped, generic, rejected)
        gpu.key_cache(False, 0, E)
        gpu.key_cache(True, 256, E)
        entries, hits, misses, cap = run(0xEDC1, a, ea, n)
        assert entries == 256 and cap == 256
        entries, hits, misses, cap = run(0xEDC1, a, ea, n)
        assert hits == 256 and misses == first - 256
        # golden vectors x 80 (every key of theirs passes the count threshold): cold, then warm
        gpu.key_cache(False, 0, E)
        gpu.key_cache(True, 1024, E)
        vs = json.load(open(os.path.join(GOLDEN, "ed25519_vectors.json")))["vectors"]
        blob = b"".join(ed.pack_tuple(bytes.fromhex(v["pk"]), bytes.fromhex(v["msg"]), bytes.fromhex(v["sig"])) for v in vs) * 80
        want = [v["accept"] for v in vs] * 80
        for rep in range(2):
            got = sbv.bitmap_to_list(gpu.ed25519_verify_batch(blob, len(want)), len(want))
            assert got == want, (rep, [i for i in range(len(want)) if got[i] != want[i]][:8])
            entries, hits, misses, cap = gpu.key_cache_stats(E)
            assert (hits == 0 and misses > 5) if rep == 0 else (misses == 0 and hits > 5), (rep, entries, hits, misses)
        gpu.key_cache(False, 0, E)
        assert run(0xEDC1, a, ea, n)[:3] == (0, 0, 0)
    finally:
        gpu.key_cache(False, 0, E)
        gpu.key_cache(True, 1024, E)
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)
