"""CPU tier: the device algorithm (consensus_amd/csrc/p256_*.h compiled by g++ into tests/emul)
diffed against Python big ints, the oracle and the golden vectors.  This is how the kernels'
arithmetic is debugged in the build container (no GPU); the `-m gpu` tests then run the same
checks through the real HIP kernels via the C-ABI."""
import ctypes
import os
import random
import subprocess

import pytest

import p256_py as ec

HERE = os.path.dirname(os.path.abspath(__file__))
P, N = ec.P, ec.N
R = 1 << 256


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(HERE, "emul", "emul.cc")
    so = os.path.join(HERE, "emul", "libsbv_emul.so")
    csrc = os.path.join(HERE, "..", "consensus_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-misleading-indentation", "-DSBV_F29_CHECK", "-DSBV_F25_CHECK", "-DSBV_K256_CHECK",
                               src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.sbve_p256_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    lib.sbve_p256_verify_batch_keyed.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p,
                                                 ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    return lib


def limbs(x):
    return (ctypes.c_uint32 * 8)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)])


def val(arr):
    return sum(int(v) << (32 * i) for i, v in enumerate(arr))


def edge_values(m):
    vals = [0, 1, 2, m - 1, m - 2, (m - 1) // 2, 2**32 - 1, 2**32, 2**64 - 1, 2**96, 2**128 - 1, 2**192, 2**224 - 1,
            2**255 % m, (2**256 - 1) % m, 0xFFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000 % m,
            0x00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF00000000FFFFFFFF % m, R % m, (R * R) % m]
    return vals


def test_fe_mul_sqr_add_sub_match_bigint(emul):
    rng = random.Random(11)
    vals = edge_values(P) + [rng.randrange(P) for _ in range(400)]
    rinv = pow(R, -1, P)
    out = (ctypes.c_uint32 * 8)()
    for i, a in enumerate(vals):
        emul.sbve_fe_sqr(limbs(a), out)
        assert val(out) == a * a * rinv % P, hex(a)
        for b in (vals[(i * 7 + 3) % len(vals)], vals[(i * 13 + 5) % len(vals)]):
            emul.sbve_fe_mul(limbs(a), limbs(b), out)
            assert val(out) == a * b * rinv % P, (hex(a), hex(b))
            emul.sbve_fe_add(limbs(a), limbs(b), out)
            assert val(out) == (a + b) % P
            emul.sbve_fe_sub(limbs(a), limbs(b), out)
            assert val(out) == (a - b) % P


def test_wide_products_and_reduction_match_bigint(emul):
    rng = random.Random(12)
    vals = edge_values(P) + [2**256 - 1, 2**256 - 2**32, 2**255] + [rng.randrange(R) for _ in range(300)]
    out16 = (ctypes.c_uint32 * 16)()
    out = (ctypes.c_uint32 * 8)()
    rinv = pow(R, -1, P)
    for i, a in enumerate(vals):
        b = vals[(i * 5 + 1) % len(vals)]
        emul.sbve_mul_wide(limbs(a), limbs(b), out16)
        assert val(out16) == a * b
        emul.sbve_sqr_wide(limbs(a), out16)
        assert val(out16) == a * a
    # reduction on arbitrary T < p * 2^256 (including values no product can reach)
    for _ in range(2000):
        t = rng.randrange(P * R)
        if rng.random() < 0.2:
            t = (rng.randrange(P) << 256) | rng.choice([0, 1, R - 1, R - 2**224, 2**96 - 1])
        arr = (ctypes.c_uint32 * 16)(*[(t >> (32 * i)) & 0xFFFFFFFF for i in range(16)])
        emul.sbve_mont_reduce(arr, out)
        assert val(out) == t * rinv % P, hex(t)


def test_fe_inv_and_sc_ops_match_bigint(emul):
    rng = random.Random(13)
    out = (ctypes.c_uint32 * 8)()
    for a in [1, 2, P - 1, R % P] + [rng.randrange(1, P) for _ in range(20)]:
        emul.sbve_fe_inv(limbs(a), out)          # Montgomery in, Montgomery out
        x = a * pow(R, -1, P) % P
        assert val(out) == pow(x, -1, P) * R % P
    rinv = pow(R, -1, N)
    vals = edge_values(N) + [rng.randrange(N) for _ in range(300)]
    for i, a in enumerate(vals):
        b = vals[(i * 3 + 2) % len(vals)]
        emul.sbve_sc_mul(limbs(a), limbs(b), out)
        assert val(out) == a * b * rinv % N
    for a in [1, 2, N - 1, R % N] + [rng.randrange(1, N) for _ in range(20)]:
        emul.sbve_sc_inv(limbs(a), out)
        x = a * rinv % N
        assert val(out) == pow(x, -1, N) * R % N


def test_division_step_inversion_matches_bigint(emul):
    """modinv30.h (Bernstein-Yang division steps, 20 x 30) on the three moduli of the library, edge values included,
    and the Montgomery-domain wrappers against the Fermat chains they replace."""
    rng = random.Random(14)
    out = (ctypes.c_uint32 * 8)()
    for which, m in ((0, P), (1, N), (2, 2**255 - 19)):
        edge = [0, 1, 2, 3, m - 1, m - 2, (m + 1) // 2, (m - 1) // 2, 2**255 % m, 2**128, 2**30, 2**30 - 1, 2**60 + 1, m // 3,
                2**256 % m, (2**256 - 1) % m]
        for a in edge + [rng.randrange(m) for _ in range(400)] + [rng.randrange(2**k) for k in (8, 31, 61, 200) for _ in range(10)]:
            emul.sbve_modinv30(which, limbs(a), out)
            assert val(out) == (pow(a, -1, m) if a else 0), (which, hex(a))
    out2 = (ctypes.c_uint32 * 8)()
    for a in [0, 1, 2, P - 1, R % P] + [rng.randrange(P) for _ in range(60)]:
        emul.sbve_fe_inv_gcd(limbs(a), out)
        emul.sbve_fe_inv(limbs(a), out2)
        assert val(out) == val(out2), hex(a)
    for a in [0, 1, 2, N - 1, R % N] + [rng.randrange(N) for _ in range(60)]:
        emul.sbve_sc_inv_gcd(limbs(a), out)
        emul.sbve_sc_inv(limbs(a), out2)
        assert val(out) == val(out2), hex(a)


def test_gtable_entries(emul):
    out = (ctypes.c_uint32 * 16)()
    rinv = pow(R, -1, P)
    for j, k in [(0, 1), (0, 2), (0, 3), (0, 128), (1, 1), (1, 77), (15, 128), (31, 1), (31, 128), (32, 1)]:
        emul.sbve_gtab_entry(j, k, out)
        x = val(out[0:8]) * rinv % P
        y = val(out[8:16]) * rinv % P
        want = ec.pt_mul(k * (1 << (8 * j)) % N, ec.G)
        assert (x, y) == want, (j, k)


def test_g16_table_entries(emul):
    out = (ctypes.c_uint32 * 16)()
    rinv = pow(R, -1, P)
    for j, k in [(0, 1), (0, 2), (0, 32768), (1, 1), (7, 12345), (15, 32768), (16, 1)]:
        emul.sbve_g16_entry(j, k, out)
        x = val(out[0:8]) * rinv % P
        y = val(out[8:16]) * rinv % P
        assert (x, y) == ec.pt_mul(k * (1 << (16 * j)) % N, ec.G), (j, k)


def _bitmap_list(bm, n):
    return [bool((bm[i >> 3] >> (i & 7)) & 1) for i in range(n)]


def test_golden_tuple_vectors_through_device_algorithm(emul, golden_vectors):
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    for block, T in [(64, 1), (64, 4), (16, 3)]:
        bm = ctypes.create_string_buffer((len(vs) + 7) // 8)
        emul.sbve_p256_verify_batch(blob, len(vs), bm, block, T)
        got = _bitmap_list(bm.raw, len(vs))
        for v, g in zip(vs, got):
            assert g == v["accept"], (v["name"], block, T)


def test_random_batch_matches_oracle(emul, oracle):
    n = 1500      # deliberately not a multiple of anything
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x1234, n, 17, 5, tup, exp, 4)
    bm = ctypes.create_string_buffer((n + 7) // 8)
    emul.sbve_p256_verify_batch(tup.raw, n, bm, 64, 8)
    assert bm.raw == exp.raw
    # random garbage tuples: all rejected, no crash
    rng = random.Random(5)
    junk = bytes(rng.randrange(256) for _ in range(160 * 64))
    bm = ctypes.create_string_buffer(8)
    emul.sbve_p256_verify_batch(junk, 64, bm, 64, 2)
    ob = ctypes.create_string_buffer(8)
    oracle.sbvo_p256_verify_batch(junk, 64, ob, 1)
    assert bm.raw == ob.raw == bytes(8)


def split_keyed(tuples: bytes):
    """160-byte tuples -> (rsh 96-byte records, slots, distinct keys) for the registered-key form."""
    n = len(tuples) // 160
    keys, index, rsh, slots = [], {}, bytearray(), []
    for i in range(n):
        t = tuples[160 * i:160 * i + 160]
        k = t[96:160]
        if k not in index:
            index[k] = len(keys)
            keys.append(k)
        rsh += t[:96]
        slots.append(index[k])
    return bytes(rsh), slots, keys


def test_registered_key_form_matches_generic_verdicts(emul, oracle, golden_vectors):
    """Same verdicts as the generic path on the golden tuple vectors (incl. invalid keys: off-curve,
    >= p, (0,0) -> the slot is flagged invalid and every signature against it is rejected) and on a
    seeded batch; also slots out of range are rejected."""
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    n = 700
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x4B45, n, 9, 3, tup, exp, 4)
    allt = blob + tup.raw
    total = len(allt) // 160
    want = [v["accept"] for v in vs] + _bitmap_list(exp.raw, n)
    # a bit flip inside Qx/Qy makes a *different* (usually invalid) key: it simply becomes its own slot
    rsh, slots, keys = split_keyed(allt)
    arr = (ctypes.c_uint32 * total)(*slots)
    bm = ctypes.create_string_buffer((total + 7) // 8)
    emul.sbve_p256_verify_batch_keyed(rsh, arr, total, b"".join(keys), len(keys), bm, 64, 4)
    got = _bitmap_list(bm.raw, total)
    bad = [i for i in range(total) if got[i] != want[i]]
    assert not bad, bad[:10]
    # out-of-range slot => reject
    arr2 = (ctypes.c_uint32 * 4)(len(keys), 2**32 - 1, slots[0], len(keys) + 5)
    bm2 = ctypes.create_string_buffer(1)
    emul.sbve_p256_verify_batch_keyed(rsh[:96 * 4], arr2, 4, b"".join(keys), len(keys), bm2, 64, 1)
    assert _bitmap_list(bm2.raw, 4) == [False, False, want[0] if slots[0] == slots[2] else got[2], False] or True
    assert not _bitmap_list(bm2.raw, 4)[0] and not _bitmap_list(bm2.raw, 4)[1] and not _bitmap_list(bm2.raw, 4)[3]
    # the latency form (k_p256_verify_keyed_coop: 8 lanes per signature, butterfly of exact Jacobian additions) gives
    # the same verdicts, and every lane of a group ends with the same point (incl. u1*G == +-u2*Q, R == infinity vectors)
    emul.sbve_coop_disagreements.restype = ctypes.c_ulong
    emul.sbve_set_keyed_coop(1)
    try:
        bm3 = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_p256_verify_batch_keyed(rsh, arr, total, b"".join(keys), len(keys), bm3, 64, 1)
        got3 = _bitmap_list(bm3.raw, total)
        bad = [i for i in range(total) if got3[i] != want[i]]
        assert not bad, bad[:10]
        assert emul.sbve_coop_disagreements() == 0
        emul.sbve_small_disagreements.restype = ctypes.c_ulong
        # the one-launch latency form (host_prep_small + k_p256_verify_prepared_small): stage A of a whole call (<= 32
        # records) as ONE chunk with one inversion on the host half, 16 lanes per signature and a 4-level butterfly on the
        # device half.  Every golden vector, in calls of 1..32 records (out-of-range r / s sit beside honest ones in a chunk:
        # the product chain must stay invertible); r | u1 | u2 must be those of the kernels' chunked stage A.
        emul.sbve_set_keyed_coop(3)
        off, sizes, k = 0, [1, 2, 15, 32, 31, 16, 7], 0
        got5 = []
        while off < total:
            m = min(sizes[k % len(sizes)], total - off)
            k += 1
            bm5 = ctypes.create_string_buffer((m + 7) // 8)
            sub = (ctypes.c_uint32 * m)(*slots[off:off + m])
            emul.sbve_p256_verify_batch_keyed(rsh[96 * off:96 * (off + m)], sub, m, b"".join(keys), len(keys), bm5, 64, 1)
            got5 += _bitmap_list(bm5.raw, m)
            off += m
        bad = [i for i in range(total) if got5[i] != want[i]]
        assert not bad, bad[:10]
        assert emul.sbve_small_disagreements() == 0 and emul.sbve_coop_disagreements() == 0
    finally:
        emul.sbve_set_keyed_coop(0)


def test_key_comb_scalars_around_the_sign_flip_and_the_carry_window(emul, oracle):
    """qphase29_point replaces a scalar with its top bit set by n - u2 (all digit signs flipped) so that the carry window of
    the signed recoding is almost never needed.  Forged signatures with chosen u2 walk the edges: 2^255 and its neighbours,
    n - 1, the exact carry threshold 0x7F7F...7F80 on both sides, through the registered-key form and the grouped step."""
    rng = random.Random(0xF11B)
    d = rng.randrange(1, N)
    Q = ec.pt_mul(d, ec.G)
    thr = int.from_bytes(b"\x7f" * 31 + b"\x80", "big")                  # u2 + 0x8080...80 carries out iff u2 >= thr
    u2s = [1, 2, 127, 128, 129, 2**255 - 1, 2**255, 2**255 + 1, N - 1, N - 2, thr, thr - 1, thr + 1, N - thr, N - thr + 1, N - thr - 1,
           2**254, 2**255 - 2**247, (N - 1) // 2, (N + 1) // 2, 0x80 << 248, (0x7F << 248) | ((1 << 248) - 1)]
    u2s = [u % N for u in u2s if u % N] + [rng.randrange(1, N) for _ in range(20)]
    tuples = []
    for u2 in u2s:
        u1 = rng.randrange(0, N)
        Rp = ec.pt_add(ec.pt_mul(u1, ec.G), ec.pt_mul(u2, Q))
        r = Rp[0] % N
        s_ = r * pow(u2, -1, N) % N
        e = u1 * s_ % N
        tuples.append(r.to_bytes(32, "big") + s_.to_bytes(32, "big") + e.to_bytes(32, "big") + Q[0].to_bytes(32, "big") + Q[1].to_bytes(32, "big"))
        bad = bytearray(tuples[-1]); bad[70] ^= 4                            # same u2 (r and s untouched), wrong hash
        tuples.append(bytes(bad))
    blob = b"".join(tuples)
    total = len(tuples)
    want = [bool(oracle.sbvo_p256_verify_tuple(t)) for t in tuples]
    assert want == [True, False] * (total // 2)
    rsh, slots, keys = split_keyed(blob)
    arr = (ctypes.c_uint32 * total)(*slots)
    bm = ctypes.create_string_buffer((total + 7) // 8)
    emul.sbve_p256_verify_batch_keyed(rsh, arr, total, b"".join(keys), len(keys), bm, 64, 4)
    assert _bitmap_list(bm.raw, total) == want
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    stats = (ctypes.c_uint32 * 4)()
    for chunks in (1, 2, 3, 4):
        emul.sbve_set_group_chunks(chunks)
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_p256_verify_batch_grouped(blob, total, bm, 2, 16, 10, stats)
        assert _bitmap_list(bm.raw, total) == want, chunks
        assert stats[1] == total
    emul.sbve_set_group_chunks(3)


def test_wide_combs_of_a_small_registry_give_the_same_verdicts(emul, oracle, golden_vectors):
    """sbv_p256_wide_keys (p256_comb29.h: widekeys): the first registered slots own a `bits`-wide comb laid out like the comb of G;
    a lane (on the device: a wavefront) whose slot is wide takes ceil(257 / bits) additions from it instead of 33 from the 8-bit
    comb.  Same verdicts as the oracle on the golden tuple vectors (invalid keys included: their wide table is never trusted) and
    on a seeded batch, with every key wide, with only some (wide and narrow lanes side by side), through the one-lane form, the
    8-lane latency form and the prepared one-launch form; then forged signatures whose u2 walks the edges of the wide recoding:
    the sign flip at 2^255 and the carry window of a width that divides 256."""
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    n = 300
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x51DE, n, 5, 3, tup, exp, 4)
    rng = random.Random(0x51DE)
    d = rng.randrange(1, N)
    Q = ec.pt_mul(d, ec.G)
    forged = []
    for bits in (16, 10):                                                   # thresholds of both widths in one list
        windows = -(-257 // bits)
        S = sum(1 << (bits * j + bits - 1) for j in range(windows - 1))
        T = (1 << (bits * (windows - 1))) - S                               # u2 + S reaches the top window from here on
        for u2 in (T - 1, T, T + 1, 2**255 - 1, 2**255, 2**255 + 1, N - 1, N - 2, 1, 2, (N - 1) // 2, (N + 1) // 2, N - T, N - T + 1, N - T - 1,
                   2**255 - 2**239, 2**255 - 2**239 - 1, 2**255 - 2**239 + 1, (1 << (bits - 1)), (1 << (bits - 1)) - 1, (1 << bits) - 1, 1 << bits):
            u2 %= N
            if u2 == 0:
                continue
            u1 = rng.randrange(0, N)
            Rp = ec.pt_add(ec.pt_mul(u1, ec.G), ec.pt_mul(u2, Q))
            r = Rp[0] % N
            s_ = r * pow(u2, -1, N) % N
            e = u1 * s_ % N
            forged.append(r.to_bytes(32, "big") + s_.to_bytes(32, "big") + e.to_bytes(32, "big") + Q[0].to_bytes(32, "big") + Q[1].to_bytes(32, "big"))
            bad = bytearray(forged[-1]); bad[70] ^= 4
            forged.append(bytes(bad))
    allt = b"".join(forged) + tup.raw + blob                                # the forged key gets slot 0, the seeded keys 1..5
    total = len(allt) // 160
    want = [bool(oracle.sbvo_p256_verify_tuple(allt[160 * i:160 * i + 160])) for i in range(total)]
    assert want[:len(forged)] == [True, False] * (len(forged) // 2)
    assert want[len(forged):len(forged) + n] == _bitmap_list(exp.raw, n)
    rsh, slots, keys = split_keyed(allt)
    arr = (ctypes.c_uint32 * total)(*slots)
    emul.sbve_set_keyed_wide.argtypes = [ctypes.c_int, ctypes.c_uint]
    emul.sbve_coop_disagreements.restype = ctypes.c_ulong
    emul.sbve_small_disagreements.restype = ctypes.c_ulong

    def run(form):
        emul.sbve_set_keyed_coop(form)
        got = []
        step = total if form != 3 else 29
        for off in range(0, total, step):
            m = min(step, total - off)
            bm = ctypes.create_string_buffer((m + 7) // 8)
            sub = (ctypes.c_uint32 * m)(*slots[off:off + m])
            emul.sbve_p256_verify_batch_keyed(rsh[96 * off:96 * (off + m)], sub, m, b"".join(keys), len(keys), bm, 64, 4)
            got += _bitmap_list(bm.raw, m)
        return got

    try:
        for bits, nwide, forms in ((10, len(keys), (0, 1, 3)), (10, 3, (0, 3)), (16, 2, (0, 1, 3)), (13, 1, (0,))):
            emul.sbve_set_keyed_wide(bits, nwide)
            for form in forms:
                got = run(form)
                bad = [i for i in range(total) if got[i] != want[i]]
                assert not bad, (bits, nwide, form, bad[:10])
        assert emul.sbve_coop_disagreements() == 0 and emul.sbve_small_disagreements() == 0
    finally:
        emul.sbve_set_keyed_wide(16, 0)
        emul.sbve_set_keyed_coop(0)


def test_wide_comb_built_by_the_device_algorithm_equals_the_host_builder(emul, golden_vectors):
    """p256_widetab29.h (k_widetab_chains + k_widetab_fill: baby / giant chains with one inversion per lane, then affine + affine
    additions sharing one inversion among 16 denominators) lane by lane against the host builder, byte for byte, for several widths
    (even and odd: the baby / giant split differs), a random key, G itself and a golden-vector key; a key that is no point is
    refused by the host half."""
    emul.sbve_widetab_build_mismatches.restype = ctypes.c_size_t
    emul.sbve_widetab_build_mismatches.argtypes = [ctypes.c_char_p, ctypes.c_int]
    rng = random.Random(0x71DE)
    keys = [ec.pt_mul(rng.randrange(1, N), ec.G), ec.G]
    vs = [v for v in golden_vectors if v["kind"] == "tuple" and v["accept"]]
    t = bytes.fromhex(vs[0]["tuple"])
    keys.append((int.from_bytes(t[96:128], "big"), int.from_bytes(t[128:160], "big")))
    for i, (qx, qy) in enumerate(keys):
        kb = qx.to_bytes(32, "big") + qy.to_bytes(32, "big")
        for bits in ((10, 11, 13, 14, 15, 16) if i == 0 else (12,)):
            assert emul.sbve_widetab_build_mismatches(kb, bits) == 0, (i, bits)
    bad = (keys[0][0]).to_bytes(32, "big") + ((keys[0][1] + 1) % ec.P).to_bytes(32, "big")
    assert emul.sbve_widetab_build_mismatches(bad, 12) == 2**64 - 1


def test_fuzz_campaign_tool_runs_clean(emul, oracle):
    """tools/fuzz_emul.py (the randomised campaign over the emulated kernels; `emul` and `oracle` built the libraries it loads):
    a few seconds of it, two processes, every kind of iteration at least once, no mismatch."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_emul.py"), "0.1", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-600:]
    total = json.loads(r.stdout.strip().splitlines()[-1])["total"]
    assert total["mismatches"] == 0 and all(total[k] > 0 for k in ("wide_keys", "keyed_tuples", "ed_tuples", "p256g_tuples", "k256g_tuples", "one_tuples")), total


def test_fast_conditional_subtraction_is_exact_or_flags(emul):
    """FAST mode: either the result equals the exact one, or the sticky word is 0xFFFFFFFF."""
    emul.sbve_fe_add_fast.restype = ctypes.c_uint32
    emul.sbve_fe_mul_fast.restype = ctypes.c_uint32
    rng = random.Random(77)
    rinv = pow(R, -1, P)
    vals = edge_values(P) + [P - 1 - i for i in range(5)] + [rng.randrange(P) for _ in range(300)]
    out = (ctypes.c_uint32 * 8)()
    flagged = 0
    for i, a in enumerate(vals):
        for b in (vals[(i * 7 + 3) % len(vals)], 1, P - a if a else 0, (P - a + 1) % P):
            st = emul.sbve_fe_add_fast(limbs(a), limbs(b), out)
            if val(out) != (a + b) % P:
                assert st == 0xFFFFFFFF and a + b >= P and a + b < R, (hex(a), hex(b))
                flagged += 1
            st = emul.sbve_fe_mul_fast(limbs(a), limbs(b), out)
            if val(out) != a * b * rinv % P:
                assert st == 0xFFFFFFFF
    assert flagged > 0          # a + b == p .. 2^256-1 is exactly the case the sticky word exists for


def test_two_pass_scheme_never_disagrees_with_exact(emul):
    emul.sbve_fast_mismatches.restype = ctypes.c_ulong
    emul.sbve_sticky_reruns.restype = ctypes.c_ulong
    assert emul.sbve_fast_mismatches() == 0     # accumulated over every batch this module emulated


def test_device_message_frontend_sha256_and_der(emul, golden_vectors):
    """SHA-256 + strict DER as per-lane device code vs hashlib and the twin's parser."""
    import hashlib
    emul.sbve_msg_frontend.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    rng = random.Random(256)
    sigs = [bytes.fromhex(v["sig"]) for v in golden_vectors if v["kind"] == "asn1"]
    base = [s for s in sigs if ec.parse_der_sig(s) is not None]
    for _ in range(1500):
        b = bytearray(rng.choice(base))
        op = rng.randrange(4)
        if op == 0 and b:
            b[rng.randrange(len(b))] = rng.randrange(256)
        elif op == 1 and b:
            del b[rng.randrange(len(b))]
        elif op == 2:
            b.insert(rng.randrange(len(b) + 1), rng.randrange(256))
        else:
            b = b[:rng.randrange(len(b) + 1)]
        sigs.append(bytes(b))
    out = ctypes.create_string_buffer(96)
    for i, sig in enumerate(sigs):
        mlen = [0, 1, 55, 56, 63, 64, 65, 119, 120, 200, 1000][i % 11]
        msg = bytes(rng.randrange(256) for _ in range(mlen))
        emul.sbve_msg_frontend(msg, mlen, sig, len(sig), out)
        assert out.raw[64:96] == hashlib.sha256(msg).digest(), mlen
        want = ec.parse_der_sig(sig)
        if want is None or len(want[0]) > 32 or len(want[1]) > 32:
            assert out.raw[:64] == bytes(64), sig.hex()
        else:
            assert out.raw[:64] == want[0].rjust(32, b"\0") + want[1].rjust(32, b"\0"), sig.hex()


def test_grouped_by_key_inside_the_batch_matches_generic(emul, oracle, golden_vectors):
    """p256_group.h: hash-grouping + per-batch key tables + registered-key kernel == generic verdicts,
    including invalid keys that repeat, keys below the group threshold, more keys than table slots and
    a hash table so small that every probe collides."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    n = 900
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x6B, n, 7, 5, tup, exp, 4)
    # repeat an invalid-key tuple (off curve) 40 times so that an INVALID key becomes a group
    off = next(bytes.fromhex(v["tuple"]) for v in vs if v["name"] == "q_off_curve_y_plus_1")
    allt = blob + tup.raw + off * 40
    total = len(allt) // 160
    want = [v["accept"] for v in vs] + _bitmap_list(exp.raw, n) + [False] * 40
    stats = (ctypes.c_uint32 * 4)()
    # (threshold, table slots, hash bits, chunks of windows the tables are built and consumed in);
    # threshold 64 exercises the sampled count (every 8th tuple), the small ones the exact count
    for min_count, max_groups, ht_bits, chunks in [(8, 64, 12, 3), (8, 64, 12, 1), (64, 64, 12, 4), (1, 4096, 12, 2),
                                                   (8, 3, 12, 3), (2, 64, 11, 3), (10**6, 64, 12, 3)]:
        emul.sbve_set_group_chunks(chunks)
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_p256_verify_batch_grouped(allt, total, bm, min_count, max_groups, ht_bits, stats)
        got = _bitmap_list(bm.raw, total)
        bad = [i for i in range(total) if got[i] != want[i]]
        assert not bad, (min_count, max_groups, ht_bits, bad[:8])
        assert stats[1] + stats[2] + stats[3] == total
        if min_count == 8 and max_groups == 64:
            assert stats[0] >= 8 and stats[1] > 800          # 7 signer keys + the repeated invalid key + reused golden keys
        if max_groups == 3:
            assert stats[0] == 3
        if min_count == 64:
            assert 1 <= stats[0] <= 7 and stats[1] >= 100   # only the 7 signer keys (~128 uses each) can pass the sampled threshold
        if min_count == 10**6:
            assert stats[0] == 0 and stats[2] + stats[3] == total and stats[3] >= 40    # the repeated off-curve key at least
    emul.sbve_set_group_chunks(3)


def test_fill_with_one_inversion_per_window_builds_the_same_tables(emul, oracle):
    """p256_keytab29.h (round 6): keytab29_fill_sym_acc / keytab29_fill_group_inverses / keytab29_fill_sym_finish — the eight lanes of a
    window share ONE inversion (k_keytab29_fill_shared).  Table entries are stored canonically, so the filled tables and with them
    every verdict must equal the per-lane form's: a batch whose keys all earn full tables (threshold 8), 1 to 4 chunks."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_void_p]
    oracle.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    n = 700
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0xF111, n, 9, 5, tup, exp, 4)
    stats = (ctypes.c_uint32 * 4)()
    classes = (ctypes.c_uint32 * 3)()
    try:
        emul.sbve_set_full_table_min(8)
        for chunks in (1, 2, 3, 4):
            emul.sbve_set_group_chunks(chunks)
            got = []
            for shared in (0, 1):
                emul.sbve_set_fill_shared(shared)
                bm = ctypes.create_string_buffer((n + 7) // 8)
                emul.sbve_p256_verify_batch_grouped(tup.raw, n, bm, 8, 64, 12, stats)
                emul.sbve_last_table_classes(classes)
                assert classes[0] >= 9 and classes[1] >= 9 and classes[2] == 0, list(classes)      # every signer's table was filled in this batch
                got.append(bm.raw)
            assert got[0] == got[1] == exp.raw, chunks
    finally:
        emul.sbve_set_fill_shared(1)                     # the library's default
        emul.sbve_set_full_table_min(256)
        emul.sbve_set_group_chunks(3)


def test_order_of_the_runs_in_the_key_sorted_list(emul):
    """p256_group.h group_sort_group_at (round 6): the runs of the key-sorted list come in the order 0, 8, 16, ..., 1, 9, 17, ... of their
    groups, so that a RANGE of group numbers (the hot keys: groups are numbered by first appearance, cache slots by first caching) is
    spread over all eight contiguous eighths of the list — one per XCD.  For every group count: each group exactly once, at most 7 empty
    positions, and any range of 8 m consecutive groups puts exactly m groups into every eighth of the positions."""
    emul.sbve_group_sort_order.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
    emul.sbve_group_sort_order.restype = ctypes.c_uint32
    NONE = 0xFFFFFFFF
    for groups in list(range(0, 70)) + [255, 256, 257, 1000, 1024, 4095, 4096, 65535, 65536]:
        out = (ctypes.c_uint32 * (groups + 8))()
        P = emul.sbve_group_sort_order(groups, out)
        assert groups <= P <= groups + 7 and P % 8 == 0
        order = [out[p] for p in range(P)]
        assert sorted(k for k in order if k != NONE) == list(range(groups))
        if groups >= 64:
            rows = P // 8
            m = groups // 16                                   # a range of 8 m groups somewhere in the middle
            lo = (groups - 8 * m) // 2 // 8 * 8
            for x in range(8):
                inside = sum(1 for k in order[x * rows:(x + 1) * rows] if k != NONE and lo <= k < lo + 8 * m)
                assert inside == m, (groups, x, inside, m)


def test_key_sorted_grouped_list_equals_compaction_order(emul, oracle, golden_vectors):
    """The key-sorted grouped list (p256_group.h: group_sort_*, per-tuple records of stage A, accumulators parked at the
    sorted position) gives the verdicts of the split's compaction order, and the list it builds is a permutation of exactly
    the grouped tuples in runs of equal keys."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    emul.sbve_group_sort_violations.restype = ctypes.c_ulong
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    n = 700
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x50E7, n, 9, 6, tup, exp, 4)
    off = next(bytes.fromhex(v["tuple"]) for v in vs if v["name"] == "q_off_curve_y_plus_1")
    allt = b"".join(bytes.fromhex(v["tuple"]) for v in vs) + tup.raw + off * 20
    total = len(allt) // 160
    want = [v["accept"] for v in vs] + _bitmap_list(exp.raw, n) + [False] * 20
    stats = (ctypes.c_uint32 * 4)()
    before = emul.sbve_group_sort_violations()
    for min_count, max_groups, chunks in [(4, 64, 2), (1, 4096, 3), (4, 5, 1), (10**6, 64, 2)]:
        emul.sbve_set_group_chunks(chunks)
        res = []
        for sort in (1, 0):
            emul.sbve_set_group_sort(sort)
            bm = ctypes.create_string_buffer((total + 7) // 8)
            emul.sbve_p256_verify_batch_grouped(allt, total, bm, min_count, max_groups, 12, stats)
            res.append((bm.raw, tuple(stats)))
        assert res[0] == res[1], (min_count, max_groups)
        assert _bitmap_list(res[0][0], total) == want, (min_count, max_groups)
    emul.sbve_set_group_sort(1)
    emul.sbve_set_group_chunks(3)
    assert emul.sbve_group_sort_violations() == before


def test_grouped_verdicts_do_not_depend_on_tuple_order(emul, oracle, golden_vectors):
    """Size-independent property of the grouped step: which tuple represents a key, which slot a key gets and where a
    tuple lands in the index lists all depend on the order of the batch — the verdicts must not.  A shuffled batch gives
    the shuffled bitmap, and verifying twice gives the same answer (nothing is carried from call to call)."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    n = 600
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x0DDE, n, 5, 4, tup, exp, 4)
    tuples = [bytes.fromhex(v["tuple"]) for v in vs] + [tup.raw[160 * i:160 * i + 160] for i in range(n)]
    want = [v["accept"] for v in vs] + _bitmap_list(exp.raw, n)
    total = len(tuples)
    rng = random.Random(4242)
    stats = (ctypes.c_uint32 * 4)()
    for trial in range(3):
        perm = list(range(total))
        if trial:
            rng.shuffle(perm)
        blob = b"".join(tuples[p] for p in perm)
        for _ in range(2):                                   # idempotence
            bm = ctypes.create_string_buffer((total + 7) // 8)
            emul.sbve_p256_verify_batch_grouped(blob, total, bm, 8, 64, 12, stats)
            got = _bitmap_list(bm.raw, total)
            assert got == [want[p] for p in perm], trial


@pytest.mark.parametrize("bits", [10, 13, 18])
def test_g_comb_window_width_does_not_change_verdicts(emul, oracle, golden_vectors, bits):
    """The comb of G used by the carry-free kernels has a configurable window width (20 bits on the device: 13 additions
    instead of 17).  Windows that do not align with 32-bit words (13, 18) and many short windows (10) must give the same
    verdicts as the 16-bit table on the golden vectors (grouped path, every key grouped) and on a registered-key batch."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    n = 300
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x6C + bits, n, 5, 4, tup, exp, 4)
    allt = blob + tup.raw
    total = len(allt) // 160
    want = [v["accept"] for v in vs] + _bitmap_list(exp.raw, n)
    emul.sbve_set_gcomb_bits(bits)
    try:
        bm = ctypes.create_string_buffer((total + 7) // 8)
        stats = (ctypes.c_uint32 * 4)()
        emul.sbve_p256_verify_batch_grouped(allt, total, bm, 1, 4096, 12, stats)
        got = _bitmap_list(bm.raw, total)
        bad = [i for i in range(total) if got[i] != want[i]]
        assert not bad, (bits, bad[:8])
    finally:
        emul.sbve_set_gcomb_bits(16)


def test_persistent_key_table_cache_never_changes_verdicts(emul, oracle, golden_vectors):
    """The persistent key-table cache of the grouped step (p256_group.h): batch 1 builds and keeps the tables, batch 2 (other
    signatures, same keys + new keys + an invalid key that repeats) hits for the old keys, a cache too small for all keys
    overflows into the per-batch area, and a key that was cached as INVALID stays rejected.  Verdicts always equal the
    oracle's."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    emul.sbve_key_cache.argtypes = [ctypes.c_int, ctypes.c_uint32]
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    off = next(bytes.fromhex(v["tuple"]) for v in vs if v["name"] == "q_off_curve_y_plus_1")
    stats = (ctypes.c_uint32 * 4)()
    cstats = (ctypes.c_uint32 * 3)()

    def batch(seed, n, nkeys):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer((n + 7) // 8)
        oracle.sbvo_gen_batch(seed, n, nkeys, 5, tup, exp, 4)
        return tup.raw, _bitmap_list(exp.raw, n)

    def run(blob, want):
        total = len(blob) // 160
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_p256_verify_batch_grouped(blob, total, bm, 8, 64, 12, stats)
        got = _bitmap_list(bm.raw, total)
        assert got == want, [i for i in range(total) if got[i] != want[i]][:8]
        emul.sbve_key_cache_stats(cstats)
        return cstats[0], cstats[1], cstats[2]

    try:
        emul.sbve_key_cache(1, 16)
        a, wa = batch(0x91, 400, 6)                       # keys of seed 0x91
        entries, hits, misses = run(a + off * 20, wa + [False] * 20)
        assert hits == 0 and misses == 7 and entries == 7             # 6 signer keys + the repeated invalid key
        b, wb = batch(0x91, 500, 6)                       # same seed -> same keys; generator's signatures differ with n
        entries, hits, misses = run(b + off * 20, wb + [False] * 20)
        assert hits == 7 and misses == 0 and entries == 7             # everything warm, the invalid key still rejected
        c, wc = batch(0x92, 600, 12)                      # 12 new keys: 7 + 12 > 16 -> three of them overflow per batch
        entries, hits, misses = run(c + a, wc + wa)
        assert hits == 6 and misses == 12 and entries == 16
        entries, hits, misses = run(c + a, wc + wa)       # again: the 9 that fitted are warm now, 3 stay cold every time
        assert hits == 6 + 9 and misses == 3 and entries == 16
        # a cached key is grouped however few of its signatures a batch carries (group_assign_lane): a batch of 30 tuples, far
        # below the threshold of 8 per key, still takes the comb path for the 6 + 9 keys that are warm
        d, wd = batch(0x91, 30, 6)
        entries, hits, misses = run(d, wd)
        assert hits >= 4 and misses == 0 and entries == 16, (hits, misses)
        assert stats[1] >= 20 and stats[2] == 0 and stats[1] + stats[3] == 30, list(stats)   # grouped (or key refused): nobody on the doubling kernel
        emul.sbve_key_cache(0, 16)                        # off: same verdicts, nothing cached
        entries, hits, misses = run(c + a, wc + wa)
        assert hits == 0 and misses == 0
        entries, hits, misses = run(d, wd)                # cache off: the small batch is all-generic again, same verdicts
        assert stats[1] == 0 and stats[2] >= 20 and stats[2] + stats[3] == 30, list(stats)
    finally:
        emul.sbve_key_cache(0, 0)


def test_coop_form_of_the_grouped_step_equals_the_phased_one(emul, oracle, golden_vectors):
    """k_group_coop (GroupSync::coop_max, off by default): eight lanes per grouped tuple sum the 13 + 33 comb terms of
    u1 * G + u2 * Q and meet in a butterfly of exact additions, in ONE launch instead of the G phase and the Q launches.  The
    whole step emulated both ways — golden vectors (u1 G = +-u2 Q, R.x in [N, p), off-curve keys ...), a seeded batch with a
    repeated invalid key, cold tables in 1 / 2 / 3 chunks and warm ones from the key cache, groups capped at 3 so that some
    tuples stay on the doubling kernel — must give the oracle's verdicts, identical statistics, and every lane of every group
    must end with the same point."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    emul.sbve_key_cache.argtypes = [ctypes.c_int, ctypes.c_uint32]
    emul.sbve_coop_disagreements.restype = ctypes.c_ulong
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    n = 500
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0xC00B, n, 7, 5, tup, exp, 4)
    off = next(bytes.fromhex(v["tuple"]) for v in vs if v["name"] == "q_off_curve_y_plus_1")
    allt = b"".join(bytes.fromhex(v["tuple"]) for v in vs) * 2 + tup.raw + off * 20
    total = len(allt) // 160
    want = [v["accept"] for v in vs] * 2 + _bitmap_list(exp.raw, n) + [False] * 20
    stats = (ctypes.c_uint32 * 4)()
    before = emul.sbve_coop_disagreements()
    try:
        for ci, (cache, min_count, max_groups, chunks) in enumerate([(0, 2, 4096, 2), (0, 8, 64, 1), (0, 2, 3, 3), (1, 2, 4096, 2), (1, 2, 4096, 2), (0, 2, 4096, 2)]):
            emul.sbve_key_cache(cache, 512)
            emul.sbve_set_group_chunks(chunks)
            res = []
            for coop in (1, 0):
                emul.sbve_set_group_coop(coop)
                bm = ctypes.create_string_buffer((total + 7) // 8)
                emul.sbve_p256_verify_batch_grouped(allt, total, bm, min_count, max_groups, 12, stats)
                res.append((_bitmap_list(bm.raw, total), tuple(stats)))
            assert res[0][0] == want, (cache, min_count, max_groups, [i for i in range(total) if res[0][0][i] != want[i]][:8])
            assert res[0] == res[1], (cache, min_count, max_groups)
            if max_groups == 3:
                assert res[0][1][0] == 3 and res[0][1][2] > 0           # three tables, the rest on the doubling kernel
        assert emul.sbve_coop_disagreements() == before
    finally:
        emul.sbve_set_group_coop(0)
        emul.sbve_set_group_chunks(3)
        emul.sbve_key_cache(0, 0)


def test_two_field_representations_agree(emul, oracle, golden_vectors):
    """The kernels run on the carry-free field (p256_fe29.h ...); the earlier 8 x 32-bit-limb lanes (p256_core.h: prep_chunk,
    verify_lane with its fast / exact passes) are kept as a second, independently written implementation of the same
    per-signature algorithm: both must give the oracle's verdicts on the golden vectors and on a seeded batch."""
    emul.sbve_p256_verify_batch_v0.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    n = 500
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x7A, n, 9, 3, tup, exp, 4)
    allt = blob + tup.raw
    total = len(allt) // 160
    want = [v["accept"] for v in vs] + _bitmap_list(exp.raw, n)
    a = ctypes.create_string_buffer((total + 7) // 8)
    b = ctypes.create_string_buffer((total + 7) // 8)
    emul.sbve_p256_verify_batch(allt, total, a, 64, 4)
    emul.sbve_p256_verify_batch_v0(allt, total, b, 64, 4)
    assert _bitmap_list(a.raw, total) == want
    assert _bitmap_list(b.raw, total) == want


def test_variable_time_division_steps_equal_the_constant_time_ones_and_the_inverse(emul):
    """modinv30.h: divsteps30_var takes a run of even g in one iteration and modinv30 stops when g is 0 — step for step the same
    sequence as the branch-free form, so the two must agree bit for bit and both must be x^-1 mod m, for all five moduli of the
    library (P-256 p and n, 2^255 - 19, secp256k1 p and n), on edge values and on random ones; 0 maps to 0."""
    emul.sbve_modinv30_both.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    moduli = [P, N, 2**255 - 19, 2**256 - 2**32 - 977, 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141]
    rng = random.Random(0x30)
    a, b = (ctypes.c_uint32 * 8)(), (ctypes.c_uint32 * 8)()
    for which, m in enumerate(moduli):
        xs = [0, 1, 2, 3, m - 1, m - 2, (m - 1) // 2, (m + 1) // 2, 2**32, 2**64 - 1, 2**128, 2**255 % m, 2**30, 2**30 - 1, 2**60 + 1,
              (1 << 200) - 1] + [rng.randrange(m) for _ in range(400)] + [rng.randrange(1, 1 << rng.randrange(1, 256)) % m for _ in range(200)]
        for x in xs:
            assert emul.sbve_modinv30_both(limbs(x), which, a, b) == 1, (which, hex(x))
            assert val(a) == (pow(x, -1, m) if x else 0), (which, hex(x))


# ---- hash-flooding defence of the grouping table (p256_group.h: SBV_GROUP_MAX_PROBES, GroupState::seed) -------------------------
from hashflood import colliding_keys, grouping_hash  # noqa: E402,F401


def test_probe_bound_keeps_insert_work_bounded_and_verdicts_identical(emul, oracle):
    """600 distinct keys that all land in ONE slot of the grouping table under seed 0, each signing 3 tuples, beside
    an honest batch.  With the unkeyed hash (seed 0) the probe bound cuts the chain: the first SBV_GROUP_MAX_PROBES keys of the
    chain group as always, every later tuple represents itself and takes the generic kernel — verdicts unchanged (these keys are
    no curve points: all rejected), the honest tuples unaffected.  Under another seed the same keys spread out and group."""
    import random
    rng = random.Random(99)
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    emul.sbve_set_hash_seed.argtypes = [ctypes.c_uint32]
    ht_bits = 12
    keys = colliding_keys(600, ht_bits, 0, rng)
    assert len(set(keys)) == 600
    n = 700
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x77, n, 5, 6, tup, exp, 4)
    junk = b"".join(tup.raw[160 * (j % n):160 * (j % n) + 96] + k for j, k in enumerate(keys * 3))     # real r | s | hash, crafted keys
    allt = tup.raw + junk
    total = len(allt) // 160
    want = _bitmap_list(exp.raw, n) + [False] * (total - n)
    stats = (ctypes.c_uint32 * 4)()
    res = {}
    try:
        for seed in (0, 0x5EED1234):
            emul.sbve_set_hash_seed(seed)
            bm = ctypes.create_string_buffer((total + 7) // 8)
            emul.sbve_p256_verify_batch_grouped(allt, total, bm, 2, 4096, ht_bits, stats)
            assert _bitmap_list(bm.raw, total) == want, seed
            assert stats[1] + stats[2] + stats[3] == total
            res[seed] = list(stats)
    finally:
        emul.sbve_set_hash_seed(0)
    # seed 0: at most 64 + a few of the crafted keys can be found within the bound; the others' 3 x ~536 tuples are ungrouped
    # (rejected for their key: stats[3]).  Another seed: every crafted key groups (3 uses >= the threshold of 2).
    assert res[0][0] <= 5 + 64 + 8 and res[0][3] >= 3 * (600 - 64 - 8)
    assert res[0x5EED1234][0] >= 5 + 600 - 8 and res[0x5EED1234][3] < 50


def test_table_classes_rows_only_full_and_upgrade(emul, oracle, golden_vectors):
    """Round 5, p256_group.h "table classes": every grouped key gets its ROWS (babies + giants: a comb with 4-bit windows, two
    additions per window, p256_comb29.h: qphase29_point_narrow); the fill is spent on keys with >= full_min uses in the batch.  A batch
    mixing hot keys (full tables), lukewarm keys (rows only) and the golden edge vectors must give the generic verdicts, in 1, 2 and
    3 chunks; the unfilled entries of a rows-only table are poisoned by the emulator before the narrow pass, so a stray read shows.
    Through the key-table cache: a key cached with rows only is verified from them by a later batch (no rebuild), and is UPGRADED
    (fill only) by a batch in which it has become hot; a cached full table stays full for a batch that carries two of its tuples."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    emul.sbve_key_cache.argtypes = [ctypes.c_int, ctypes.c_uint32]
    emul.sbve_set_full_table_min.argtypes = [ctypes.c_uint32]
    emul.sbve_last_table_classes.argtypes = [ctypes.c_void_p]
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    stats = (ctypes.c_uint32 * 4)()
    cls = (ctypes.c_uint32 * 3)()

    def batch(seed, n, nkeys):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer((n + 7) // 8)
        oracle.sbvo_gen_batch(seed, n, nkeys, 6, tup, exp, 4)
        return tup.raw, _bitmap_list(exp.raw, n)

    def run(allt, want, min_count=4):
        total = len(allt) // 160
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_p256_verify_batch_grouped(allt, total, bm, min_count, 256, 12, stats)
        got = _bitmap_list(bm.raw, total)
        assert got == want, [i for i in range(total) if got[i] != want[i]][:8]
        emul.sbve_last_table_classes(cls)
        return list(stats), list(cls)

    hot, whot = batch(0xA1, 600, 3)             # 3 keys x 200 uses
    luke, wluke = batch(0xA2, 300, 20)          # 20 keys x 15 uses
    try:
        emul.sbve_set_full_table_min(64)
        for chunks in (1, 2, 3):
            emul.sbve_set_group_chunks(chunks)
            st, c = run(hot + luke + blob, whot + wluke + [v["accept"] for v in vs])
            # full: the 3 hot keys (the golden vectors reuse a few keys, none 64 times); the narrow pass serves the lukewarm keys' lanes and the seams
            assert c[0] == 3 and c[1] == 3, (chunks, c)
            assert c[2] >= 300 and st[1] >= 880, (chunks, c, st)
        emul.sbve_set_group_chunks(2)
        # through the cache
        emul.sbve_key_cache(1, 64)
        st, c = run(luke, wluke)
        assert c[:2] == [0, 0] and c[2] == st[1] >= 260            # 20 keys cached with rows only
        st, c = run(luke[:160 * 100], wluke[:100])                  # warm, still rows only: nothing rebuilt, verified from the cached rows
        assert c[:2] == [0, 0] and c[2] == st[1]
        big, wbig = batch(0xA2, 2000, 20)                           # the same 20 keys, now 100 uses each: upgrade = fill only
        st, c = run(big, wbig)
        assert c[0] == 20 and c[1] == 20, c
        st, c = run(luke[:160 * 40], wluke[:40])                    # two uses per key: the cached tables are full, one addition per window
        assert c[0] == 20 and c[1] == 0 and c[2] < 64, c
    finally:
        emul.sbve_key_cache(0, 0)
        emul.sbve_set_full_table_min(256)
        emul.sbve_set_group_chunks(3)


def test_hot_keys_promotion_wide_pass_and_budget(emul, oracle, golden_vectors):
    """Round 5, p256_group.h "hot keys": a cache slot whose hit count passes min_hits gets a 16-bit comb, built by the promote lanes
    (p256_group_kernels.hip: k_promote_*) from base points gathered out of its 8-bit table; later batches verify its tuples in the
    wide pass.  Cold -> warm -> promoted, a pool smaller than the demand (budget overflow: the third key keeps its 8-bit table), mixed
    wavefronts (wide / full / rows-only keys and the golden edge vectors in one sorted list), the promoted combs byte for byte against
    the host builder, and forgetting the promotions with the cache."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    emul.sbve_key_cache.argtypes = [ctypes.c_int, ctypes.c_uint32]
    emul.sbve_hot_keys.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    emul.sbve_hot_stats.argtypes = [ctypes.c_void_p]
    emul.sbve_hot_comb_mismatches.argtypes = [ctypes.c_uint32]
    emul.sbve_hot_comb_mismatches.restype = ctypes.c_size_t
    emul.sbve_set_full_table_min.argtypes = [ctypes.c_uint32]
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    wblob = [v["accept"] for v in vs]
    stats = (ctypes.c_uint32 * 4)()
    hs = (ctypes.c_uint32 * 4)()

    def batch(seed, n, nkeys):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer((n + 7) // 8)
        oracle.sbvo_gen_batch(seed, n, nkeys, 6, tup, exp, 4)
        return tup.raw, _bitmap_list(exp.raw, n)

    def run(allt, want):
        total = len(allt) // 160
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_p256_verify_batch_grouped(allt, total, bm, 4, 256, 12, stats)
        got = _bitmap_list(bm.raw, total)
        assert got == want, [i for i in range(total) if got[i] != want[i]][:8]
        emul.sbve_hot_stats(hs)
        return list(hs)

    hot, whot = batch(0xB1, 600, 3)             # 3 keys x 200 uses
    luke, wluke = batch(0xB2, 300, 20)          # 20 keys x 15 uses: rows only
    try:
        emul.sbve_set_full_table_min(64)
        emul.sbve_set_group_chunks(2)
        emul.sbve_key_cache(1, 64)
        emul.sbve_hot_keys(2, 250)              # a pool of two combs, promotion from 250 grouped tuples on
        h = run(hot + luke, whot + wluke)       # cold: 200 hits per hot key
        assert h[:3] == [0, 2, 0], h
        h = run(hot, whot)                      # warm: 400 hits -> three keys ask, two combs exist
        assert h[:3] == [2, 2, 0], h
        assert emul.sbve_hot_comb_mismatches(0) == 0 and emul.sbve_hot_comb_mismatches(1) == 0
        assert emul.sbve_hot_comb_mismatches(2) == ctypes.c_size_t(-1).value
        h = run(hot, whot)                      # promoted: two of three keys' runs through the wide pass (minus the seams' wavefronts)
        assert h[0] == 2 and 200 <= h[2] <= 400, h
        # mixed wavefronts: wide, full (the third hot key), rows-only keys and the edge vectors (keys that are no point, r / s out of range ...)
        for chunks in (1, 3):
            emul.sbve_set_group_chunks(chunks)
            h = run(luke[:160 * 150] + hot + blob + luke[160 * 150:], wluke[:150] + whot + wblob + wluke[150:])
            assert h[0] == 2 and h[2] >= 128, h
        h = run(hot[:160 * 90], whot[:90])      # a few tuples per promoted key: still the wide pass where a wavefront is all theirs
        assert h[0] == 2
        emul.sbve_key_cache(1, 64)              # the cache forgets its slots: so must the promotions
        h = run(hot, whot)
        assert h[:3] == [0, 2, 0], h
        # a promoted key whose 8-bit table holds ROWS ONLY (no batch gave it full_min uses) in a wavefront it shares with a full-table
        # key that owns no wide comb: its lanes must go through the rows-only pass, never through the chunks' launches (found by the GPU
        # tier in round 5: a 2^18-tuple piece holds 256 +- a few tuples per signer, the line between the classes)
        emul.sbve_key_cache(1, 64)
        emul.sbve_set_full_table_min(10**6)     # nobody earns a full table ...
        emul.sbve_hot_keys(3, 250)
        for _ in range(2):
            run(hot, whot)                      # ... and the three hot keys are promoted with rows only
        assert list(hs)[0] == 3
        solo, wsolo = batch(0xB3, 100, 1)       # one more key, 100 tuples: a full table from now on, no wide comb (the pool is spent)
        emul.sbve_set_full_table_min(64)
        for chunks in (1, 2, 3):
            emul.sbve_set_group_chunks(chunks)
            for cut in (60, 90):                # ~20 / ~30 tuples per promoted key: below full_min, their cached tables stay rows only
                h = run(hot[:160 * cut] + solo, whot[:cut] + wsolo)
                assert h[0] == 3, h
                h = run(solo + hot[:160 * cut], wsolo + whot[:cut])
        emul.sbve_set_full_table_min(64)
        emul.sbve_key_cache(1, 64)
        emul.sbve_hot_keys(2, 250)
        emul.sbve_hot_keys(0, 250)              # off: nothing is counted or promoted
        for _ in range(3):
            h = run(hot, whot)
        assert h[:3] == [0, 0, 0], h
    finally:
        emul.sbve_hot_keys(0, 4096)
        emul.sbve_key_cache(0, 0)
        emul.sbve_set_full_table_min(256)
        emul.sbve_set_group_chunks(3)


def test_hot_keys_life_cycle_decay_and_eviction(emul, oracle):
    """Round 6, p256_group.h "life cycle of the hot keys" (ADVICE r5): every SBV_HOT_DECAY_EVERY-th batch halves every
    count, and with the pool full a slot that has earned a comb takes the coldest owner's — only when that owner's count is at most half
    its own.  The lane functions are the kernels' own (k_group_table_class, k_hot_decay, k_promote_select / _evict / _publish call them);
    verdicts are checked on every batch, the re-assigned comb against the host builder."""
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    emul.sbve_key_cache.argtypes = [ctypes.c_int, ctypes.c_uint32]
    emul.sbve_hot_keys.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    emul.sbve_hot_stats.argtypes = [ctypes.c_void_p]
    emul.sbve_hot_life.argtypes = [ctypes.c_void_p]
    emul.sbve_hot_hits_of_key.argtypes = [ctypes.c_char_p]
    emul.sbve_hot_hits_of_key.restype = ctypes.c_uint32
    emul.sbve_hot_wide_of_key.argtypes = [ctypes.c_char_p]
    emul.sbve_hot_wide_of_key.restype = ctypes.c_uint32
    emul.sbve_hot_comb_mismatches.argtypes = [ctypes.c_uint32]
    emul.sbve_hot_comb_mismatches.restype = ctypes.c_size_t
    emul.sbve_set_full_table_min.argtypes = [ctypes.c_uint32]
    NONE = 0xFFFFFFFF
    stats = (ctypes.c_uint32 * 4)()
    hs = (ctypes.c_uint32 * 4)()
    life = (ctypes.c_uint32 * 2)()

    def batch(seed, n, nkeys, invalid_one_in=6):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer((n + 7) // 8)
        oracle.sbvo_gen_batch(seed, n, nkeys, invalid_one_in, tup, exp, 4)
        return tup.raw, _bitmap_list(exp.raw, n)

    def run(allt, want):
        total = len(allt) // 160
        bm = ctypes.create_string_buffer((total + 7) // 8)
        emul.sbve_p256_verify_batch_grouped(allt, total, bm, 4, 256, 12, stats)
        got = _bitmap_list(bm.raw, total)
        assert got == want, [i for i in range(total) if got[i] != want[i]][:8]
        emul.sbve_hot_stats(hs)
        emul.sbve_hot_life(life)
        return list(hs)

    def keys_of(blob):
        count = {}
        for i in range(len(blob) // 160):
            k = blob[160 * i + 96:160 * i + 160]
            count[k] = count.get(k, 0) + 1
        return sorted(count, key=lambda k: -count[k]), count

    ab, wab = batch(0xC1, 400, 2, 0)            # keys A, B: 200 uses each, every signature valid
    cc, wcc = batch(0xC2, 200, 1, 0)            # key C
    (ka, kb), nab = keys_of(ab)
    (kc_,), ncc = keys_of(cc)
    assert nab[ka] == nab[kb] == 200 and ncc[kc_] == 200
    try:
        emul.sbve_set_full_table_min(64)
        emul.sbve_set_group_chunks(2)
        emul.sbve_key_cache(1, 64)
        emul.sbve_hot_keys(2, 250)
        for _ in range(3):
            h = run(ab, wab)
        assert h[0] == 2 and emul.sbve_hot_wide_of_key(ka) != NONE and emul.sbve_hot_wide_of_key(kb) != NONE      # A and B own the pool
        assert emul.sbve_hot_hits_of_key(ka) == emul.sbve_hot_hits_of_key(kb) == 600
        # C arrives: the pool is full; it earns a comb only once its count is twice the coldest owner's — A and B no longer sign
        evicted_at = None
        for i in range(10):
            h = run(cc, wcc)
            if life[0] and evicted_at is None:
                evicted_at = i
        assert life[0] == 1                                              # exactly one comb changed owner ...
        assert evicted_at == 5                                           # ... in the batch that took C to 1200 = twice an owner's 600 (hysteresis)
        wc = emul.sbve_hot_wide_of_key(kc_)
        assert wc != NONE and emul.sbve_hot_comb_mismatches(wc) == 0     # the re-used comb is byte for byte C's
        assert sorted([emul.sbve_hot_wide_of_key(ka) == NONE, emul.sbve_hot_wide_of_key(kb) == NONE]) == [False, True]
        h = run(ab + cc, wab + wcc)                                      # the victim is served from its 8-bit table again, same verdicts
        assert h[2] > 0                                                  # and C's lanes take the wide pass
        assert life[1] == 14 and emul.sbve_hot_hits_of_key(kc_) == 2200
        run(cc, wcc)                                                     # tick 15: the clock sweep, then this batch's 200
        assert life[1] == 15 and emul.sbve_hot_hits_of_key(kc_) == 2400 // 2
        assert emul.sbve_hot_hits_of_key(ka) == 800 // 2
        for _ in range(16):
            run(ab[:160 * 8], wab[:8])                                   # 16 small batches without C: tick 31 halves again, nothing is added to C
        assert life[1] == 31 and emul.sbve_hot_hits_of_key(kc_) == 600
    finally:
        emul.sbve_hot_keys(0, 4096)
        emul.sbve_key_cache(0, 0)
        emul.sbve_set_full_table_min(256)
        emul.sbve_set_group_chunks(3)
