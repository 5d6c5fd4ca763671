"""GPU tier (`-m gpu`): the HIP path, called through the C-ABI (include/sbv.h), against the
oracle on the same inputs.  Bit-exact: the accept bitmap must equal the oracle's byte for byte.

The reference's own tests pin only control flow at this seam; these tests pin the
arithmetic: golden vectors (tests/golden), seeded synthetic batches at ragged sizes, and the
full 2^20 batch of BASELINE.json config[1]."""
import ctypes
import hashlib
import os
import random

import pytest

import consensus_amd as sbv
import p256_py as ec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    sbv.init(0)          # raises (loudly) when the HIP library or a gfx950 device is missing
    yield sbv
    sbv.shutdown()


def _expect(oracle, tuples: bytes, n: int, threads: int = 0) -> bytes:
    bm = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    oracle.sbvo_p256_verify_batch(tuples, n, bm, threads or (os.cpu_count() or 1))
    return bm.raw[:(n + 7) // 8]


def test_golden_tuple_vectors(gpu, golden_vectors):
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    got = sbv.bitmap_to_list(gpu.verify_batch(blob), len(vs))
    bad = [v["name"] for v, g in zip(vs, got) if g != v["accept"]]
    assert not bad, bad


def test_golden_asn1_vectors_via_parse_der(gpu, golden_vectors):
    """VerifyASN1-level vectors: host DER parse + hash truncation, then the kernel."""
    vs = [v for v in golden_vectors if v["kind"] == "asn1"]
    tuples = []
    for v in vs:
        rs = gpu.parse_der(bytes.fromhex(v["sig"])) or bytes(64)   # parse failure -> r = s = 0
        h = bytes.fromhex(v["hash"])
        h32 = h[:32] if len(h) >= 32 else bytes(32 - len(h)) + h
        tuples.append(rs + h32 + bytes.fromhex(v["qx"]) + bytes.fromhex(v["qy"]))
    got = sbv.bitmap_to_list(gpu.verify_batch(b"".join(tuples)), len(vs))
    bad = [v["name"] for v, g in zip(vs, got) if g != v["accept"]]
    assert not bad, bad


def test_rfc6979_known_answers(gpu, rfc6979):
    qx, qy = bytes.fromhex(rfc6979["qx"]), bytes.fromhex(rfc6979["qy"])
    tuples = []
    for sig in rfc6979["signatures"]:
        h = bytes.fromhex(sig["hash"])
        h32 = h[:32] if len(h) >= 32 else bytes(32 - len(h)) + h
        tuples.append(bytes.fromhex(sig["r"]) + bytes.fromhex(sig["s"]) + h32 + qx + qy)
    n = len(tuples)
    assert gpu.verify_batch(b"".join(tuples)) == bytes([0xFF] * (n // 8)) + (bytes([(1 << (n % 8)) - 1]) if n % 8 else b"")


@pytest.mark.parametrize("n", [0, 1, 7, 8, 63, 64, 65, 255, 256, 257, 1000, 4097, 20000])
def test_ragged_sizes_match_oracle(gpu, oracle, n):
    if n == 0:
        assert gpu.verify_batch(b"", 0) == b""
        return
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0xA000 + n, n, 33, 3, tup, exp, os.cpu_count() or 1)
    got = gpu.verify_batch(tup.raw, n)
    assert got == exp.raw[:(n + 7) // 8]
    assert got == _expect(oracle, tup.raw, n)


def test_garbage_and_all_invalid(gpu, oracle):
    rng = random.Random(99)
    n = 3000
    junk = bytes(rng.getrandbits(8) for _ in range(160 * n))
    assert gpu.verify_batch(junk, n) == _expect(oracle, junk, n) == bytes((n + 7) // 8)
    zeros = bytes(160 * 100)
    assert gpu.verify_batch(zeros, 100) == bytes(13)
    ones = b"\xff" * (160 * 100)
    assert gpu.verify_batch(ones, 100) == bytes(13)


def test_same_key_and_same_signature_lanes(gpu, oracle):
    """All lanes of a wavefront with identical data, and with one key (consenter-style)."""
    n = 512
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer(n // 8)
    oracle.sbvo_gen_batch(7, n, 1, 0, tup, exp, 4)              # one key, all valid
    assert gpu.verify_batch(tup.raw, n) == exp.raw == b"\xff" * (n // 8)
    same = tup.raw[:160] * n
    assert gpu.verify_batch(same, n) == b"\xff" * (n // 8)


def test_full_batch_2_20_bitmap_equals_generator_and_checksum(gpu, oracle):
    """BASELINE.json config[1]: 2^20 tuples, 1024 keys, 7/8 valid + 1/8 single-bit-corrupted.
    Full-size properties: bitmap == the generator's verdicts (oracle-checked on every corrupted
    tuple), popcount == 7n/8, and a SHA-256 of the bitmap agrees with the oracle's own bitmap on
    a 64 Ki-tuple slice (the oracle cannot redo 2^20 in seconds)."""
    n = 1 << 20
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer(n // 8)
    oracle.sbvo_gen_batch(0x5B7F2026, n, 1024, 8, tup, exp, os.cpu_count() or 1)
    got = ctypes.create_string_buffer(n // 8)
    gpu.verify_batch_ptr(ctypes.addressof(tup), n, ctypes.addressof(got))
    assert got.raw == exp.raw
    assert sum(bin(b).count("1") for b in got.raw) == n - n // 8
    m = 1 << 16
    sl = _expect(oracle, tup.raw[:160 * m], m)
    assert hashlib.sha256(got.raw[:m // 8]).digest() == hashlib.sha256(sl).digest()
    t = gpu.last_timing()
    print(f"\n[2^20 batch] h2d {t.h2d_us:.0f} us  prep {t.prep_us:.0f} us  verify {t.verify_us:.0f} us  "
          f"d2h {t.d2h_us:.0f} us  -> {n / (t.prep_us + t.verify_us) :.2f} M verifies/s (kernels)")


def test_batches_larger_than_one_launch_are_chunked(gpu):
    """More tuples than one launch holds (2^21): both entries walk the batch in chunks, each chunk grouped by key on
    its own; a chunk boundary that is not a multiple of 64 tuples away from the end must not disturb the bitmap."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import numpy as np
    import torch
    import synth
    base_n = 1 << 20
    tuples, valid = synth.gen_batch(0x5B7F2026, base_n)
    n = (1 << 21) + (1 << 19) + 37
    reps = -(-n // base_n)
    big = np.tile(tuples.reshape(base_n, 160), (reps, 1))[:n].copy()
    bits = np.tile(np.unpackbits(valid, bitorder="little")[:base_n], reps)[:n]
    want = np.packbits(bits, bitorder="little")
    out = np.zeros((n + 7) // 8, dtype=np.uint8)
    gpu.verify_batch_ptr(big.ctypes.data, n, out.ctypes.data)
    assert (out == want).all()
    d_t = torch.from_numpy(big.reshape(-1)).cuda()
    d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
    gpu.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (d_b.cpu().numpy() == want).all()


def test_device_pointer_entry_with_torch(gpu, oracle):
    import torch
    n = 10000
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0xD0D0, n, 16, 4, tup, exp, os.cpu_count() or 1)
    d_t = torch.frombuffer(bytearray(tup.raw), dtype=torch.uint8).cuda()
    d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    gpu.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert bytes(d_b.cpu().numpy().tobytes()) == exp.raw[:(n + 7) // 8]
    # twice in a row on different streams: the internal scratch is ordered by an event
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        d_b2 = torch.zeros_like(d_b)
        gpu.verify_batch_dev(d_t.data_ptr(), n, d_b2.data_ptr(), s2.cuda_stream)
    gpu.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    assert bytes(d_b2.cpu().numpy().tobytes()) == bytes(d_b.cpu().numpy().tobytes()) == exp.raw[:(n + 7) // 8]


def _split_keyed(tuples: bytes):
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


def test_registered_key_form_equals_generic_verdicts(gpu, oracle, golden_vectors):
    """sbv_p256_register_keys + sbv_p256_verify_batch_keyed: same verdicts as the generic entry on the
    golden vectors (invalid keys become invalid slots) and on a seeded batch; re-registration is
    idempotent; out-of-range slots reject."""
    gpu.clear_keys()
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    n = 34000                                   # > 32768 in total: the one-lane-per-signature kernel
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x4B45, n, 37, 3, tup, exp, os.cpu_count() or 1)
    allt = blob + tup.raw
    total = len(allt) // 160
    want = [v["accept"] for v in vs] + sbv.bitmap_to_list(exp.raw, n)
    rsh, slots, keys = _split_keyed(allt)
    reg = gpu.register_keys(keys)
    assert reg == list(range(len(keys))) and gpu.key_count() == len(keys)
    assert gpu.register_keys(keys[:5]) == reg[:5] and gpu.key_count() == len(keys)     # idempotent
    got = sbv.bitmap_to_list(gpu.verify_batch_keyed(rsh, [reg[s] for s in slots], total), total)
    bad = [i for i in range(total) if got[i] != want[i]]
    assert not bad, bad[:10]
    assert got == sbv.bitmap_to_list(gpu.verify_batch(allt, total), total)
    # batches <= 32768 take the latency kernel (8 lanes per signature, k_p256_verify_keyed_coop): the golden vectors
    # (first in the blob) and every ragged size around the 8-signatures-per-wavefront granularity
    for m in (1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, len(vs), len(vs) + 777, 5000, 8192, 32768):
        gotm = sbv.bitmap_to_list(gpu.verify_batch_keyed(rsh[:96 * m], [reg[s] for s in slots[:m]], m), m)
        badm = [i for i in range(m) if gotm[i] != want[i]]
        assert not badm, (m, badm[:10])
    # batches <= 32 take the one-launch latency form (host_prep_small + k_p256_verify_prepared_small: stage A on the host half
    # with one inversion per call, 16 lanes per signature, records and verdicts in mapped host memory): every golden vector
    # through it, 32 at a time, 15 at a time (a commit quorum: out-of-range r / s share an inversion with honest ones), 1 at a time
    for width in (32, 15, 1):
        for lo in range(0, len(vs), width):
            m = min(width, len(vs) - lo)
            gotm = sbv.bitmap_to_list(gpu.verify_batch_keyed(rsh[96 * lo:96 * (lo + m)], [reg[s] for s in slots[lo:lo + m]], m), m)
            badm = [vs[lo + i]["name"] for i in range(m) if gotm[i] != want[lo + i]]
            assert not badm, (width, lo, badm)
    oob = gpu.verify_batch_keyed(rsh[:96 * 3], [len(keys), 2**32 - 1, len(keys) + 7], 3)
    assert oob == b"\x00"
    gpu.clear_keys()
    assert gpu.key_count() == 0


def test_wide_combs_of_consenter_keys_on_gpu(gpu, oracle, golden_vectors):
    """sbv_p256_widen_keys / sbv_p256_wide_keys (round 4): registered slots named as consenters' get a second, wide comb; wavefronts
    whose signatures all belong to wide slots take 13 + ceil(257 / bits) additions, every other wavefront the 8-bit combs.  The
    golden tuple vectors (their invalid keys included: a widened slot of a key that is no point still rejects), forged signatures
    whose u2 walks the edges of the wide recoding (sign flip at 2^255, carry window of a width dividing 256) and a seeded batch,
    through the one-lane kernel (> 32768), the 8-lane latency kernel (ragged sizes) and the one-launch form (<= 32), with some
    slots wide, all of them, another width (rebuilt in place), and the feature off: always the oracle's verdicts."""
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    n = 36000
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x51DE, n, 6, 3, tup, exp, os.cpu_count() or 1)
    rng = random.Random(0x51DE)
    d = rng.randrange(1, ec.N)
    Q = ec.pt_mul(d, ec.G)
    forged = []
    for bits in (16, 18):
        windows = -(-257 // bits)
        S = sum(1 << (bits * j + bits - 1) for j in range(windows - 1))
        T = (1 << (bits * (windows - 1))) - S
        for u2 in (T - 1, T, T + 1, 2**255 - 1, 2**255, 2**255 + 1, ec.N - 1, ec.N - 2, 1, 2, (ec.N - 1) // 2, (ec.N + 1) // 2, ec.N - T, ec.N - T + 1,
                   2**255 - 2**239, 2**255 - 2**239 + 1, (1 << (bits - 1)), (1 << (bits - 1)) - 1, (1 << bits) - 1, 1 << bits):
            u2 %= ec.N
            if u2 == 0:
                continue
            u1 = rng.randrange(0, ec.N)
            Rp = ec.pt_add(ec.pt_mul(u1, ec.G), ec.pt_mul(u2, Q))
            r = Rp[0] % ec.N
            s_ = r * pow(u2, -1, ec.N) % ec.N
            e = u1 * s_ % ec.N
            forged.append(r.to_bytes(32, "big") + s_.to_bytes(32, "big") + e.to_bytes(32, "big") + Q[0].to_bytes(32, "big") + Q[1].to_bytes(32, "big"))
            bad = bytearray(forged[-1]); bad[70] ^= 4
            forged.append(bytes(bad))
    nf = len(forged)
    allt = b"".join(forged) + blob + tup.raw                    # forged key first, then the golden keys, then the 6 seeded ones
    total = len(allt) // 160
    want = [True, False] * (nf // 2) + [v["accept"] for v in vs] + sbv.bitmap_to_list(exp.raw, n)
    assert [bool(oracle.sbvo_p256_verify_tuple(t)) for t in forged] == want[:nf]
    rsh, slots, keys = _split_keyed(allt)
    gpu.clear_keys()

    def check(tag):
        sl = [reg[s] for s in slots]
        got = sbv.bitmap_to_list(gpu.verify_batch_keyed(rsh, sl, total), total)                   # > 32768: one lane per signature
        bad = [i for i in range(total) if got[i] != want[i]]
        assert not bad, (tag, bad[:10])
        lo = nf + len(vs)                                                                          # the seeded part: its keys are wide in every configuration
        for a, m in ((0, 1), (0, 7), (0, nf), (0, nf + len(vs) + 100), (lo, 8), (lo, 64), (lo + 3, 5000), (lo, 32768)):
            gotm = sbv.bitmap_to_list(gpu.verify_batch_keyed(rsh[96 * a:96 * (a + m)], sl[a:a + m], m), m)
            badm = [i for i in range(m) if gotm[i] != want[a + i]]
            assert not badm, (tag, a, m, badm[:10])
        for width in (32, 15, 1):                                                                  # the one-launch form
            for a in list(range(0, nf + len(vs), width)) + [lo, lo + 32]:
                m = min(width, total - a)
                gotm = sbv.bitmap_to_list(gpu.verify_batch_keyed(rsh[96 * a:96 * (a + m)], sl[a:a + m], m), m)
                badm = [a + i for i in range(m) if gotm[i] != want[a + i]]
                assert not badm, (tag, width, a, badm)

    try:
        gpu.wide_keys(16, 64)
        reg = gpu.register_keys(keys)
        assert gpu.wide_key_stats()[0] == 0                      # registering alone builds no wide comb
        check("8-bit combs only")
        slot_of = dict(zip(keys, reg))
        from collections import Counter
        often = Counter(tup.raw[160 * i + 96:160 * i + 160] for i in range(n))
        seeded = [slot_of[k] for k, c in often.items() if c >= 64]          # the 6 signers (corrupted keys appear once each)
        assert len(seeded) == 6
        gkeys = list(dict.fromkeys(blob[160 * i + 96:160 * i + 160] for i in range(len(vs))))
        odd = [reg[0]] + [slot_of[k] for k in gkeys[::7]]        # the forged key and every 7th golden key (valid and invalid ones)
        gpu.widen_keys(seeded + odd)
        wide, bits, cap, kib = gpu.wide_key_stats()
        assert wide == len(set(seeded + odd)) and bits == 16 and cap == 64 and kib == 17 * 32768 * 64 // 1024
        gpu.widen_keys(seeded)                                   # idempotent
        assert gpu.wide_key_stats()[0] == wide
        # the combs are built on the device (p256_widetab29.h): byte for byte the host builder's tables — a seeded key, the forged
        # key, a golden key that is a point and one that is not (its comb is zeros on both sides)
        for sl in (seeded[0], seeded[5], reg[0], odd[1], odd[-1]):
            assert gpu.wide_selfcheck(sl), sl
        with pytest.raises(Exception):
            gpu.wide_selfcheck(max(reg) + 1)                     # no such wide slot
        check("16 bits, some slots")
        gpu.wide_keys(18, 64)                                    # another width: the same slots, rebuilt
        assert gpu.wide_key_stats()[:2] == (wide, 18)
        assert gpu.wide_selfcheck(seeded[1]) and gpu.wide_selfcheck(reg[0])
        check("18 bits, some slots")
        gpu.wide_keys(20, 64)                                    # 436 MB per key: the widest the entry takes
        assert gpu.wide_key_stats()[:2] == (wide, 20) and gpu.wide_selfcheck(seeded[2])
        check("20 bits, some slots")
        gpu.wide_keys(16, 8)                                     # a cap below what it holds: the first 8 stay
        assert gpu.wide_key_stats()[0] == 8
        check("16 bits, cap 8")
        with pytest.raises(Exception):
            gpu.widen_keys([len(keys) + 5])                      # not a registered slot
        gpu.wide_keys(0, 0)
        assert gpu.wide_key_stats()[0] == 0
        check("off")
        # the default policy: the width follows the number of wide keys — 20 bits up to 16 of them, 16 bits beyond (what was there is rebuilt)
        gpu.wide_keys(sbv.WIDE_BITS_AUTO, 64)
        gpu.widen_keys(seeded)
        assert gpu.wide_key_stats()[:3] == (6, 20, 64) and gpu.wide_selfcheck(seeded[3])
        check("auto, 6 keys")
        gpu.widen_keys(odd)
        assert gpu.wide_key_stats()[:2] == (wide, 16) and wide > 16 and gpu.wide_selfcheck(seeded[3]) and gpu.wide_selfcheck(odd[2])
        check("auto, more than 16 keys")
    finally:
        gpu.wide_keys()                                          # the default: auto, 64 keys
        gpu.clear_keys()


def test_registered_key_full_batch_2_20(gpu):
    """2^20 signatures against 1024 registered keys (the headline batch re-expressed with key slots)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import numpy as np
    import synth
    n = 1 << 20
    tuples, valid = synth.gen_batch(0x5B7F2026, n)
    t = tuples.reshape(n, 160)
    keys = [bytes(t[i, 96:160]) for i in range(1024)]         # tuple i uses key i % 1024 (uncorrupted ones)
    # corrupted tuples whose flip landed in Qx/Qy carry a key that is not registered: give them their own slot
    slots = np.arange(n, dtype=np.uint32) % 1024
    gpu.clear_keys()
    reg = gpu.register_keys(keys)
    index = {k: s for k, s in zip(keys, reg)}
    extra = []
    for i in np.nonzero(~np.unpackbits(valid, bitorder="little")[:n].astype(bool))[0]:
        k = bytes(t[i, 96:160])
        if k not in index:
            index[k] = gpu.register_keys([k])[0]
        slots[i] = index[k]
    rsh = np.ascontiguousarray(t[:, :96]).reshape(-1)
    out = np.zeros(n // 8, dtype=np.uint8)
    sbv._check(sbv.load().sbv_p256_verify_batch_keyed(rsh.ctypes.data, slots.ctypes.data, n, out.ctypes.data))
    assert (out == valid).all()
    tm = gpu.last_timing()
    print(f"\n[2^20 keyed] prep {tm.prep_us:.0f} us  verify {tm.verify_us:.0f} us -> {n / (tm.prep_us + tm.verify_us):.1f} M verifies/s (kernels)")
    gpu.clear_keys()


def test_exact_pass_forced(golden_vectors):
    """Experimental two-launch stage B (SBV_STAGEB_FAST=1): fast kernel, then the exact kernel on flagged
    wavefronts; SBV_FORCE_EXACT=1 flags all of them.  Verdicts must equal the default single-launch mode."""
    import subprocess
    import sys
    code = r'''
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import consensus_amd as sbv, synth
sbv.init(0)
vs = [v for v in json.load(open("tests/golden/p256_vectors.json"))["vectors"] if v["kind"] == "tuple"]
blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
got = sbv.bitmap_to_list(sbv.verify_batch(blob), len(vs))
assert got == [v["accept"] for v in vs]
t, valid = synth.gen_batch(0x77, 5000, 16, 4, cache=False)
assert sbv.verify_batch(t.tobytes(), 5000) == valid.tobytes()
keys = sorted(set(bytes(t[i * 160 + 96:i * 160 + 160]) for i in range(5000)))
reg = dict(zip(keys, sbv.register_keys(keys)))
rsh = b"".join(bytes(t[i * 160:i * 160 + 96]) for i in range(5000))
assert sbv.verify_batch_keyed(rsh, [reg[bytes(t[i * 160 + 96:i * 160 + 160])] for i in range(5000)], 5000) == valid.tobytes()
print("forced-exact ok")
'''
    env = dict(os.environ, SBV_FORCE_EXACT="1", SBV_STAGEB_FAST="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "forced-exact ok" in out.stdout, out.stdout + out.stderr


def test_device_message_frontend(gpu, oracle, golden_vectors):
    """sbv_p256_verify_msgs_keyed: SHA-256 + strict DER on the device == VerifyASN1 on (key, sha256(msg), sig)."""
    import hashlib
    import random
    rng = random.Random(4242)
    gpu.clear_keys()
    msgs, sigs, slots, want = [], [], [], []
    # the DER classes of the golden vectors, each against its own key and with a real message hash
    d = 0x1234567
    q = ec.pt_mul(d, ec.G)
    keyslot = gpu.register_keys([q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big")])[0]
    for i in range(3000):
        mlen = rng.choice([0, 1, 31, 55, 56, 64, 100, 300])
        m = bytes(rng.getrandbits(8) for _ in range(mlen))
        h = hashlib.sha256(m).digest()
        r, s = ec.sign(d, rng.randrange(1, ec.N), h)
        sig = ec.der_encode_sig(r, s)
        mode = i % 6
        if mode == 1:
            sig = sig[:-1]                       # truncated
        elif mode == 2:
            sig = sig + b"\x00"                  # trailing byte
        elif mode == 3:
            m = m + b"x"                         # different message
        elif mode == 4:
            sig = b"\x30\x81" + bytes([len(sig) - 2]) + sig[2:]   # non-minimal length
        msgs.append(m); sigs.append(sig); slots.append(keyslot)
        want.append(ec.verify_asn1(q[0], q[1], hashlib.sha256(m).digest(), sig))
    got = sbv.bitmap_to_list(gpu.verify_msgs_keyed(msgs, sigs, slots), len(msgs))
    bad = [i for i in range(len(msgs)) if got[i] != want[i]]
    assert not bad, bad[:10]
    assert sum(want) > 500 and sum(want) < 2500
    gpu.clear_keys()


def _der_class_cases(golden_vectors):
    """(name, message, signature bytes, 64-byte key, golden verdict or None) for the device parser: the 28 DER classes of
    tests/golden/p256_vectors.json — their hash IS SHA-256(b"der classes") (tests/golden/gen_p256_vectors.py), so the vectors' own bytes
    go through the device's SHA-256 + parser unchanged — and the same shapes applied to s (the golden set bends r): non-minimal,
    negative, 0xFF-padded, empty, 33 bytes, = N, long-form length, wrong tag."""
    import hashlib
    msg = b"der classes"
    vs = [v for v in golden_vectors if v["kind"] == "asn1" and v["class"] == "der"]
    assert len(vs) == 28 and all(v["hash"] == hashlib.sha256(msg).hexdigest() for v in vs)
    key = bytes.fromhex(vs[0]["qx"]) + bytes.fromhex(vs[0]["qy"])
    assert all(bytes.fromhex(v["qx"]) + bytes.fromhex(v["qy"]) == key for v in vs)
    cases = [(v["name"], msg, bytes.fromhex(v["sig"]), key, bool(v["accept"])) for v in vs]
    good = bytes.fromhex(next(v["sig"] for v in vs if v["name"] == "der_good"))
    rl = good[3]
    rmin, smin = good[4:4 + rl], good[4 + rl + 2:]
    assert good[2] == 2 and good[4 + rl] == 2 and good[4 + rl + 1] == len(smin)

    def der_len(n):
        return bytes([n]) if n < 128 else bytes([0x81, n])

    def der_int(b):
        return b"\x02" + der_len(len(b)) + b

    def der_seq(body):
        return b"\x30" + der_len(len(body)) + body

    nn = ec.N.to_bytes(32, "big")
    snb = smin[1:] if smin[0] == 0 else smin            # s without its sign byte: top bit set -> negative as an INTEGER
    extra = {
        "s_33_bytes": der_seq(der_int(rmin) + der_int(b"\x01" + bytes(32))),
        "s_eq_N": der_seq(der_int(rmin) + der_int(b"\x00" + nn)),
        "s_eq_N_minus_1": der_seq(der_int(rmin) + der_int(b"\x00" + (ec.N - 1).to_bytes(32, "big"))),
        "s_empty_int": der_seq(der_int(rmin) + der_int(b"")),
        "s_negative": der_seq(der_int(rmin) + der_int(snb if snb[0] & 0x80 else b"\x80" + snb)),
        "s_ff_padded": der_seq(der_int(rmin) + der_int(b"\xff" + (snb if snb[0] & 0x80 else b"\x80" + snb))),
        "s_int_len_long_form": der_seq(der_int(rmin) + b"\x02\x81" + bytes([len(smin)]) + smin),
        "s_wrong_int_tag": der_seq(der_int(rmin) + b"\x03" + der_int(smin)[1:]),
        "both_33_bytes": der_seq(der_int(b"\x01" + bytes(32)) + der_int(b"\x01" + bytes(32))),
        "r_34_bytes_leading_zeros": der_seq(der_int(b"\x00\x00" + rmin[-32:]) + der_int(smin)),
        "len_4_bytes": b"\x30\x84\x00\x00\x00" + bytes([len(good) - 2]) + good[2:],
        "len_2_bytes_nonminimal": b"\x30\x82\x00" + bytes([len(good) - 2]) + good[2:],
        "seq_in_seq": der_seq(good),
        "one_byte": b"\x30",
        "int_runs_past_seq": der_seq(der_int(rmin) + b"\x02" + bytes([len(smin) + 1]) + smin),
        "good_again": good,
    }
    cases += [(name, msg, sig, key, None) for name, sig in extra.items()]
    return cases


def test_device_der_parser_all_golden_classes(gpu, oracle, openssl_check, golden_vectors):
    """VERDICT r5 #4 / Weak #1: the 28 golden DER classes and the same shapes bent on s — negative and non-minimal INTEGERs, three
    integers, empty INTEGER, indefinite / 4- and 5-byte lengths, 33-byte r and s, r = N — through the DEVICE parser (k_msg_frontend:
    sbv_p256_verify_msgs_keyed, and the sharded front end) on the GPU, each verdict against the golden one, the oracle's
    sbvo_p256_verify_asn1 (types.Signature.Value: pkg/types/types.go:25-29; SURVEY 8c rule 1) and the strict-DER OpenSSL judge.
    The hash-LENGTH classes of the golden set cannot reach this entry by construction — the device computes SHA-256(Msg) itself,
    always 32 bytes; they are diffed on the GPU as tuples (test_golden_vectors) and through the host's left-pad / truncation."""
    import hashlib
    openssl_check.sbvssl_p256_verify_asn1.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    openssl_check.sbvssl_p256_der_is_strict.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    cases = _der_class_cases(golden_vectors)
    key = cases[0][3]
    h = hashlib.sha256(b"der classes").digest()
    want = []
    for name, msg, sig, k, golden in cases:
        o = bool(oracle.sbvo_p256_verify_asn1(k[:32], k[32:], h, 32, sig, len(sig)))
        j = bool(openssl_check.sbvssl_p256_verify_asn1(k[:32], k[32:], h, 32, sig, len(sig)))
        assert o == j, (name, o, j)                      # the oracle and the strict-DER OpenSSL judge agree on every class
        if golden is not None:
            assert o == golden, (name, o, golden)
        if o:
            assert openssl_check.sbvssl_p256_der_is_strict(sig, len(sig)) == 1, name
        want.append(o)
    assert sum(want) >= 4 and want.count(False) >= 35
    gpu.clear_keys()
    try:
        slot = gpu.register_keys([key])[0]
        # every class alone in a call, then all of them in one call (lanes of one wavefront parse different shapes), then interleaved
        # with 4 000 honest signatures so that the batch takes the throughput kernels instead of the latency form
        for i, (name, msg, sig, _, _) in enumerate(cases):
            got = sbv.bitmap_to_list(gpu.verify_msgs_keyed([msg], [sig], [slot]), 1)
            assert got == [want[i]], (name, got, want[i])
        msgs = [c[1] for c in cases]
        sigs = [c[2] for c in cases]
        got = sbv.bitmap_to_list(gpu.verify_msgs_keyed(msgs, sigs, [slot] * len(cases)), len(cases))
        assert got == want, [cases[i][0] for i in range(len(cases)) if got[i] != want[i]]
        good = next(c[2] for c in cases if c[0] == "der_good")
        big_m, big_s, big_w = [], [], []
        for rep in range(100):
            for i, c in enumerate(cases):
                big_m += [c[1], b"der classes"]
                big_s += [c[2], good]
                big_w += [want[i], True]
        got = sbv.bitmap_to_list(gpu.verify_msgs_keyed(big_m, big_s, [slot] * len(big_m)), len(big_m))
        assert got == big_w, [i for i in range(len(big_m)) if got[i] != big_w[i]][:10]
        # the sharded front end: the same batch in pieces (each piece carries a slice of the offset tables)
        os.environ["SBV_SHARD_PIECE_KEYED"] = "1024"
        sbv.shutdown()
        assert sbv.init_all() >= 1
        slot = sbv.register_keys([key])[0]
        got2, _, info = sbv.verify_msgs_keyed_sharded(big_m, big_s, [slot] * len(big_m))
        assert sbv.bitmap_to_list(got2, len(big_m)) == big_w
    finally:
        os.environ.pop("SBV_SHARD_PIECE_KEYED", None)
        sbv.shutdown()
        sbv.init(0)
        gpu.clear_keys()


def test_sharded_message_frontend_in_pieces(oracle):
    """sbv_p256_verify_msgs_keyed_sharded (round 5): 70 000 raw messages of 0..300 bytes + DER signatures + key slots over 5 registered
    keys, uploaded in 11-piece fashion (SBV_SHARD_PIECE_KEYED = 7168: each piece carries a slice of the caller's offset tables with
    its base subtracted on the device) == the one-launch front end == VerifyASN1's verdict by construction (signatures from the
    device's RFC 6979 signer; every 5th message altered after signing, every 11th signature truncated, every 13th slot unknown);
    quorum bits by distinct slot == the numpy twin; forced one-rank RCCL all-gather."""
    import hashlib
    import random
    from consensus_amd import shard
    rng = random.Random(777)
    os.environ["SBV_RCCL"] = "1"
    os.environ["SBV_SHARD_PIECE_KEYED"] = "7168"
    try:
        sbv.shutdown()
        assert sbv.init_all() >= 1
        n, nk, G, Qm = 70000, 5, 7, 3
        ds = [rng.randrange(1, ec.N) for _ in range(nk)]
        pubs = [ec.pt_mul(d, ec.G) for d in ds]
        slots_of = sbv.register_keys([q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big") for q in pubs])
        msgs = [bytes([i & 255, (i >> 8) & 255, (i >> 16) & 255]) * rng.choice([0, 1, 10, 18, 19, 21, 33, 100]) for i in range(n)]
        digests = b"".join(hashlib.sha256(m).digest() for m in msgs)
        kidx = [(i * 3 + i // G) % nk for i in range(n)]
        rs, okb = sbv.sign_batch(b"".join(d.to_bytes(32, "big") for d in ds), digests, kidx)
        assert okb == b"\x01" * n
        sigs, slots, want = [], [], []
        for i in range(n):
            sig = ec.der_encode_sig(int.from_bytes(rs[64 * i:64 * i + 32], "big"), int.from_bytes(rs[64 * i + 32:64 * i + 64], "big"))
            good = True
            if i % 5 == 2:
                msgs[i] = msgs[i] + b"!"
                good = False
            if i % 11 == 3:
                sig = sig[:-1]
                good = False
            slot = slots_of[kidx[i]]
            if i % 13 == 4:
                slot = 0xFFFFFFFF if i % 2 else 1000
                good = False
            sigs.append(sig); slots.append(slot); want.append(good)
        got, qb, info = sbv.verify_msgs_keyed_sharded(msgs, sigs, slots, G, Qm)
        bits = sbv.bitmap_to_list(got, n)
        assert bits == want, [i for i in range(n) if bits[i] != want[i]][:10]
        assert info.mode == 1 and info.h2d_us > 0
        assert qb == shard.quorum_bits_slots(slots, got, n, G, Qm)
        assert 0 < sum(sbv.bitmap_to_list(qb, n // G)) < n // G
        m = 30000                                   # the one-launch front end on a prefix (its own staging, no pieces)
        assert sbv.verify_msgs_keyed(msgs[:m], sigs[:m], slots[:m]) == got[:m // 8]
        # empty messages / empty signatures at piece boundaries, n not a multiple of anything
        m2 = 7168 * 2 + 5
        msgs2, sigs2 = list(msgs[:m2]), list(sigs[:m2])
        for j in (0, 7167, 7168, m2 - 1):
            sigs2[j] = b""
        got2, _, _ = sbv.verify_msgs_keyed_sharded(msgs2, sigs2, slots[:m2])
        assert sbv.bitmap_to_list(got2, m2) == [want[i] and i not in (0, 7167, 7168, m2 - 1) for i in range(m2)]
        # malformed offset tables: an entry out of order INSIDE a piece costs its neighbours' verdicts (their byte ranges are wrong:
        # rejected), never a read outside the upload and never another lane's verdict; a piece BOUNDARY out of order is refused
        mo, so, a, b = [0], [0], 0, 0
        for i in range(m2):
            a += len(msgs2[i]); b += len(sigs2[i])
            mo.append(a); so.append(b)
        bad_mo = list(mo)
        bad_mo[100] = mo[100] + (1 << 40)                      # tuples 99 and 100 see a range that leaves the upload / runs backwards
        got3, _, _ = sbv.verify_msgs_keyed_sharded(msgs2, sigs2, slots[:m2], offsets=(bad_mo, so))
        w3 = sbv.bitmap_to_list(got2, m2)
        w3[99] = w3[100] = False
        assert sbv.bitmap_to_list(got3, m2) == w3
        bad_so = list(so)
        bad_so[5120] = so[10240] + 5                           # 14 341 signatures go up as 3 equal pieces of 5 120: this boundary now lies behind the next one
        with pytest.raises(sbv.SbvError) as ei:
            sbv.verify_msgs_keyed_sharded(msgs2, sigs2, slots[:m2], offsets=(mo, bad_so))
        assert ei.value.code == -2
    finally:
        os.environ.pop("SBV_RCCL", None)
        os.environ.pop("SBV_SHARD_PIECE_KEYED", None)
        try:
            sbv.clear_keys()
        except sbv.SbvError:
            pass
        sbv.shutdown()
        sbv.init(0)


def test_in_step_key_grouping_equals_generic(gpu, oracle, golden_vectors):
    """sbv_p256_set_grouping: tuples grouped by key on the device, per-batch comb tables, registered-key kernel for
    the grouped ones — verdicts must equal the plain generic kernel in every configuration."""
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    blob = b"".join(bytes.fromhex(v["tuple"]) for v in vs)
    n = 20000
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    oracle.sbvo_gen_batch(0x6B, n, 23, 5, tup, exp, os.cpu_count() or 1)
    off = next(bytes.fromhex(v["tuple"]) for v in vs if v["name"] == "q_off_curve_y_plus_1")
    allt = blob + tup.raw + off * 100
    total = len(allt) // 160
    want = [v["accept"] for v in vs] + sbv.bitmap_to_list(exp.raw, n) + [False] * 100
    try:
        gpu.set_grouping(False)
        assert sbv.bitmap_to_list(gpu.verify_batch(allt, total), total) == want
        for min_count, max_groups in [(8, 64), (1, 4096), (8, 3), (1000000, 64)]:
            gpu.set_grouping(True, 1, min_count, max_groups)
            got = sbv.bitmap_to_list(gpu.verify_batch(allt, total), total)
            bad = [i for i in range(total) if got[i] != want[i]]
            assert not bad, (min_count, max_groups, bad[:8])
    finally:
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)


def test_coop_form_of_the_grouped_step_on_the_gpu(oracle, golden_vectors):
    """Batches up to 2^15 tuples finish in one launch of eight lanes per grouped tuple (k_group_coop; the default since round 4,
    SBV_GROUP_COOP_MAX=0 switches it off).  In child processes (the knob is read when the context is created), once as
    shipped and once with the phased form every larger batch takes: golden vectors + a seeded batch + a repeated invalid key,
    key cache cold and warm, thresholds that leave tuples on the doubling kernel; verdicts = the oracle's."""
    import subprocess
    import sys
    code = (
        "import ctypes, json, os, sys\n"
        "root = %r\n"
        "sys.path.insert(0, root)\n"
        "import consensus_amd as sbv\n"
        "oracle = ctypes.CDLL(os.path.join(root, 'oracle', 'libsbv_oracle.so'))\n"
        "oracle.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]\n"
        "vs = [v for v in json.load(open(os.path.join(root, 'tests', 'golden', 'p256_vectors.json')))['vectors'] if v['kind'] == 'tuple']\n"
        "n = 6000\n"
        "tup = ctypes.create_string_buffer(160 * n); exp = ctypes.create_string_buffer((n + 7) // 8)\n"
        "oracle.sbvo_gen_batch(0xC00B, n, 23, 5, tup, exp, os.cpu_count() or 1)\n"
        "off = next(bytes.fromhex(v['tuple']) for v in vs if v['name'] == 'q_off_curve_y_plus_1')\n"
        "allt = b''.join(bytes.fromhex(v['tuple']) for v in vs) + tup.raw + off * 100\n"
        "total = len(allt) // 160\n"
        "want = [v['accept'] for v in vs] + sbv.bitmap_to_list(exp.raw, n) + [False] * 100\n"
        "sbv.init(0)\n"
        "for cache in (False, True, True):\n"
        "    sbv.key_cache(cache)\n"
        "    for min_count, max_groups in [(8, 64), (1, 4096), (8, 3)]:\n"
        "        sbv.set_grouping(True, 1, min_count, max_groups)\n"
        "        got = sbv.bitmap_to_list(sbv.verify_batch(allt, total), total)\n"
        "        bad = [i for i in range(total) if got[i] != want[i]]\n"
        "        assert not bad, (cache, min_count, max_groups, bad[:8])\n"
        "print('ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    assert any(f.startswith("p256_vectors") for f in os.listdir(golden))
    for coop in ("32768", "0"):
        env = dict(os.environ, SBV_GROUP_COOP_MAX=coop)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (coop, r.stderr[-1500:])


def test_grouped_vs_ungrouped_full_batch(gpu):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import numpy as np
    import synth
    n = 1 << 20
    tuples, valid = synth.gen_batch(0x5B7F2026, n)
    out = np.zeros(n // 8, dtype=np.uint8)
    times = {}
    try:
        for mode in (True, False):
            gpu.set_grouping(mode, 131072, 64, 2048)
            out[:] = 0
            gpu.verify_batch_ptr(tuples.ctypes.data, n, out.ctypes.data)
            gpu.verify_batch_ptr(tuples.ctypes.data, n, out.ctypes.data)
            assert (out == valid).all(), mode
            tm = gpu.last_timing()
            times[mode] = (tm.prep_us, tm.verify_us)
    finally:
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)
    print(f"\n[2^20] grouped: prep {times[True][0]:.0f} us stageB {times[True][1]:.0f} us | ungrouped: prep {times[False][0]:.0f} us "
          f"stageB {times[False][1]:.0f} us")


def test_concurrent_host_pointer_calls_are_pipelined_and_correct(gpu, oracle):
    """The host-pointer entry keeps two staging slots and releases the context lock while it waits, so concurrent callers
    (the reference calls the Verifier from several goroutines: view.go:539-541, controller.go:239) overlap one call's
    upload with another's kernels.  Four threads, different batches and sizes (one of them larger than a launch), many
    rounds: every bitmap must be its own batch's."""
    import threading
    sizes = [300000, 4097, 70000, (1 << 21) + 777]
    batches = []
    for k, n in enumerate(sizes):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer((n + 7) // 8)
        oracle.sbvo_gen_batch(0xD00 + k, n, 50 + k, 5, tup, exp, os.cpu_count() or 1)
        batches.append((tup, exp.raw[:(n + 7) // 8], n))
    errors = []

    def worker(k):
        tup, exp, n = batches[k]
        got = ctypes.create_string_buffer((n + 7) // 8)
        for _ in range(4 if n > 1000000 else 12):
            gpu.verify_batch_ptr(ctypes.addressof(tup), n, ctypes.addressof(got))
            if got.raw != exp:
                errors.append((k, n))
                return

    th = [threading.Thread(target=worker, args=(k,)) for k in range(len(sizes))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_persistent_key_table_cache_on_gpu(gpu, oracle):
    """sbv_p256_key_cache: batch 1 builds the tables, batch 2 (other signatures of the same 300 keys) is all hits, a
    capacity smaller than the key set overflows into the per-batch area; verdicts always equal the oracle's, cache on or off."""
    n = 1 << 18

    def batch(seed, nkeys):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer(n // 8)
        oracle.sbvo_gen_batch(seed, n, nkeys, 7, tup, exp, os.cpu_count() or 1)
        return tup, exp.raw

    def run(tup, exp):
        got = ctypes.create_string_buffer(n // 8)
        gpu.verify_batch_ptr(ctypes.addressof(tup), n, ctypes.addressof(got))
        assert got.raw == exp
        return gpu.key_cache_stats()

    a, ea = batch(0xE1, 300)
    b, eb = batch(0xE1, 300)          # same seed: same keys; different corruption pattern? (same n -> identical) so use another too
    c, ec = batch(0xE2, 300)
    try:
        gpu.key_cache(False)
        gpu.key_cache(True, 4096)
        entries, hits, misses, cap = run(a, ea)
        assert hits == 0 and misses >= 300 and entries == misses and cap == 4096
        first = entries
        entries, hits, misses, cap = run(b, eb)
        assert misses == 0 and hits == first and entries == first
        entries, hits, misses, cap = run(c, ec)
        assert hits == 0 and misses >= 300 and entries == first + misses
        gpu.key_cache(False)
        gpu.key_cache(True, 256)      # smaller than one batch's key set: the overflow is rebuilt per batch
        entries, hits, misses, cap = run(a, ea)
        assert entries == 256 and cap == 256
        entries, hits, misses, cap = run(a, ea)
        assert hits == 256 and misses == first - 256
    finally:
        gpu.key_cache(False)
        gpu.key_cache(True, 4096)


def test_key_table_caches_survive_buffer_growth(gpu, oracle):
    """A batch larger than any before regrows the per-tuple arrays of the grouped step; the comb pools and key-table caches of the
    three schemes depend on (capacity, max_groups) only and must stay (round 4: the first GPU run of the Ed25519 cache test missed
    every key on its warm batch because that batch was 384 tuples longer than the cold one).  Another max_groups first: every
    buffer of the grouped step is then allocated afresh, sized for the small batch that follows."""
    oracle.sbvo_ed25519_gen_batch.argtypes = oracle.sbvo_k256_gen_batch.argtypes = oracle.sbvo_gen_batch.argtypes
    threads = os.cpu_count() or 1
    legs = [(sbv.SCHEME_P256, oracle.sbvo_gen_batch, 160, gpu.verify_batch),
            (sbv.SCHEME_SECP256K1, oracle.sbvo_k256_gen_batch, 160, gpu.secp256k1_verify_batch),
            (sbv.SCHEME_ED25519, oracle.sbvo_ed25519_gen_batch, 128, gpu.ed25519_verify_batch)]

    def batch(gen, width, n):
        tup, exp = ctypes.create_string_buffer(width * n), ctypes.create_string_buffer(n // 8)
        gen(0x6A0, n, 40, 7, tup, exp, threads)          # the same seed: the same 40 keys at every size
        return tup.raw, exp.raw

    try:
        gpu.set_grouping(True, 64, 64, 1024)
        for scheme, gen, width, verify in legs:
            gpu.key_cache(False, 0, scheme)
            gpu.key_cache(True, 512, scheme)
        small = {s: batch(g, w, 1 << 13) for s, g, w, _ in legs}
        big = {s: batch(g, w, 1 << 15) for s, g, w, _ in legs}
        first = {}
        for scheme, gen, width, verify in legs:             # cold, every scheme at 2^13: the buffers are sized for 8192 tuples
            assert verify(small[scheme][0], 1 << 13) == small[scheme][1]
            entries, hits, misses, cap = gpu.key_cache_stats(scheme)
            assert hits == 0 and misses >= 40 and entries == misses and cap == 512, (scheme, entries, hits, misses, cap)
            first[scheme] = entries
        for scheme, gen, width, verify in legs:             # four times the tuples: growth; every key must still be found
            assert verify(big[scheme][0], 1 << 15) == big[scheme][1]
            entries, hits, misses, cap = gpu.key_cache_stats(scheme)
            assert misses == 0 and hits == first[scheme] and entries == first[scheme], (scheme, entries, hits, misses)
    finally:
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)
        gpu.key_cache(False); gpu.key_cache(True, 4096)
        for scheme in (sbv.SCHEME_SECP256K1, sbv.SCHEME_ED25519):
            gpu.key_cache(False, 0, scheme); gpu.key_cache(True, 1024, scheme)


def test_key_sorted_list_and_compaction_order_give_the_same_bitmap():
    """The grouped step with the key-sorted list (default) and with the split's compaction order (SBV_GROUP_SORT=0, the
    first half of round 2's pipeline) on the same 2^18-tuple batch: both equal the generator's verdicts.  The switch is
    read at sbv_init, hence the child processes."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import consensus_amd as sbv, synth
sbv.init(0)
sbv.key_cache(False)
n = 1 << 18
t, valid = synth.gen_batch(0x50E7ED, n, 256, 8)
for _ in range(2):
    assert sbv.verify_batch(t.tobytes(), n) == valid.tobytes()
groups, grouped, ungrouped, rejected = sbv.last_group_stats()
assert 250 <= groups <= 256 and grouped + ungrouped + rejected == n and grouped > n * 9 // 10
print("sorted=%s ok" % os.environ.get("SBV_GROUP_SORT", "1"))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sort in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SBV_GROUP_SORT=sort), capture_output=True, text=True,
                             timeout=300, cwd=root)
        assert out.returncode == 0 and "sorted=%s ok" % sort in out.stdout, out.stdout + out.stderr


def test_table_classes_on_the_device(gpu, oracle, golden_vectors):
    """Round 5: one batch over keys of very different frequency — 8 hot keys (16 384 uses each: full 8-bit combs),
    4 000 lukewarm keys (32 uses: rows only, two additions per window), 30 000 keys used once (the one-lane kernel) and the golden
    edge vectors — cold with the cache off, cold and warm with it on, and an UPGRADE: a later batch in which 400 of the lukewarm
    keys have become hot fills their cached rows-only tables.  Every verdict == the generator's expectation (which the whole-batch
    oracle / OpenSSL tests pin) and == the pinned verdict of each golden vector; the class counters say which path served what."""
    import numpy as np
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    gold = np.frombuffer(b"".join(bytes.fromhex(v["tuple"]) for v in vs), dtype=np.uint8).reshape(len(vs), 160)
    gold_want = np.array([1 if v["accept"] else 0 for v in vs], dtype=np.uint8)

    def gen(seed, n, nkeys, inv=7):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer((n + 7) // 8)
        oracle.sbvo_gen_batch(seed, n, nkeys, inv, tup, exp, os.cpu_count() or 1)
        return (np.frombuffer(tup, dtype=np.uint8).reshape(n, 160).copy(),
                np.unpackbits(np.frombuffer(exp, dtype=np.uint8), bitorder="little")[:n].copy())

    def run(t, want):
        n = len(want)
        got = ctypes.create_string_buffer((n + 7) // 8)
        buf = np.ascontiguousarray(t).reshape(-1)
        gpu.verify_batch_ptr(buf.ctypes.data, n, ctypes.addressof(got))
        bits = np.unpackbits(np.frombuffer(got, dtype=np.uint8), bitorder="little")[:n]
        bad = np.nonzero(bits != want)[0]
        assert len(bad) == 0, bad[:8]
        return gpu.last_group_stats(), gpu.last_table_classes()

    hot, whot = gen(0x51, 8 * 16384, 8)
    luke, wluke = gen(0x52, 4000 * 32, 4000)
    once, wonce = gen(0x53, 30000, 30000)
    t = np.concatenate([hot, luke, once, gold])
    w = np.concatenate([whot, wluke, wonce, gold_want])
    perm = np.random.default_rng(5).permutation(len(w))
    t, w = t[perm], w[perm]
    try:
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)
        gpu.key_cache(False)
        st, cl = run(t, w)
        assert st[0] >= 8 + 3000 and cl[0] == 8 and cl[1] == 8, (st, cl)       # 8 full tables; ~91 % of the 32-use keys pass the soft threshold
        assert cl[2] >= 3000 * 32 * 0.8 and st[2] >= 25000, (st, cl)            # rows-only pass; single-use keys on the one-lane kernel
        gpu.key_cache(True)
        st, cl = run(t, w)                                                      # cold, tables kept
        st2, cl2 = run(t, w)                                                    # warm: nothing is built, the classes are remembered per slot
        assert cl2[0] == 8 and cl2[1] == 0 and st2[0] >= st[0], (st2, cl2)
        # upgrade: 400 of the lukewarm keys (the generator derives key i from (seed, i)) now sign 512 tuples each
        up, wup = gen(0x52, 400 * 512, 400)
        st3, cl3 = run(up, wup)
        assert cl3[0] >= 380 and cl3[1] >= 300, (st3, cl3)                      # filled now, from the rows an earlier batch cached
        st4, cl4 = run(t, w)                                                    # the mixed batch again: those keys' slots are full now
        assert cl4[0] >= 8 + 300 and cl4[1] == 0, (st4, cl4)
    finally:
        gpu.key_cache(True)
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)


def test_hot_keys_on_the_device(gpu, oracle, golden_vectors):
    """Round 5, p256_group.h "hot keys": cache slots that keep being hit are promoted to 16-bit combs built on the
    device, and later batches verify their tuples in the wide pass.  Cold -> warm -> promoted with a pool smaller than the demand
    (budget overflow: 12 hot keys, 8 combs), the promoted combs byte for byte against the host builder (sbv_p256_hot_selfcheck), mixed
    wavefronts (promoted keys, full 8-bit tables, rows-only keys, single-use keys and the golden edge vectors in one permuted batch),
    promotions forgotten with the cache, and the feature off.  Every verdict == the generator's expectation / the pinned verdicts."""
    import numpy as np
    vs = [v for v in golden_vectors if v["kind"] == "tuple"]
    gold = np.frombuffer(b"".join(bytes.fromhex(v["tuple"]) for v in vs), dtype=np.uint8).reshape(len(vs), 160)
    gold_want = np.array([1 if v["accept"] else 0 for v in vs], dtype=np.uint8)

    def gen(seed, n, nkeys, inv=7):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer((n + 7) // 8)
        oracle.sbvo_gen_batch(seed, n, nkeys, inv, tup, exp, os.cpu_count() or 1)
        return (np.frombuffer(tup, dtype=np.uint8).reshape(n, 160).copy(),
                np.unpackbits(np.frombuffer(exp, dtype=np.uint8), bitorder="little")[:n].copy())

    def run(t, want):
        n = len(want)
        got = ctypes.create_string_buffer((n + 7) // 8)
        buf = np.ascontiguousarray(t).reshape(-1)
        gpu.verify_batch_ptr(buf.ctypes.data, n, ctypes.addressof(got))
        bits = np.unpackbits(np.frombuffer(got, dtype=np.uint8), bitorder="little")[:n]
        bad = np.nonzero(bits != want)[0]
        assert len(bad) == 0, bad[:8]
        return gpu.hot_key_stats()

    hot, whot = gen(0x61, 12 * 4096, 12)            # 12 keys x 4096 uses
    mid, wmid = gen(0x62, 40 * 512, 40)             # full 8-bit tables, never hot enough
    luke, wluke = gen(0x63, 1000 * 32, 1000)        # rows only
    once, wonce = gen(0x64, 5000, 5000)             # the one-lane kernel
    mix = np.concatenate([hot, mid, luke, once, gold])
    wmix = np.concatenate([whot, wmid, wluke, wonce, gold_want])
    perm = np.random.default_rng(6).permutation(len(wmix))
    mix, wmix = mix[perm], wmix[perm]
    try:
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)
        gpu.key_cache(False)
        gpu.key_cache(True)                         # an empty cache
        gpu.hot_keys(8, 6000)                       # a pool of 8 combs; promotion from 6000 hits on
        h = run(hot, whot)                          # cold: 4096 hits per key
        assert h[:3] == (0, 8, 0), h
        h = run(mix, wmix)                          # warm: 8192 hits -> 12 keys ask at the end of this batch, 8 combs exist
        assert h[1] == 8 and h[2] == 0, h
        h = run(hot, whot)                          # promoted: 8 of the 12 keys' runs through the wide pass
        assert h[0] == 8 and 8 * 4096 * 0.8 <= h[2] <= 8 * 4096, h      # (a corrupted key byte moves a tuple out of its signer's run)
        for i in range(8):
            assert gpu.hot_selfcheck(i), i
        for _ in range(2):
            h = run(mix, wmix)                      # mixed wavefronts, twice (nothing left to promote: the pool is full)
            assert h[0] == 8 and h[2] >= 8 * 4096 * 0.75, h
        small, wsmall = hot[:3000], whot[:3000]     # the latency form (k_group_coop) serves small batches: no wide pass, same verdicts
        h = run(small, wsmall)
        assert h[0] == 8, h
        gpu.key_cache(False)                        # forgets the slots and with them the promotions
        gpu.key_cache(True)
        h = run(hot, whot)
        assert h[:3] == (0, 8, 0), h
        # promoted keys whose 8-bit tables hold ROWS ONLY (200 uses per batch: a full table takes 256) next to full-table keys without a
        # wide comb: a wavefront that mixes them belongs to the rows-only pass.  (The first full run of the tier with hot keys on found the
        # chunks' launches reading unfilled entries here; the emulator test of the same name now holds the scenario too.)  Batches above
        # 2^15 tuples: the latency form fills every table.
        gpu.hot_keys(64, 600)
        rows, wrows = gen(0x65, 200 * 200, 200)
        for _ in range(6):                          # ~188 hits per key and batch: promoted behind the fourth, the wide pass from the fifth on
            h = run(rows, wrows)
        assert h[0] == 64 and h[2] > 0, h
        solo, wsolo = gen(0x66, 20 * 400, 20)
        both, wboth = np.concatenate([rows, solo]), np.concatenate([wrows, wsolo])
        for seed in (1, 2, 3):
            perm2 = np.random.default_rng(seed).permutation(len(wboth))
            h = run(both[perm2], wboth[perm2])
            assert h[0] == 64, h
        gpu.hot_keys(0, 0)                          # off
        for _ in range(3):
            h = run(hot, whot)
        assert h[:3] == (0, 0, 0), h
        gpu.hot_keys(1024, 4096)                    # the default: 12 hot keys all fit; promoted behind the second batch (a signer's run is a little short of 4096)
        run(hot, whot)
        run(hot, whot)
        h = run(hot, whot)
        assert h[0] == 12 and h[1] >= 12 and h[2] >= 12 * 4096 * 0.8, h
        run(mix, wmix)
        assert gpu.hot_selfcheck(11)
    finally:
        gpu.hot_keys(1024, 4096)
        gpu.key_cache(True)
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)


def test_hot_key_pool_follows_a_changing_signer_set(gpu, oracle):
    """Round 6 (p256_group.h "life cycle of the hot keys") on the device: a pool of 8 wide combs, three disjoint sets
    of 8 signers one after another (a reconfiguration replaces the consenters: pkg/consensus/consensus.go:185-252).  The set that
    signs NOW takes the pool over from the set that stopped — k_promote_evict hands a comb over once its owner's count is at most half
    the newcomer's — every comb a new owner got is byte for byte the host builder's for ITS key (sbv_debug_hot_check), the clock sweep
    halves the counts, and every verdict of every batch is the generator's.  Then the rate: two batches after a rotation has settled the
    new set runs through the wide pass."""
    import numpy as np

    def gen(seed, n, nkeys, inv=0):
        tup = ctypes.create_string_buffer(160 * n)
        exp = ctypes.create_string_buffer((n + 7) // 8)
        oracle.sbvo_gen_batch(seed, n, nkeys, inv, tup, exp, os.cpu_count() or 1)
        return (np.frombuffer(tup, dtype=np.uint8).reshape(n, 160).copy(),
                np.unpackbits(np.frombuffer(exp, dtype=np.uint8), bitorder="little")[:n].copy())

    def run(t, want):
        n = len(want)
        got = ctypes.create_string_buffer((n + 7) // 8)
        buf = np.ascontiguousarray(t).reshape(-1)
        gpu.verify_batch_ptr(buf.ctypes.data, n, ctypes.addressof(got))
        bits = np.unpackbits(np.frombuffer(got, dtype=np.uint8), bitorder="little")[:n]
        bad = np.nonzero(bits != want)[0]
        assert len(bad) == 0, bad[:8]
        return gpu.hot_key_stats()

    sets = [gen(0x71 + k, 8 * 8192, 8, 9) for k in range(3)]          # 8 signers x 8192 signatures, 1/9 of them corrupted
    try:
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)
        gpu.key_cache(False)
        gpu.key_cache(True)
        gpu.hot_keys(8, 4096)
        for k, (t, w) in enumerate(sets):
            settled_at = None
            for call in range(48):
                h = run(t, w)
                if h[2] >= 8 * 8192 * 0.85:                  # (nearly) the whole batch through the wide pass: this set owns the pool
                    settled_at = call
                    break
            # set 0 finds an empty pool (promoted behind its first batch: the wide pass from the second on).  A later set has to out-count
            # the owners — twice their count, which the clock sweep halves every 16 batches while they are silent —, so it takes the pool
            # over after roughly as many batches as the owners had signed, never at once (hysteresis) and never never
            assert settled_at is not None and (settled_at <= 2 if k == 0 else 3 <= settled_at), (k, settled_at)
            for _ in range(2):
                h = run(t, w)
            assert h[0] == 8 and h[1] == 8 and h[2] >= 8 * 8192 * 0.85, (k, h)
            promoted, differ, _, _, _, inconsistent, shared, handed_out = gpu.debug_hot_check(0)
            assert promoted == 8 and handed_out == 8 and differ == 0 and inconsistent == 0 and shared == 0, (k, promoted, differ, inconsistent, shared)
        # all three sets in one batch: 24 hot signers, 8 combs — the pool stays with whoever holds it (hysteresis), verdicts as ever
        t3 = np.concatenate([s[0] for s in sets])
        w3 = np.concatenate([s[1] for s in sets])
        perm = np.random.default_rng(3).permutation(len(w3))
        for _ in range(3):
            h = run(t3[perm], w3[perm])
            assert h[0] == 8, h
        promoted, differ, _, _, _, inconsistent, shared, _ = gpu.debug_hot_check(0)
        assert promoted == 8 and differ == 0 and inconsistent == 0 and shared == 0
    finally:
        gpu.hot_keys(1024, 4096)
        gpu.key_cache(True)
        gpu.set_grouping(True, sbv.GROUP_MIN_BATCH_DEFAULT, 0, 0)


def test_lds_staged_form_of_the_q_phase_gives_the_same_verdicts(oracle, golden_vectors):
    """SBV_QPHASE_LDS=1 (round 6; VERDICT r5 #3, north_star "LDS-staged ... tables"): the chunks' launches stage a key-uniform workgroup's
    comb rows in LDS (two coalesced loads per lane and window instead of a 64-byte gather; two buffers, one barrier per window),
    workgroups at the seam of two keys take the gathers.  Measured 4 % slower per launch than the gathers and moving the same HBM bytes
    (profiles/r06/ab_qphase_lds_r06v.jsonl, pmc_qphase_lds_vs_gather_r06w.json) — it stays opt-in; here its verdicts: 2^18 tuples over
    300 keys (runs that end inside workgroups), the golden edge vectors spliced in, cold and warm, two chunk settings."""
    import subprocess
    import sys
    code = r"""
import ctypes, json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import consensus_amd as sbv
oracle = ctypes.CDLL(os.path.join("oracle", "libsbv_oracle.so"))
oracle.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
vs = [v for v in json.load(open(os.path.join("tests", "golden", "p256_vectors.json")))["vectors"] if v["kind"] == "tuple"]
n = 1 << 18
tup = ctypes.create_string_buffer(160 * n); exp = ctypes.create_string_buffer(n // 8)
oracle.sbvo_gen_batch(0x1D5, n, 300, 8, tup, exp, os.cpu_count() or 1)
t = np.frombuffer(tup, dtype=np.uint8).reshape(n, 160).copy()
w = np.unpackbits(np.frombuffer(exp, dtype=np.uint8), bitorder="little")[:n].copy()
for k, v in enumerate(vs):                       # the edge vectors spliced in, spread over the batch
    i = (k * 1201 + 7) % n
    t[i] = np.frombuffer(bytes.fromhex(v["tuple"]), dtype=np.uint8); w[i] = 1 if v["accept"] else 0
sbv.init(0)
for cache in (False, True, True):
    sbv.key_cache(cache)
    got = ctypes.create_string_buffer(n // 8)
    sbv.verify_batch_ptr(t.reshape(-1).ctypes.data, n, ctypes.addressof(got))
    bits = np.unpackbits(np.frombuffer(got, dtype=np.uint8), bitorder="little")[:n]
    bad = np.nonzero(bits != w)[0]
    assert len(bad) == 0, (cache, bad[:8])
print("lds ok", sbv.last_group_stats())
"""
    for chunks in ("2", "3"):
        env = dict(os.environ, SBV_QPHASE_LDS="1", SBV_GROUP_CHUNKS=chunks, SBV_GROUP_MIN_BATCH="64")
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert out.returncode == 0 and "lds ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
