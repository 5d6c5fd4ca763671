"""Pin the oracle (CPU tier): C restatement vs golden vectors, RFC 6979 KATs, the Python
big-int twin and OpenSSL.  The reference holds no vectors for this path;
these are what "the oracle is pinned against" means in oracle/p256_oracle.c's header."""
import ctypes
import hashlib
import random

import p256_py as ec


def test_rfc6979_known_answers(oracle, openssl_check, rfc6979):
    qx, qy = bytes.fromhex(rfc6979["qx"]), bytes.fromhex(rfc6979["qy"])
    assert len(rfc6979["signatures"]) >= 9
    for sig in rfc6979["signatures"]:
        h = bytes.fromhex(sig["hash"])
        assert hashlib.new(sig["hash_alg"], sig["message"].encode()).digest() == h
        r, s = int(sig["r"], 16), int(sig["s"], 16)
        der = ec.der_encode_sig(r, s)
        assert oracle.sbvo_p256_verify_asn1(qx, qy, h, len(h), der, len(der)) == 1, sig
        assert ec.verify_asn1(int.from_bytes(qx, "big"), int.from_bytes(qy, "big"), h, der)
        # tuple ABI: hash truncated / left-padded to 32 bytes
        h32 = h[:32] if len(h) >= 32 else bytes(32 - len(h)) + h
        t = ec.pack_tuple(r, s, h32, int.from_bytes(qx, "big"), int.from_bytes(qy, "big"))
        assert oracle.sbvo_p256_verify_tuple(t) == 1
        assert openssl_check.sbvssl_p256_verify_tuple(t) == 1
        # any single corrupted byte of s must be rejected
        t2 = bytearray(t); t2[40] ^= 0x10
        assert oracle.sbvo_p256_verify_tuple(bytes(t2)) == 0


def test_golden_vectors_c_oracle(oracle, golden_vectors):
    assert len(golden_vectors) > 200
    for v in golden_vectors:
        if v["kind"] == "tuple":
            got = oracle.sbvo_p256_verify_tuple(bytes.fromhex(v["tuple"]))
        else:
            h, sig = bytes.fromhex(v["hash"]), bytes.fromhex(v["sig"])
            got = oracle.sbvo_p256_verify_asn1(bytes.fromhex(v["qx"]), bytes.fromhex(v["qy"]), h, len(h), sig, len(sig))
        assert bool(got) == v["accept"], v["name"]


def test_golden_vectors_openssl_agrees_on_math_classes(openssl_check, golden_vectors):
    """OpenSSL is lenient on encodings, so it is only consulted on raw tuples."""
    n = 0
    for v in golden_vectors:
        if v["kind"] != "tuple":
            continue
        assert bool(openssl_check.sbvssl_p256_verify_tuple(bytes.fromhex(v["tuple"]))) == v["accept"], v["name"]
        n += 1
    assert n > 150


def test_der_parser_matches_twin(oracle, golden_vectors):
    rng = random.Random(7)
    sigs = [bytes.fromhex(v["sig"]) for v in golden_vectors if v["kind"] == "asn1"]
    # plus random mutations of valid encodings
    base = [s for s in sigs if ec.parse_der_sig(s) is not None]
    for _ in range(3000):
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
    for sig in sigs:
        out = ctypes.create_string_buffer(64)
        rc = oracle.sbvo_p256_parse_der(sig, len(sig), out)
        want = ec.parse_der_sig(sig)
        if want is None or len(want[0]) > 32 or len(want[1]) > 32:
            assert rc != 0, sig.hex()
        else:
            assert rc == 0, sig.hex()
            assert out.raw == want[0].rjust(32, b"\0") + want[1].rjust(32, b"\0")


def test_synthetic_batch_three_way(oracle, openssl_check):
    """Seeded synthetic batch (SURVEY.md §8d mix): oracle == OpenSSL == twin (sampled)."""
    n = 1024
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer(n // 8)
    oracle.sbvo_gen_batch(0x5B7F2026, n, 32, 8, tup, exp, 4)
    bm = ctypes.create_string_buffer(n // 8)
    oracle.sbvo_p256_verify_batch(tup, n, bm, 4)
    assert bm.raw == exp.raw
    acc = sum(bin(b).count("1") for b in bm.raw)
    assert acc == n - n // 8          # exactly the corrupted eighth is rejected
    for i in range(0, n, 5):
        t = tup.raw[160 * i:160 * i + 160]
        want = bool((bm.raw[i >> 3] >> (i & 7)) & 1)
        assert bool(openssl_check.sbvssl_p256_verify_tuple(t)) == want
        if i % 25 == 0:
            assert ec.verify_tuple(t) == want


def test_sha256_matches_hashlib(oracle):
    rng = random.Random(3)
    for ln in [0, 1, 31, 32, 55, 56, 57, 63, 64, 65, 119, 120, 128, 1000]:
        m = bytes(rng.randrange(256) for _ in range(ln))
        out = ctypes.create_string_buffer(32)
        oracle.sbvo_sha256(m, ln, out)
        assert out.raw == hashlib.sha256(m).digest()
