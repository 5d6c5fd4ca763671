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


def _judge(openssl_check):
    openssl_check.sbvssl_p256_verify_asn1.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    openssl_check.sbvssl_p256_der_is_strict.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    return openssl_check


def test_encoding_classes_three_opinions(oracle, openssl_check, golden_vectors):
    """the 28 DER classes and the hash lengths 0..64 were confirmed by the builder's C restatement and the builder's
    Python twin only.  Third opinion: a strict-DER judge assembled from OpenSSL primitives the builder did not write
    (oracle/openssl_check.c: d2i_ECDSA_SIG -> everything consumed -> both integers non-negative -> i2d_ECDSA_SIG reproduces the
    input byte for byte -> ECDSA_do_verify on the hash as given).  oracle == twin == judge == the pinned verdict on every asn1
    vector, and on every hash length 0..64 for a valid and a spoiled signature."""
    j = _judge(openssl_check)
    seen_classes = set()
    for v in golden_vectors:
        if v["kind"] != "asn1":
            continue
        qx, qy, h, sig = (bytes.fromhex(v[k]) for k in ("qx", "qy", "hash", "sig"))
        a = bool(oracle.sbvo_p256_verify_asn1(qx, qy, h, len(h), sig, len(sig)))
        b = bool(ec.verify_asn1(int.from_bytes(qx, "big"), int.from_bytes(qy, "big"), h, sig))
        c = bool(j.sbvssl_p256_verify_asn1(qx, qy, h, len(h), sig, len(sig)))
        assert a == b == c == v["accept"], (v["name"], a, b, c)
        if v["class"] == "der":
            seen_classes.add(v["name"])
            strict = bool(j.sbvssl_p256_der_is_strict(sig, len(sig)))
            assert strict == (ec.parse_der_sig(sig) is not None), v["name"]
    assert len(seen_classes) == 28
    # hash lengths 0 .. 64: signatures made over the integer hashToNat derives, judged on the digest bytes as handed over
    rng = random.Random(64)
    d = rng.randrange(1, ec.N)
    q = ec.pt_mul(d, ec.G)
    qx, qy = q[0].to_bytes(32, "big"), q[1].to_bytes(32, "big")
    for hlen in range(65):
        h = bytes(rng.getrandbits(8) for _ in range(hlen))
        e = h[:32]                                     # leftmost 32 bytes; a shorter digest is the integer it spells
        r, s = ec.sign(d, rng.randrange(1, ec.N), e.rjust(32, b"\0") if hlen < 32 else e)
        sig = ec.der_encode_sig(r, s)
        for hh, sg in ((h, sig), (h + b"\x01" if hlen < 32 else bytes([h[0] ^ 1]) + h[1:], sig), (h, ec.der_encode_sig(r, (s + 1) % ec.N or 1))):
            a = bool(oracle.sbvo_p256_verify_asn1(qx, qy, hh, len(hh), sg, len(sg)))
            b = bool(ec.verify_asn1(q[0], q[1], hh, sg))
            c = bool(j.sbvssl_p256_verify_asn1(qx, qy, hh, len(hh), sg, len(sg)))
            assert a == b == c, (hlen, a, b, c)
        assert j.sbvssl_p256_verify_asn1(qx, qy, h, len(h), sig, len(sig)) == 1, hlen


def test_der_mutations_three_opinions(oracle, openssl_check, golden_vectors):
    """4000 random mutations of valid encodings (byte overwritten / deleted / inserted, truncation, and the classic BER liberties:
    non-minimal length forms, padded integers, negative integers, indefinite length): the strictness verdict of OpenSSL's
    decode-and-re-encode judge equals the twin's parse verdict, and the whole VerifyASN1 verdict of oracle, twin and judge agree
    (a mutation that leaves a strict encoding of OTHER integers is a verification question, not a parsing one)."""
    j = _judge(openssl_check)
    rng = random.Random(70)
    d = rng.randrange(1, ec.N)
    q = ec.pt_mul(d, ec.G)
    qx, qy = q[0].to_bytes(32, "big"), q[1].to_bytes(32, "big")
    n_strict = n_accept = 0
    for it in range(4000):
        h = bytes(rng.getrandbits(8) for _ in range(32))
        r, s = ec.sign(d, rng.randrange(1, ec.N), h)
        if it % 7 == 0:
            r >>= rng.choice([1, 8, 9, 17])            # short integers: other length bytes
        b = bytearray(ec.der_encode_sig(r, s))
        op = rng.randrange(9)
        if op == 0 and b:
            b[rng.randrange(len(b))] = rng.randrange(256)
        elif op == 1 and b:
            del b[rng.randrange(len(b))]
        elif op == 2:
            b.insert(rng.randrange(len(b) + 1), rng.randrange(256))
        elif op == 3:
            b = b[:rng.randrange(len(b) + 1)]
        elif op == 4:                                   # long-form length where the short form fits
            b = bytearray(b[:1] + b"\x81" + b[1:])
        elif op == 5:                                   # zero-padded first integer (non-minimal), lengths fixed up
            rl = b[3]
            b = bytearray(b[:1] + bytes([b[1] + 1]) + b[2:3] + bytes([rl + 1]) + b"\x00" + b[4:])
        elif op == 6:                                   # indefinite length
            b = bytearray(b[:1] + b"\x80" + b[2:] + b"\x00\x00")
        elif op == 7:                                   # the encoding as it is: must stay accepted
            pass
        else:                                           # first integer made negative (strip a leading zero byte if there is one)
            if b[4] == 0:
                b = bytearray(b[:1] + bytes([b[1] - 1]) + b[2:3] + bytes([b[3] - 1]) + b[5:])
            else:
                b[4] |= 0x80
        sig = bytes(b)
        strict = bool(j.sbvssl_p256_der_is_strict(sig, len(sig)))
        assert strict == (ec.parse_der_sig(sig) is not None), sig.hex()
        a = bool(oracle.sbvo_p256_verify_asn1(qx, qy, h, len(h), sig, len(sig)))
        bb = bool(ec.verify_asn1(q[0], q[1], h, sig))
        c = bool(j.sbvssl_p256_verify_asn1(qx, qy, h, len(h), sig, len(sig)))
        assert a == bb == c, (sig.hex(), a, bb, c)
        n_strict += strict
        n_accept += c
    assert 300 < n_accept < n_strict < 3500


def test_der_shapes_bent_on_s_oracle_equals_strict_openssl(oracle, openssl_check, golden_vectors):
    """The case list the GPU tier feeds through the DEVICE parser (tests/test_gpu_parity.py: _der_class_cases — the 28 golden DER
    classes plus the same shapes applied to s, 4-byte lengths, nested sequences): here the two CPU opinions on it, so that the GPU test's
    expectation is pinned before a GPU is involved."""
    import hashlib
    from test_gpu_parity import _der_class_cases
    openssl_check.sbvssl_p256_verify_asn1.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    openssl_check.sbvssl_p256_der_is_strict.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    h = hashlib.sha256(b"der classes").digest()
    cases = _der_class_cases(golden_vectors)
    assert len(cases) == 28 + 16
    accepted = []
    for name, msg, sig, k, golden in cases:
        o = bool(oracle.sbvo_p256_verify_asn1(k[:32], k[32:], h, 32, sig, len(sig)))
        j = bool(openssl_check.sbvssl_p256_verify_asn1(k[:32], k[32:], h, 32, sig, len(sig)))
        t = ec.verify_asn1(int.from_bytes(k[:32], "big"), int.from_bytes(k[32:], "big"), h, sig)
        assert o == j == t, (name, o, j, t)
        if golden is not None:
            assert o == golden, name
        if o:
            accepted.append(name)
    assert sorted(accepted) == sorted(["der_good", "der_both_high_bit_valid", "der_short_r_valid", "good_again"]), accepted
