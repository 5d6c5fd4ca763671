#!/usr/bin/env python3
"""Generate tests/golden/ed25519_vectors.json: RFC 8032 §7.1 tests 1-3 plus seeded edge classes for Go
crypto/ed25519.Verify semantics (expected verdicts from oracle/ed25519_py.py).

Run from the repo root:  python tests/golden/gen_ed25519_vectors.py"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ed25519_py as ed  # noqa: E402

rng = random.Random(0xED25519)
P, L = ed.P, ed.L
vectors = []


def add(name, pk, msg, sig, cls, note=""):
    vectors.append({"name": name, "class": cls, "pk": pk.hex(), "msg": msg.hex(), "sig": sig.hex(),
                    "accept": ed.verify(pk, msg, sig), "note": note})


def enc_xy(x, y):
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


RFC = [("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60", "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
        "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
       ("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb", "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
        "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"),
       ("c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7", "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
        "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a")]
for i, (sk, pk, m, s) in enumerate(RFC):
    assert ed.public_key(bytes.fromhex(sk)) == bytes.fromhex(pk) and ed.sign(bytes.fromhex(sk), bytes.fromhex(m)) == bytes.fromhex(s)
    add(f"rfc8032_test{i + 1}", bytes.fromhex(pk), bytes.fromhex(m), bytes.fromhex(s), "rfc8032", "RFC 8032 section 7.1")
# RFC 8032 section 7.1 "TEST SHA(abc)" (round 4): the message is SHA-512("abc"), 64 bytes.  Transcribed from the published text;
# the transcription is self-checking — the seed must give the public key, the deterministic signer must reproduce the
# signature bit for bit and the verifier must accept it (asserted here), which no mistyped vector survives.  The fourth
# vector of that section (TEST 1024, a 1023-byte message) is not here: its message cannot be reproduced offline.
import hashlib
sk, pk, m, s = ("833fe62409237b9d62ec77587520911e9a759cec1d19755b7da901b96dca3d42", "ec172b93ad5e563bf4932c70e1245034c35467ef2efd4d64ebf819683467e2bf",
                hashlib.sha512(b"abc").hexdigest(),
                "dc2a4459e7369633a52b1bf277839a00201009a3efbf3ecb69bea2186c26b58909351fc9ac90b3ecfdfbc7c66431e0303dca179c138ac17ad9bef1177331a704")
assert ed.public_key(bytes.fromhex(sk)) == bytes.fromhex(pk) and ed.sign(bytes.fromhex(sk), bytes.fromhex(m)) == bytes.fromhex(s)
add("rfc8032_test_sha_abc", bytes.fromhex(pk), bytes.fromhex(m), bytes.fromhex(s), "rfc8032", "RFC 8032 section 7.1, TEST SHA(abc)")

# honest + flips
for i in range(16):
    seed = bytes(rng.getrandbits(8) for _ in range(32))
    pk = ed.public_key(seed)
    msg = b"sbv ed honest %d" % i
    sig = ed.sign(seed, msg)
    add(f"honest_{i}", pk, msg, sig, "honest")
    if i < 8:
        for part, lo, hi in [("R", 0, 256), ("S", 256, 512)]:
            bit = rng.randrange(lo, hi)
            s2 = bytearray(sig); s2[bit // 8] ^= 1 << (bit % 8)
            add(f"honest_{i}_flip_{part}_{bit}", pk, msg, bytes(s2), "bitflip")
        bit = rng.randrange(256)
        p2 = bytearray(pk); p2[bit // 8] ^= 1 << (bit % 8)
        add(f"honest_{i}_flip_pk_{bit}", bytes(p2), msg, sig, "bitflip")
        add(f"honest_{i}_other_msg", pk, msg + b"!", sig, "bitflip")

# S range
seed = bytes(rng.getrandbits(8) for _ in range(32)); pk = ed.public_key(seed); msg = b"s range"; sig = ed.sign(seed, msg)
S = int.from_bytes(sig[32:], "little")
add("s_plus_L", pk, msg, sig[:32] + (S + L).to_bytes(32, "little"), "s_range", "S + L is congruent but non-canonical")
add("s_eq_L", pk, msg, sig[:32] + L.to_bytes(32, "little"), "s_range")
add("s_eq_L_minus_1", pk, msg, sig[:32] + (L - 1).to_bytes(32, "little"), "s_range")
for b in (0x20, 0x40, 0x80):
    s2 = bytearray(sig); s2[63] |= b
    add(f"s_top_bit_{b:02x}", pk, msg, bytes(s2), "s_range", "sig[63] & 0xE0 != 0")
add("s_zero", pk, msg, sig[:32] + bytes(32), "s_range")
add("sig_all_zero", pk, msg, bytes(64), "s_range")
add("sig_all_ff", pk, msg, b"\xff" * 64, "s_range")

# public keys that are not points / special points
for y in range(2, 40):
    e = y.to_bytes(32, "little")
    if ed.decompress(e) is None:
        add(f"pk_not_on_curve_y{y}", e, msg, sig, "pubkey", "u/v is not a square")
        break
ident = enc_xy(0, 1)
neg_zero_ident = bytes(ident[:31]) + bytes([ident[31] | 0x80])
order2 = enc_xy(0, P - 1)
order4 = enc_xy(ed._recover_x(0, 0), 0)
small = {"identity": ident, "identity_negzero": neg_zero_ident, "order2": order2, "order2_negzero": order2[:31] + bytes([order2[31] | 0x80]),
         "order4": order4}
# an order-8 point: decompress candidates until 8*Pt = identity and 4*Pt != identity
for y in range(2, 2000):
    pt = ed.decompress(y.to_bytes(32, "little"))
    if pt is None:
        continue
    t8 = ed.pt_mul(L, pt)          # kills the prime-order component
    if ed.encode(ed.pt_mul(4, t8)) != ident:
        small["order8"] = ed.encode(t8)
        break
for nm, a_enc in small.items():
    a_pt = ed.decompress(a_enc)
    assert a_pt is not None, nm
    # valid signature under a small-order key: S random, R = [S]B - [k]A needs k = H(R, A, M) consistent
    done = False
    for ctr in range(400):
        s_ = rng.randrange(L)
        m_ = b"small order %s %d" % (nm.encode(), ctr)
        for t in range(8):
            cand = ed.encode(ed.pt_add(ed.pt_mul(s_, ed.B), ed.pt_neg(ed.pt_mul(t, a_pt))))
            if ed.verify(a_enc, m_, cand + s_.to_bytes(32, "little")):
                add(f"small_order_{nm}_valid", a_enc, m_, cand + s_.to_bytes(32, "little"), "small_order",
                    "cofactorless verification accepts small-order keys when the equation holds")
                done = True
                break
        if done:
            break
    add(f"small_order_{nm}_random_sig", a_enc, msg, sig, "small_order")
# non-canonical encodings of y (y + p < 2^255 needs y < 19)
for y in range(0, 19):
    pt = ed.decompress(y.to_bytes(32, "little"))
    if pt is None:
        continue
    nc = (y + P).to_bytes(32, "little")
    assert ed.decompress(nc) is not None
    for ctr in range(400):
        s_ = rng.randrange(L)
        m_ = b"noncanonical y%d %d" % (y, ctr)
        found = False
        for t in range(8):
            cand = ed.encode(ed.pt_add(ed.pt_mul(s_, ed.B), ed.pt_neg(ed.pt_mul(t, pt))))
            if ed.verify(nc, m_, cand + s_.to_bytes(32, "little")):
                add(f"pk_noncanonical_y{y}_valid", nc, m_, cand + s_.to_bytes(32, "little"), "noncanonical",
                    "Go accepts non-canonical encodings of valid points (y >= p)")
                found = True
                break
        if found:
            break
# non-canonical R: R' = identity but R_enc = 1 + p  -> byte compare fails
seed = bytes(rng.getrandbits(8) for _ in range(32)); pk = ed.public_key(seed); a, _ = ed.secret_expand(seed)
for nm, renc in [("canonical", ident), ("noncanonical", (1 + P).to_bytes(32, "little"))]:
    m_ = b"R is the identity"
    k = ed.hram(renc, pk, m_)
    s_ = k * a % L
    add(f"r_identity_{nm}", pk, m_, renc + s_.to_bytes(32, "little"), "noncanonical",
        "R' = identity; only the canonical encoding can match byte-wise")
# mixed-order key: A' = A + T8; accepted iff [k]T8 vanishes
t8 = ed.decompress(small["order8"])
A = ed.decompress(pk)
Am = ed.encode(ed.pt_add(A, t8))
got = {True: 0, False: 0}
for ctr in range(200):
    m_ = b"mixed order %d" % ctr
    r_ = rng.randrange(L)
    renc = ed.encode(ed.pt_mul(r_, ed.B))
    k = ed.hram(renc, Am, m_)
    s_ = (r_ + k * a) % L
    v = ed.verify(Am, m_, renc + s_.to_bytes(32, "little"))
    if got[v] < 2:
        add(f"mixed_order_key_{'ok' if v else 'bad'}_{got[v]}", Am, m_, renc + s_.to_bytes(32, "little"), "mixed_order",
            "A' = A + (order-8 point): equation holds only when 8 | k")
        got[v] += 1
    if got[True] >= 2 and got[False] >= 2:
        break

with open(os.path.join(HERE, "ed25519_vectors.json"), "w") as f:
    json.dump({"generator": "tests/golden/gen_ed25519_vectors.py", "seed": "0xED25519",
               "expected_from": "oracle/ed25519_py.py (Go crypto/ed25519.Verify rules)", "vectors": vectors}, f, indent=0)
by = {}
for v in vectors:
    by.setdefault(v["class"], [0, 0])[0 if v["accept"] else 1] += 1
print(len(vectors), "vectors")
for k, (a_, b_) in sorted(by.items()):
    print(f"  {k:14s} accept={a_:3d} reject={b_:3d}")
