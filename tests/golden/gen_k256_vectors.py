#!/usr/bin/env python3
"""Generate tests/golden/k256_vectors.json: raw-ABI secp256k1 tuples (r | s | hash | Qx | Qy) with the verdict of the
Python big-int twin (oracle/k256_py.py).  The reference holds no vectors for this path and no signature
arithmetic at all, so the build commits its own, seeded and reproducible; the tests require the C oracle, OpenSSL
(NID_secp256k1), the emulated device algorithm and the HIP path to agree with them bit for bit.

Run from the repo root:  python tests/golden/gen_k256_vectors.py
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import k256_py as ec  # noqa: E402

rng = random.Random(0x6B323536)
P, N, G = ec.P, ec.N, ec.G
vectors = []


def add(name, r, s, hbytes, qx, qy, cls, note=""):
    assert len(hbytes) == 32
    t = r.to_bytes(32, "big") + s.to_bytes(32, "big") + hbytes + qx.to_bytes(32, "big") + qy.to_bytes(32, "big")
    vectors.append({"name": name, "class": cls, "tuple": t.hex(), "accept": ec.verify_tuple(t), "note": note})


def scalar():
    return rng.randrange(1, N)


def keypair():
    d = scalar()
    return d, ec.pt_mul(d, G)


def lift_x(x):
    v = (x * x * x + 7) % P
    y = pow(v, (P + 1) // 4, P)
    return (x, y) if y * y % P == v else None


def sha(b):
    return hashlib.sha256(b).digest()


# 1. honest signatures, high-S twins, single-bit corruptions of every field
for i in range(16):
    d, Q = keypair()
    hb = sha(b"k256 honest %d" % i)
    r, s = ec.sign(d, scalar(), hb)
    add(f"honest_{i}", r, s, hb, Q[0], Q[1], "honest")
    add(f"honest_{i}_high_s", r, N - s, hb, Q[0], Q[1], "high_s", "no low-S rule at this layer")
    if i < 8:
        t = bytearray(ec.make_tuple(r, s, hb, Q))
        for fld, fname in enumerate(["r", "s", "hash", "qx", "qy"]):
            bit = rng.randrange(256)
            t2 = bytearray(t)
            t2[fld * 32 + bit // 8] ^= 1 << (bit % 8)
            f = [int.from_bytes(t2[k * 32:(k + 1) * 32], "big") for k in range(5)]
            add(f"honest_{i}_flip_{fname}_{bit}", f[0], f[1], bytes(t2[64:96]), f[3], f[4], "bitflip")

# 2. range edges
d, Q = keypair()
hb = sha(b"k256 range edges")
r, s = ec.sign(d, scalar(), hb)
for name, rr, ss in [("r_zero", 0, s), ("s_zero", r, 0), ("r_eq_N", N, s), ("s_eq_N", r, N), ("r_max", 2**256 - 1, s),
                     ("s_max", r, 2**256 - 1), ("r_one_s_one", 1, 1), ("r_plus_N_wraps", (r + N) % 2**256, s)]:
    add(name, rr, ss, hb, Q[0], Q[1], "range")
k = scalar()
r1 = ec.pt_mul(k, G)[0] % N
for name, sv in [("s_eq_N_minus_1_valid", N - 1), ("s_eq_1_valid", 1)]:
    e1 = (sv * k - r1 * d) % N
    add(name, r1, sv, e1.to_bytes(32, "big"), Q[0], Q[1], "range")

# 3. hash edges (hashToNat: one conditional subtraction of N)
d, Q = keypair()
for name, hv in [("e_zero", 0), ("e_eq_N", N), ("e_gt_N", N + 977), ("e_max", 2**256 - 1), ("e_N_minus_1", N - 1)]:
    hb = hv.to_bytes(32, "big")
    r, s = ec.sign(d, scalar(), hb)
    add(f"{name}_valid", r, s, hb, Q[0], Q[1], "hash_edge")
    add(f"{name}_wrong_key", r, s, hb, G[0], G[1], "hash_edge")

# 4. public-key edges
d, Q = keypair()
hb = sha(b"k256 key edges")
r, s = ec.sign(d, scalar(), hb)
add("q_off_curve_y_plus_1", r, s, hb, Q[0], (Q[1] + 1) % P, "key")
add("q_x_eq_p", r, s, hb, P, Q[1], "key")
add("q_y_eq_p", r, s, hb, Q[0], P, "key")
add("q_x_plus_p", r, s, hb, Q[0] + P if Q[0] + P < 2**256 else Q[0], Q[1], "key", "x + p is congruent but out of range (when it fits)")
add("q_zero_zero", r, s, hb, 0, 0, "key", "(0, 0) is not on y^2 = x^3 + 7")
add("q_neg", r, s, hb, Q[0], P - Q[1], "key", "-Q: on the curve, wrong key")
# x = 0 has no point (7 is a non-residue?) -- take the smallest x that does and sign with a forged triple
x = 1
while lift_x(x) is None:
    x += 1
Qs = lift_x(x)
u1, u2 = scalar(), scalar()
R = ec.pt_add(ec.pt_mul(u1, G), ec.pt_mul(u2, Qs))
rr = R[0] % N
ss = rr * pow(u2, -1, N) % N
add("q_smallest_x_forged_valid", rr, ss, (u1 * ss % N).to_bytes(32, "big"), Qs[0], Qs[1], "key")
for name, Qk in [("q_is_G", G), ("q_is_minus_G", ec.pt_neg(G))]:
    u1, u2 = scalar(), scalar()
    R = ec.pt_add(ec.pt_mul(u1, G), ec.pt_mul(u2, Qk))
    rr = R[0] % N
    ss = rr * pow(u2, -1, N) % N
    add(name + "_valid", rr, ss, (u1 * ss % N).to_bytes(32, "big"), Qk[0], Qk[1], "key")

# 5. the exceptional cases of the group law inside u1 * G + u2 * Q
#    u1 * G == u2 * Q (the last addition is a doubling): Q = (u1 / u2) * G
for i in range(3):
    u1, u2 = scalar(), scalar()
    Qd = ec.pt_mul(u1 * pow(u2, -1, N) % N, G)
    R = ec.pt_add(ec.pt_mul(u1, G), ec.pt_mul(u2, Qd))
    rr = R[0] % N
    ss = rr * pow(u2, -1, N) % N
    add(f"u1G_eq_u2Q_{i}", rr, ss, (u1 * ss % N).to_bytes(32, "big"), Qd[0], Qd[1], "group_law", "final addition doubles")
#    u1 * G == -u2 * Q (the sum is infinity -> reject): any r, s with e = -r * d
for i in range(3):
    d, Q = keypair()
    rr, ss = scalar(), scalar()
    e = (-rr * d) % N
    add(f"u1G_eq_minus_u2Q_{i}", rr, ss, e.to_bytes(32, "big"), Q[0], Q[1], "group_law", "R = infinity")
#    small scalars: u2 = 1 .. 9 and u1 = 0 (e = 0) walk the window table and the skip paths
d, Q = keypair()
for u2 in [1, 2, 7, 8, 9, 16, 2**128, N - 1]:
    R = ec.pt_mul(u2, Q)
    rr = R[0] % N
    ss = rr * pow(u2, -1, N) % N
    add(f"u1_zero_u2_{u2 if u2 < 100 else hex(u2)[:10]}", rr, ss, (0).to_bytes(32, "big"), Q[0], Q[1], "group_law", "e = 0: u1 = 0")

# 6. R.x in [N, p): r = R.x - N (p - N is 129 bits wide, so this has to be constructed)
found = 0
x = N + rng.randrange(1, P - N)
while found < 3:
    Rb = lift_x(x)
    x += 1
    if Rb is None:
        continue
    u1, u2 = scalar(), scalar()
    Qf = ec.pt_mul(pow(u2, -1, N), ec.pt_add(Rb, ec.pt_neg(ec.pt_mul(u1, G))))
    rr = Rb[0] - N
    ss = rr * pow(u2, -1, N) % N
    add(f"rx_ge_N_{found}", rr, ss, (u1 * ss % N).to_bytes(32, "big"), Qf[0], Qf[1], "rx_wrap", "R.x in [N, p): R.x mod N = R.x - N")
    add(f"rx_ge_N_{found}_r_not_reduced", Rb[0] % 2**256, ss, (u1 * ss % N).to_bytes(32, "big"), Qf[0], Qf[1], "rx_wrap", "r = R.x >= N is out of range")
    found += 1

out = os.path.join(HERE, "k256_vectors.json")
with open(out, "w") as f:
    json.dump({"curve": "secp256k1", "generator": "tests/golden/gen_k256_vectors.py", "vectors": vectors}, f, indent=0)
print(len(vectors), "vectors,", sum(v["accept"] for v in vectors), "accepted ->", out)
