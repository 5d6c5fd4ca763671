"""Python big-int twin of the secp256k1 ECDSA verification oracle (oracle/secp256k1_oracle.c).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (consensus_amd/, bench.py's timed GPU leg) may import this
module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.

Parity status: **parity unpinned by the reference** (see secp256k1_oracle.c's header: the reference holds no signature
arithmetic, and this curve is not in Go's standard library).  Restates SEC 1 v2.0 §4.1.4 with the SEC 2 v2.0 §2.4.1
parameters and the input rules of the P-256 twin (oracle/p256_py.py); pinned against the community RFC 6979 secp256k1
known answers (tests/golden/rfc6979_k256.json) and OpenSSL's NID_secp256k1.

Deliberately simple: affine arithmetic with modular inverses, one function per step.
"""
from __future__ import annotations

from typing import Optional, Tuple

P = 2**256 - 2**32 - 977
B = 7
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141

Point = Optional[Tuple[int, int]]  # None = point at infinity
G: Point = (GX, GY)


def on_curve(x: int, y: int) -> bool:
    return (y * y - (x * x * x + B)) % P == 0


def pt_neg(p: Point) -> Point:
    return None if p is None else (p[0], (-p[1]) % P)


def pt_add(p: Point, q: Point) -> Point:
    """Exact affine group law (infinity, P == Q, P == -Q)."""
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P          # a = 0
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def pt_mul(k: int, p: Point) -> Point:
    acc: Point = None
    for bit in bin(k)[2:] if k else "":
        acc = pt_add(acc, acc)
        if bit == "1":
            acc = pt_add(acc, p)
    return acc


def verify_raw(r: int, s: int, e_bytes: bytes, qx: int, qy: int) -> bool:
    """r, s, qx, qy: integers as decoded from the 32-byte big-endian fields; e_bytes: the hash as given (any length)."""
    if not (1 <= r < N and 1 <= s < N):
        return False
    if not (0 <= qx < P and 0 <= qy < P and on_curve(qx, qy)):
        return False
    e = int.from_bytes(e_bytes[:32], "big") % N
    w = pow(s, -1, N)
    u1, u2 = e * w % N, r * w % N
    R = pt_add(pt_mul(u1, G), pt_mul(u2, (qx, qy)))
    if R is None:
        return False
    return R[0] % N == r


def verify_tuple(t: bytes) -> bool:
    assert len(t) == 160
    f = [int.from_bytes(t[32 * i:32 * i + 32], "big") for i in range(5)]
    return verify_raw(f[0], f[1], t[64:96], f[3], f[4])


def sign(d: int, k: int, h32: bytes) -> Tuple[int, int]:
    """Textbook ECDSA with an explicit nonce (test data only)."""
    e = int.from_bytes(h32[:32], "big") % N
    R = pt_mul(k, G)
    assert R is not None
    r = R[0] % N
    s = pow(k, -1, N) * (e + r * d) % N
    assert r and s
    return r, s


def make_tuple(r: int, s: int, h32: bytes, q: Point) -> bytes:
    assert q is not None
    return r.to_bytes(32, "big") + s.to_bytes(32, "big") + h32 + q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big")
