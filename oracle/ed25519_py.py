"""Python big-int twin of the Ed25519 verification oracle (Go crypto/ed25519.Verify semantics).

TEST INFRASTRUCTURE ONLY — see oracle/p256_py.py for the rules on what may import oracle/.

Parity status: unpinned by the reference (SmartBFT ships no signature code).
Restates Go >= 1.20 crypto/ed25519.verify + filippo.io/edwards25519 (vendored in std as
crypto/internal/edwards25519):

  * len(sig) != 64 or sig[63] & 0xE0 != 0              -> false
  * A = Point.SetBytes(pk): y = low 255 bits, NON-canonical y (>= p) accepted (reduced mod p);
    x = sqrt((y^2-1)/(d y^2+1)), not a square -> false; sign bit selects -x, and "x = 0 with
    sign bit 1" is accepted (no RFC 8032 §5.1.3 step 4 check)
  * k = SHA-512(R_enc || pk || msg) reduced mod L (SetUniformBytes)
  * S = sig[32:] must be canonical (< L) else false (SetCanonicalBytes)
  * R' = [S]B + [k](-A)  (cofactorless, no small-order rejection)
  * accept iff encode(R') == sig[:32] BYTE-WISE (so a non-canonical R encoding is rejected)

Pinned against RFC 8032 §7.1 test vectors 1-3 and OpenSSL EVP Ed25519 on honest / bit-flipped
signatures (oracle/openssl_check.c), where both implementations agree by construction.
"""
from __future__ import annotations

import hashlib
from typing import Optional, Tuple

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, -1, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)

Point = Tuple[int, int, int, int]  # extended (X, Y, Z, T), x = X/Z, y = Y/Z, xy = T/Z


def _recover_x(y: int, sign: int) -> Optional[int]:
    """field.SqrtRatio(u, v) + sign selection exactly as edwards25519.Point.SetBytes does."""
    u = (y * y - 1) % P
    v = (D * y * y + 1) % P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u) % P * SQRT_M1 % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    if r & 1:                      # Absolute(): the non-negative (even) root
        r = (-r) % P
    if not (correct or flipped):
        return None
    if sign:
        r = (-r) % P               # note: -0 = 0, "negative zero" is accepted
    return r


def decompress(b: bytes) -> Optional[Point]:
    if len(b) != 32:
        return None
    y = int.from_bytes(b, "little") & (2**255 - 1)
    sign = b[31] >> 7
    y %= P                          # non-canonical encodings accepted
    x = _recover_x(y, sign)
    if x is None:
        return None
    return (x, y, 1, x * y % P)


def pt_add(p: Point, q: Point) -> Point:
    x1, y1, z1, t1 = p
    x2, y2, z2, t2 = q
    a = (y1 - x1) * (y2 - x2) % P
    b = (y1 + x1) * (y2 + x2) % P
    c = 2 * D * t1 % P * t2 % P
    dd = 2 * z1 * z2 % P
    e, f, g, h = (b - a) % P, (dd - c) % P, (dd + c) % P, (b + a) % P
    return (e * f % P, g * h % P, f * g % P, e * h % P)


def pt_neg(p: Point) -> Point:
    return ((-p[0]) % P, p[1], p[2], (-p[3]) % P)


IDENT: Point = (0, 1, 1, 0)
_by = 4 * pow(5, -1, P) % P
_bx = _recover_x(_by, 0)
B: Point = (_bx, _by, 1, _bx * _by % P)


def pt_mul(k: int, p: Point) -> Point:
    acc = IDENT
    while k:
        if k & 1:
            acc = pt_add(acc, p)
        p = pt_add(p, p)
        k >>= 1
    return acc


def encode(p: Point) -> bytes:
    zi = pow(p[2], -1, P)
    x, y = p[0] * zi % P, p[1] * zi % P
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def hram(r_enc: bytes, pk: bytes, msg: bytes) -> int:
    return int.from_bytes(hashlib.sha512(r_enc + pk + msg).digest(), "little") % L


def verify_k(pk: bytes, sig: bytes, k: int) -> bool:
    """Verification with the SHA-512 scalar k supplied (the 128-byte ABI tuple: sig | pk | k)."""
    if len(sig) != 64 or sig[63] & 0xE0:
        return False
    a = decompress(pk)
    if a is None:
        return False
    s = int.from_bytes(sig[32:], "little")
    if s >= L or k >= L:            # k comes out of a mod-L reduction; anything else is not a valid tuple
        return False
    r = pt_add(pt_mul(s, B), pt_mul(k, pt_neg(a)))
    return encode(r) == sig[:32]


def verify(pk: bytes, msg: bytes, sig: bytes) -> bool:
    if len(pk) != 32:
        raise ValueError("ed25519: bad public key length")   # Go panics
    if len(sig) != 64:
        return False
    return verify_k(pk, sig, hram(sig[:32], pk, msg))


def verify_tuple(t: bytes) -> bool:
    """128-byte ABI tuple: sig(64) | pk(32) | k(32, little-endian, reduced mod L; k >= L rejects)."""
    assert len(t) == 128
    return verify_k(t[64:96], t[:64], int.from_bytes(t[96:128], "little"))


def secret_expand(seed: bytes):
    h = hashlib.sha512(seed).digest()
    a = int.from_bytes(h[:32], "little")
    a &= (1 << 254) - 8
    a |= 1 << 254
    return a, h[32:]


def public_key(seed: bytes) -> bytes:
    a, _ = secret_expand(seed)
    return encode(pt_mul(a, B))


def sign(seed: bytes, msg: bytes) -> bytes:
    a, prefix = secret_expand(seed)
    pk = encode(pt_mul(a, B))
    r = int.from_bytes(hashlib.sha512(prefix + msg).digest(), "little") % L
    r_enc = encode(pt_mul(r, B))
    k = hram(r_enc, pk, msg)
    s = (r + k * a) % L
    return r_enc + s.to_bytes(32, "little")


def pack_tuple(pk: bytes, msg: bytes, sig: bytes) -> bytes:
    sig = (sig + bytes(64))[:64]
    return sig + pk + hram(sig[:32], pk, msg).to_bytes(32, "little")
