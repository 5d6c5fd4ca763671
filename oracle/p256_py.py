"""Python big-int twin of the P-256 ECDSA verification oracle.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (consensus_amd/, bench.py's
timed GPU leg) may import this module; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may use anything under oracle/.

Parity status: **parity unpinned by the reference** — SmartBFT-Go/consensus ships only
no-op / mock Verifiers (examples/naive_chain/node.go:86-96, test/test_app.go:231-248,
internal/bft/mocks/verifier_mock.go) and holds no golden vectors for this path.  The
arithmetic the north star names lives in Go's standard library (crypto/ecdsa,
crypto/internal/nistec, crypto/internal/bigmod, x/crypto/cryptobyte; toolchain-pinned by
go.mod:3 `go 1.20`), which is absent from /root/reference and cannot be executed here
(no Go toolchain).  This file restates the *published* algorithm:

  * SEC 1 v2.0 §4.1.4 (ECDSA verification) as implemented by Go >= 1.20
    crypto/ecdsa.verifyNISTEC: DER parse with cryptobyte strictness, 1 <= r,s <= N-1,
    public key coordinates non-negative, < p and on the curve, hash truncated to the
    leftmost 32 bytes and reduced mod N, R = u1*G + u2*Q with the exact group law,
    R = infinity rejected, accept iff R.x mod N == r.

and is pinned against (i) the RFC 6979 appendix A.2.5 known-answer signatures
(tests/golden/rfc6979_p256.json), (ii) OpenSSL 3.0 ECDSA_do_verify on every
mathematically-defined vector class (oracle/openssl_check.c).

All functions here are deliberately simple (affine arithmetic with modular inverses).
"""
from __future__ import annotations

import hashlib
from typing import Optional, Tuple

# --- domain parameters (SEC 2 secp256r1 / NIST P-256); SURVEY.md §8c lists them ------
P = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF
A = P - 3
B = 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B
GX = 0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296
GY = 0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5
N = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551

Point = Optional[Tuple[int, int]]  # None = point at infinity
G: Point = (GX, GY)


def on_curve(x: int, y: int) -> bool:
    return (y * y - (x * x * x + A * x + B)) % P == 0


def pt_neg(p: Point) -> Point:
    if p is None:
        return None
    return (p[0], (-p[1]) % P)


def pt_add(p: Point, q: Point) -> Point:
    """Exact affine group law (handles infinity, P == Q, P == -Q)."""
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1 + A) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def pt_mul(k: int, p: Point) -> Point:
    k %= N
    acc: Point = None
    add = p
    while k:
        if k & 1:
            acc = pt_add(acc, add)
        add = pt_add(add, add)
        k >>= 1
    return acc


# --- Go x/crypto/cryptobyte-strict DER parsing of an ECDSA-Sig-Value ------------------

def _read_asn1(buf: bytes, want_tag: int):
    """cryptobyte.String.ReadASN1: returns (contents, rest) or None.

    Mirrors readASN1: single-byte tags only, definite lengths, DER-minimal long form,
    at most 4 length bytes.
    """
    if len(buf) < 2:
        return None
    tag, len_byte = buf[0], buf[1]
    if tag & 0x1F == 0x1F:
        return None  # high-tag-number form unsupported
    if len_byte & 0x80 == 0:
        length = len_byte
        header = 2
    else:
        len_len = len_byte & 0x7F
        if len_len == 0 or len_len > 4:
            return None  # indefinite length or too long
        if len(buf) < 2 + len_len:
            return None
        length = int.from_bytes(buf[2:2 + len_len], "big")
        if length < 128:
            return None  # should have used short form
        if length >> ((len_len - 1) * 8) == 0:
            return None  # leading zero length octet
        header = 2 + len_len
    if len(buf) < header + length:
        return None
    if tag != want_tag:
        return None
    return buf[header:header + length], buf[header + length:]


def _read_asn1_uint_bytes(buf: bytes):
    """cryptobyte ReadASN1Integer(*[]byte): minimal, non-negative, zero-stripped."""
    got = _read_asn1(buf, 0x02)
    if got is None:
        return None
    body, rest = got
    if len(body) == 0:
        return None
    if len(body) > 1:
        if body[0] == 0x00 and body[1] & 0x80 == 0:
            return None  # non-minimal positive
        if body[0] == 0xFF and body[1] & 0x80 == 0x80:
            return None  # non-minimal negative
    if body[0] & 0x80:
        return None  # negative
    while len(body) > 1 and body[0] == 0:
        body = body[1:]
    return body, rest


def parse_der_sig(sig: bytes):
    """crypto/ecdsa.parseSignature: -> (r_bytes, s_bytes) or None."""
    got = _read_asn1(sig, 0x30)
    if got is None:
        return None
    inner, rest = got
    if rest:
        return None
    g = _read_asn1_uint_bytes(inner)
    if g is None:
        return None
    r, inner = g
    g = _read_asn1_uint_bytes(inner)
    if g is None:
        return None
    s, inner = g
    if inner:
        return None
    return r, s


def der_encode_sig(r: int, s: int) -> bytes:
    """What Go's SignASN1 / encodeSignature emits (minimal DER)."""
    def enc_int(v: int) -> bytes:
        b = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")
        if b[0] & 0x80:
            b = b"\x00" + b
        return b"\x02" + _der_len(len(b)) + b
    body = enc_int(r) + enc_int(s)
    return b"\x30" + _der_len(len(body)) + body


def _der_len(n: int) -> bytes:
    if n < 128:
        return bytes([n])
    b = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([0x80 | len(b)]) + b


# --- verification ---------------------------------------------------------------------

def hash_to_int(h: bytes) -> int:
    """crypto/ecdsa.hashToNat for P-256: leftmost 32 bytes, big-endian (then mod N)."""
    if len(h) >= 32:
        h = h[:32]
    return int.from_bytes(h, "big")


def verify_raw(r: int, s: int, e_bytes: bytes, qx: int, qy: int) -> bool:
    """Accept/reject for already-parsed integers; e_bytes is the (untruncated) hash."""
    if qx < 0 or qy < 0 or qx >= P or qy >= P:
        return False
    if not on_curve(qx, qy):
        return False
    if not (1 <= r < N and 1 <= s < N):
        return False
    e = hash_to_int(e_bytes) % N
    w = pow(s, -1, N)
    u1 = e * w % N
    u2 = r * w % N
    R = pt_add(pt_mul(u1, G), pt_mul(u2, (qx, qy)))
    if R is None:
        return False
    return R[0] % N == r


def verify_tuple(t: bytes) -> bool:
    """160-byte ABI tuple r|s|hash|Qx|Qy (5 x 32 B big-endian) -> accept?"""
    assert len(t) == 160
    f = [int.from_bytes(t[i * 32:(i + 1) * 32], "big") for i in range(5)]
    return verify_raw(f[0], f[1], t[64:96], f[3], f[4])


def verify_asn1(qx: int, qy: int, h: bytes, sig: bytes) -> bool:
    """crypto/ecdsa.VerifyASN1 semantics."""
    rs = parse_der_sig(sig)
    if rs is None:
        return False
    rb, sb = rs
    if len(rb) > 32 or len(sb) > 32:
        return False  # bigmod setBytes: overflows the modulus size
    return verify_raw(int.from_bytes(rb, "big"), int.from_bytes(sb, "big"), h, qx, qy)


def sign(d: int, k: int, e_bytes: bytes) -> Tuple[int, int]:
    """Textbook ECDSA signing with caller-supplied nonce (synthetic-data generator)."""
    e = hash_to_int(e_bytes) % N
    R = pt_mul(k, G)
    assert R is not None
    r = R[0] % N
    s = pow(k, -1, N) * (e + r * d) % N
    assert r != 0 and s != 0
    return r, s


def pack_tuple(r: int, s: int, h32: bytes, qx: int, qy: int) -> bytes:
    assert len(h32) == 32
    return (r.to_bytes(32, "big") + s.to_bytes(32, "big") + h32 +
            qx.to_bytes(32, "big") + qy.to_bytes(32, "big"))


def sha256(b: bytes) -> bytes:
    return hashlib.sha256(b).digest()
