"""Batch signing (consensus_amd/csrc/p256_sign.h), CPU tier: the per-lane device source compiled for the
host (tests/emul), pinned on the RFC 6979 A.2.5 known answers and diffed against the host Signer (consensus_amd/host,
sign_rfc6979) and the oracle's verifier on random keys and digests."""
import ctypes
import hashlib

import numpy as np
import pytest

import hostlib
from test_emul_device_algo import emul  # noqa: F401  (fixture: builds tests/emul/libsbv_emul.so)

N_ORDER = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551


@pytest.fixture(scope="module")
def host():
    return hostlib.load()


def sign(emul, keys: bytes, digests: bytes, index=None):
    n, nk = len(digests) // 32, len(keys) // 32
    sigs, ok = ctypes.create_string_buffer(64 * n), ctypes.create_string_buffer(n)
    idx = None if index is None else (ctypes.c_uint32 * n)(*index)
    emul.sbve_p256_sign_batch.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t,
                                          ctypes.c_char_p, ctypes.c_char_p]
    emul.sbve_p256_sign_batch.restype = None
    emul.sbve_p256_sign_batch(keys, nk, idx, digests, n, sigs, ok)
    return sigs.raw, ok.raw


def test_rfc6979_known_answers(emul, rfc6979):
    d = bytes.fromhex(rfc6979["private_key"])
    kat = [s for s in rfc6979["signatures"] if s["hash_alg"] == "sha256"]
    assert len(kat) == 2
    sigs, ok = sign(emul, d, b"".join(bytes.fromhex(s["hash"]) for s in kat))
    assert ok == b"\x01\x01"
    for i, s in enumerate(kat):
        assert sigs[64 * i:64 * i + 64].hex() == s["r"] + s["s"], s["message"]


def test_matches_the_host_signer_and_verifies_under_the_oracle(emul, host, oracle):
    rng = np.random.default_rng(6979)
    nk, n = 5, 48
    keys = b"".join(int.to_bytes(int.from_bytes(rng.bytes(32), "big") % (N_ORDER - 1) + 1, 32, "big") for _ in range(nk))
    digests = b"".join(hashlib.sha256(b"msg %d" % i).digest() for i in range(n - 3))
    digests += b"\x00" * 32 + b"\xff" * 32 + N_ORDER.to_bytes(32, "big")            # e = 0, e >= N (reduced once), e = N
    index = [int(x) for x in rng.integers(0, nk, n)]
    sigs, ok = sign(emul, keys, digests, index)
    assert ok == b"\x01" * n
    for i in range(n):
        d = keys[32 * index[i]:32 * index[i] + 32]
        rs = ctypes.create_string_buffer(64)
        assert host.sbvh_sign_rfc6979(d, digests[32 * i:32 * i + 32], rs) == 0
        assert sigs[64 * i:64 * i + 64] == rs.raw, i
        q = ctypes.create_string_buffer(64)
        assert host.sbvh_pubkey(d, q) == 0
        assert oracle.sbvo_p256_verify_tuple(sigs[64 * i:64 * i + 64] + digests[32 * i:32 * i + 32] + q.raw) == 1
    # default key choice: i % n_keys
    sigs2, ok2 = sign(emul, keys, digests[:32 * 7])
    for i in range(7):
        rs = ctypes.create_string_buffer(64)
        host.sbvh_sign_rfc6979(keys[32 * (i % nk):32 * (i % nk) + 32], digests[32 * i:32 * i + 32], rs)
        assert sigs2[64 * i:64 * i + 64] == rs.raw


def test_keys_outside_the_scalar_range_and_bad_indices_produce_nothing(emul):
    keys = b"\x00" * 32 + N_ORDER.to_bytes(32, "big") + b"\xff" * 32 + (1).to_bytes(32, "big") + (N_ORDER - 1).to_bytes(32, "big")
    digests = hashlib.sha256(b"x").digest() * 6
    sigs, ok = sign(emul, keys, digests, [0, 1, 2, 3, 4, 5])
    assert ok == b"\x00\x00\x00\x01\x01\x00"                                           # index 5 does not exist
    for i in (0, 1, 2, 5):
        assert sigs[64 * i:64 * i + 64] == b"\x00" * 64
    assert sigs[64 * 3:64 * 4] != b"\x00" * 64 and sigs[64 * 4:64 * 5] != b"\x00" * 64
