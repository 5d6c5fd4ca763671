"""GPU tier for the host side: the C++ api.Verifier mirror over the real backend (libsbv.so)."""
import ctypes
import hashlib

import pytest

import hostlib
from hostlib import INVALID, OK

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return hostlib.load()


def _new(lib, wait_us=200, cache=0):
    cb = hostlib.BACKEND_FN(lambda *a: -1)
    return lib.sbvh_verifier_new(0, 0, cb, None, 4096, wait_us, cache)


def test_sign_then_verify_through_the_gpu(lib):
    v = _new(lib)
    try:
        s = lib.sbvh_signer_new(7, hashlib.sha256(b"gpu-node").digest())
        q = ctypes.create_string_buffer(64)
        lib.sbvh_signer_public_key(s, q)
        lib.sbvh_register_consenter(v, 7, q.raw)
        out = ctypes.create_string_buffer(80)
        msg = b"raw view data"
        n = lib.sbvh_sign(s, msg, len(msg), out, 80)
        assert lib.sbvh_verify_signature(v, 7, out.raw[:n], n, msg, len(msg)) == OK
        assert lib.sbvh_verify_signature(v, 7, out.raw[:n], n, msg + b"!", len(msg) + 1) == INVALID
        assert lib.sbvh_verify_signature(v, 8, out.raw[:n], n, msg, len(msg)) == INVALID
    finally:
        lib.sbvh_verifier_free(v)


def test_replay_config1_shape_on_gpu(lib):
    """4 nodes, K = 100 (DefaultConfig RequestBatchMaxCount, pkg/types/config.go:94), 3 sequences + 50 decisions."""
    v = _new(lib)
    try:
        res = hostlib.ReplayResult()
        assert lib.sbvh_replay(v, 4, 100, 3, 50, 16, ctypes.byref(res)) == 0
        assert res.status == 0 and res.proposals_with_quorum == 50 and res.batch_tuples == 150
        assert res.max_backend_batch >= 100
    finally:
        lib.sbvh_verifier_free(v)


def test_ed25519_scheme_through_the_gpu(lib):
    """BASELINE.json configs[4] at the seam: Scheme::ED25519 Verifier + Signer over sbv_ed25519_verify_batch — a
    100-request proposal in one device batch, commit votes, tampered and wrong-length signatures."""
    cb = hostlib.BACKEND_FN(lambda *a: -1)
    v = lib.sbvh_verifier_new_scheme(1, 0, 0, cb, None, 4096, 200, 0)
    try:
        node = lib.sbvh_signer_new_scheme(1, 3, hashlib.sha256(b"ed-node").digest())
        q = ctypes.create_string_buffer(64)
        lib.sbvh_signer_public_key(node, q)
        lib.sbvh_register_consenter(v, 3, q.raw)
        out = ctypes.create_string_buffer(80)
        msg = b"raw view data"
        n = lib.sbvh_sign(node, msg, len(msg), out, 80)
        assert n == 64
        assert lib.sbvh_verify_signature(v, 3, out.raw[:n], n, msg, len(msg)) == OK
        assert lib.sbvh_verify_signature(v, 3, out.raw[:n], n, msg + b"!", len(msg) + 1) == INVALID
        assert lib.sbvh_verify_signature(v, 3, out.raw[:63], 63, msg, len(msg)) == INVALID
        clients = {}
        for i in range(5):
            c = lib.sbvh_signer_new_scheme(1, 0, hashlib.sha256(b"ed-client%d" % i).digest())
            lib.sbvh_signer_public_key(c, q)
            lib.sbvh_register_client(v, b"carol%d" % i, q.raw)
            clients[i] = c

        def request(i, corrupt=False):
            u = hostlib.request_unsigned("carol%d" % (i % 5), "r%d" % i, bytes([i % 251]))
            k = lib.sbvh_sign(clients[i % 5], u, len(u), out, 80)
            sig = bytearray(out.raw[:k])
            if corrupt:
                sig[5] ^= 0x40
            return hostlib.request_encode(u, bytes(sig))

        reqs = [request(i) for i in range(100)]
        info = ctypes.create_string_buffer(1 << 16)
        ln, cnt = ctypes.c_size_t(), ctypes.c_size_t()
        p = hostlib.payload_encode(reqs)
        assert lib.sbvh_verify_proposal(v, p, len(p), b"h", 1, b"m", 1, 0, info, 1 << 16, ctypes.byref(ln), ctypes.byref(cnt)) == OK
        assert cnt.value == 100
        reqs[77] = request(77, corrupt=True)
        p = hostlib.payload_encode(reqs)
        assert lib.sbvh_verify_proposal(v, p, len(p), b"h", 1, b"m", 1, 0, info, 1 << 16, ctypes.byref(ln), ctypes.byref(cnt)) == INVALID
    finally:
        lib.sbvh_verifier_free(v)
