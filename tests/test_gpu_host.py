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
