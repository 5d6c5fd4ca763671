"""Batch signing on the GPU (sbv_p256_sign_batch) through the C-ABI: bit-identical to the host Signer
(RFC 6979) and to the RFC 6979 A.2.5 known answers; every signature verifies on the device and under the oracle."""
import ctypes
import hashlib
import time

import numpy as np
import pytest

import consensus_amd as sbv
import hostlib

pytestmark = pytest.mark.gpu
N_ORDER = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551


@pytest.fixture(scope="module")
def host():
    return hostlib.load()


@pytest.fixture(scope="module", autouse=True)
def _init():
    sbv.init(0)
    yield


def test_rfc6979_known_answers_on_the_gpu(rfc6979):
    d = bytes.fromhex(rfc6979["private_key"])
    kat = [s for s in rfc6979["signatures"] if s["hash_alg"] == "sha256"]
    sigs, ok = sbv.sign_batch(d, b"".join(bytes.fromhex(s["hash"]) for s in kat))
    assert ok == b"\x01\x01"
    for i, s in enumerate(kat):
        assert sigs[64 * i:64 * i + 64].hex() == s["r"] + s["s"]


def test_gpu_signatures_equal_the_host_signers_and_verify(host, oracle):
    rng = np.random.default_rng(20260921)
    nk, n = 37, 1 << 14
    keys = b"".join(int.to_bytes(int.from_bytes(rng.bytes(32), "big") % (N_ORDER - 1) + 1, 32, "big") for _ in range(nk))
    digests = rng.bytes(32 * (n - 2)) + b"\x00" * 32 + b"\xff" * 32
    index = rng.integers(0, nk, n).astype(np.uint32)
    sigs, ok = sbv.sign_batch(keys, digests, [int(x) for x in index])
    assert ok == b"\x01" * n
    pubs = []
    for k in range(nk):
        q = ctypes.create_string_buffer(64)
        assert host.sbvh_pubkey(keys[32 * k:32 * k + 32], q) == 0
        pubs.append(q.raw)
    for i in list(range(0, n, 257)) + [n - 2, n - 1]:                  # the host signer is slow: a stride of the batch
        rs = ctypes.create_string_buffer(64)
        assert host.sbvh_sign_rfc6979(keys[32 * int(index[i]):32 * int(index[i]) + 32], digests[32 * i:32 * i + 32], rs) == 0
        assert sigs[64 * i:64 * i + 64] == rs.raw, i
    tuples = b"".join(sigs[64 * i:64 * i + 64] + digests[32 * i:32 * i + 32] + pubs[int(index[i])] for i in range(n))
    bm = sbv.verify_batch(tuples, n)                                     # sign -> verify round trip on the device
    assert bm == b"\xff" * (n // 8)
    want = ctypes.create_string_buffer(n // 8)
    oracle.sbvo_p256_verify_batch(tuples[:160 * 2048], 2048, want, 8)    # and under the oracle
    assert want.raw[:256] == b"\xff" * 256


def test_bad_keys_and_indices(host):
    keys = b"\x00" * 32 + N_ORDER.to_bytes(32, "big") + (1).to_bytes(32, "big")
    sigs, ok = sbv.sign_batch(keys, hashlib.sha256(b"x").digest() * 4, [0, 1, 2, 3])
    assert ok == b"\x00\x00\x01\x00"
    assert sigs[:128] == b"\x00" * 128 and sigs[192:] == b"\x00" * 64
    rs = ctypes.create_string_buffer(64)
    assert host.sbvh_sign_rfc6979((1).to_bytes(32, "big"), hashlib.sha256(b"x").digest(), rs) == 0
    assert sigs[128:192] == rs.raw


def test_signing_rate_is_reported(capsys):
    """Not a pass/fail bar: prints the device signing rate at 2^18 signatures (device-resident buffers, event-free wall clock)."""
    import torch
    n, nk = 1 << 18, 1024
    rng = np.random.default_rng(7)
    keys = torch.from_numpy(np.frombuffer(b"".join(int.to_bytes(int.from_bytes(rng.bytes(32), "big") % (N_ORDER - 1) + 1, 32, "big")
                                                   for _ in range(nk)), dtype=np.uint8).copy()).cuda()
    dig = torch.from_numpy(np.frombuffer(rng.bytes(32 * n), dtype=np.uint8).copy()).cuda()
    sig = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    ok = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sbv.sign_batch_dev(keys.data_ptr(), nk, 0, dig.data_ptr(), n, sig.data_ptr(), ok.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert int(ok.sum().item()) == n
    with capsys.disabled():
        print("\n[sign] %d signatures in %.3f ms = %.1f M signatures/s" % (n, dt * 1e3, n / dt / 1e6))
