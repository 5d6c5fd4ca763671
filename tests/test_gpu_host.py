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
        import consensus_amd as sbv
        before = sbv.wide_key_stats()[0]
        lib.sbvh_register_consenter(v, 7, q.raw)
        # RegisterConsenter hands the key to the device AND widens it (sbv_p256_widen_keys): consenters sign every vote of the epoch
        assert sbv.wide_key_stats()[0] in (before, before + 1) and sbv.wide_key_stats()[0] >= 1
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


class _Seam:
    """Verifier over the real backend + 3 registered clients, for the leader / pool call sites."""

    def __init__(self, lib):
        self.lib = lib
        self.v = _new(lib)
        self.clients = {}
        q = ctypes.create_string_buffer(64)
        for i in range(3):
            s = lib.sbvh_signer_new(0, hashlib.sha256(b"gpu-client%d" % i).digest())
            lib.sbvh_signer_public_key(s, q)
            lib.sbvh_register_client(self.v, b"bob%d" % i, q.raw)
            self.clients["bob%d" % i] = s

    def request(self, client, rid, corrupt=False, payload=b"tx"):
        u = hostlib.request_unsigned(client, rid, payload)
        out = ctypes.create_string_buffer(80)
        k = self.lib.sbvh_sign(self.clients[client], u, len(u), out, 80)
        sig = bytearray(out.raw[:k])
        if corrupt:
            sig[-1] ^= 1
        return hostlib.request_encode(u, bytes(sig))

    def verify_request(self, raw):
        out = ctypes.create_string_buffer(1024)
        n = ctypes.c_size_t()
        st = self.lib.sbvh_verify_request(self.v, raw, len(raw), out, 1024, ctypes.byref(n))
        return st, (hostlib.split_infos(out.raw[:n.value]) or [None])[0]


def test_leader_verify_request_on_gpu(lib):
    """Controller.HandleRequest -> VerifyRequest (internal/bft/controller.go:233-246): one signature per call on an
    arbitrary goroutine; an error means the request is not pooled (controller_test.go:548)."""
    sx = _Seam(lib)
    try:
        good = sx.request("bob0", "req-1")
        assert sx.verify_request(good) == (OK, ("bob0", "req-1"))
        assert sx.verify_request(sx.request("bob1", "req-2", corrupt=True))[0] == INVALID
        stranger = hostlib.request_encode(hostlib.request_unsigned("mallory", "x", b""), b"\x30\x00")
        assert sx.verify_request(stranger)[0] == INVALID
        assert sx.verify_request(good[:-3])[0] == INVALID
        assert sx.verify_request(b"")[0] == INVALID
    finally:
        lib.sbvh_verifier_free(sx.v)


def test_pool_prune_pattern_on_gpu(lib):
    """Controller.MaybePruneRevokedRequests -> Pool.Prune (controller.go:733-746, requestpool.go:335-371): every pooled request
    (RequestPoolSize = 400, pkg/types/config.go:98) is re-verified one call at a time; exactly the rejects leave the pool
    (requestpool_test.go:264)."""
    sx = _Seam(lib)
    try:
        pool = [sx.request("bob%d" % (i % 3), "r%d" % i, corrupt=(i % 7 == 3)) for i in range(400)]
        kept = [r for r in pool if sx.verify_request(r)[0] == OK]
        assert kept == [r for i, r in enumerate(pool) if i % 7 != 3]
    finally:
        lib.sbvh_verifier_free(sx.v)


def test_config2_verify_proposal_k_10000_on_gpu(lib):
    """BASELINE.json configs[2] at the seam: 4 nodes, one K = 10 000-request proposal per sequence through VerifyProposal
    (view.go:553-559) as ONE device batch, then the commit path; the replay driver signs the traffic itself."""
    v = _new(lib)
    try:
        res = hostlib.ReplayResult()
        assert lib.sbvh_replay(v, 4, 10000, 2, 0, 16, ctypes.byref(res)) == 0
        assert res.status == 0
        assert res.max_backend_batch >= 10000
    finally:
        lib.sbvh_verifier_free(v)


def test_config3_decision_replay_50000_x_11_on_gpu(lib):
    """BASELINE.json configs[3] at the seam: 16 nodes (f = 5, Q = 11), 50 000 decisions x 11 consenter signatures through
    VerifyConsenterSigBatch; every proposal must reach its quorum."""
    v = _new(lib)
    try:
        res = hostlib.ReplayResult()
        assert lib.sbvh_replay(v, 16, 10, 1, 50000, 16, ctypes.byref(res)) == 0
        assert res.status == 0 and res.batch_tuples == 550000 and res.proposals_with_quorum == 50000
    finally:
        lib.sbvh_verifier_free(v)


def test_decision_batch_rejects_exactly_the_spoiled_signatures_on_gpu(lib):
    """50 000 decisions x 11 signatures of a 16-node cluster (configs[3]'s shape) in one VerifyConsenterSigBatch with every 7th
    signature spoiled in one of four ways (flipped value byte, unknown signer, message bound to another proposal, another
    consenter's valid signature under this signer's ID): through the device front end and the consenters' wide combs none of
    the 78 571 spoiled signatures is accepted and none of the honest ones rejected."""
    v = _new(lib)
    try:
        counts = (ctypes.c_uint64 * 4)()
        assert lib.sbvh_batch_faults(v, 16, 50000, 16, counts) == 0
        spoiled, spoiled_accepted, honest, honest_rejected = list(counts)
        assert spoiled + honest == 550000 and spoiled == len(range(3, 550000, 7))
        assert spoiled_accepted == 0 and honest_rejected == 0
    finally:
        lib.sbvh_verifier_free(v)


def test_host_verifier_over_all_gpus_of_the_node(lib):
    """Backend device = -1: sbv_init_all + sbv_p256_verify_batch_sharded behind the same api.Verifier (one Verifier per
    replica process drives every GPU of its node: pkg/consensus/consensus.go:35).  On this box that is one GPU."""
    cb = hostlib.BACKEND_FN(lambda *a: -1)
    v = lib.sbvh_verifier_new(0, -1, cb, None, 4096, 200, 0)
    try:
        res = hostlib.ReplayResult()
        assert lib.sbvh_replay(v, 4, 2000, 2, 20, 16, ctypes.byref(res)) == 0
        assert res.status == 0 and res.proposals_with_quorum == 20 and res.max_backend_batch >= 2000
    finally:
        lib.sbvh_verifier_free(v)


def _noop_factory(lib, keep):
    def backend(_tuples, n, bitmap, _user):
        ctypes.memset(bitmap, 0xFF, (n + 7) // 8)
        return 0
    cb = hostlib.BACKEND_FN(backend)
    keep.append(cb)
    return lambda: lib.sbvh_verifier_new(1, 0, cb, None, 4096, 200, 1)


def test_chain_over_the_gpu_delivers_the_same_blocks_as_a_noop_verifier_run(lib):
    """the Verifier traffic of the reference's 4-node chain test (examples/naive_chain/chain_test.go:71-98;
    consensus_amd/host/chain_emul.cc), every node with its own Verifier over libsbv.so, against the same run over no-op
    Verifiers (examples/naive_chain/node.go:64-100): same blocks, same order, same signature sets at every node."""
    keep = []
    cb = hostlib.BACKEND_FN(lambda *a: -1)
    gpu = lambda: lib.sbvh_verifier_new(0, 0, cb, None, 4096, 200, 1)       # noqa: E731
    real = hostlib.ChainRun(lib, gpu, n_nodes=4, blocks=9, batch_size=1)
    ref = hostlib.ChainRun(lib, _noop_factory(lib, keep), n_nodes=4, blocks=9, batch_size=1)
    assert real.rc == 0 and real.unavailable == 0
    assert all(len(l) == 9 for l in real.ledgers) and all(l == real.ledgers[0] for l in real.ledgers)
    assert real.ledgers == ref.ledgers and real.signers == ref.signers
    assert real.rejected_proposals == 0 and real.dropped_votes == 0
    # larger cluster, batched blocks
    real = hostlib.ChainRun(lib, gpu, n_nodes=10, blocks=5, batch_size=100)
    ref = hostlib.ChainRun(lib, _noop_factory(lib, keep), n_nodes=10, blocks=5, batch_size=100)
    assert real.rc == 0 and real.ledgers == ref.ledgers and real.signers == ref.signers
    assert all(len(l) == 5 for l in real.ledgers) and all(len(s) == 7 for node in real.signers for s in node)


def test_chain_faults_are_caught_by_the_gpu_verifier(lib):
    """view_test.go:466 TestBadCommit / view.go:387-392 through the chain emulation: a commit vote signed with the wrong key and a
    client request with a forged signature are well-formed — only the curve arithmetic on the device can reject them."""
    keep = []
    cb = hostlib.BACKEND_FN(lambda *a: -1)
    gpu = lambda: lib.sbvh_verifier_new(0, 0, cb, None, 4096, 200, 1)       # noqa: E731
    real = hostlib.ChainRun(lib, gpu, blocks=5, byzantine_node=2)
    ref = hostlib.ChainRun(lib, _noop_factory(lib, keep), blocks=5, byzantine_node=2)
    assert real.rc == 0 and all(len(l) == 5 for l in real.ledgers)
    assert real.dropped_votes == 15 and ref.dropped_votes == 0
    assert all(2 not in s for node in (0, 2, 3) for s in real.signers[node])
    assert any(2 in s for s in ref.signers[0])
    real = hostlib.ChainRun(lib, gpu, blocks=4, batch_size=3, bad_request_block=2)
    assert real.rejected_proposals == 1 and all(len(l) == 3 for l in real.ledgers)
