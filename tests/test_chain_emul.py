"""SURVEY.md §8 row a12 (CPU tier): the Verifier traffic of a 4-node chain (consensus_amd/host/chain_emul.cc, mirroring
examples/naive_chain/chain_test.go:71-98), once over Verifiers that really check signatures (stand-in backend: the oracle)
and once over no-op Verifiers (the reference's own Node: examples/naive_chain/node.go:64-100 returns nil everywhere).

  * without faults both runs deliver the same blocks, in the same order, at every node   (the a12 assertion)
  * a forged commit vote / a forged client request is where the two runs must part
"""
import pytest

import hostlib


@pytest.fixture(scope="module")
def lib():
    return hostlib.load()


def checking(lib, oracle, keep):
    def backend(tuples, n, bitmap, _user):
        oracle.sbvo_p256_verify_batch(tuples, n, bitmap, 1)
        return 0
    cb = hostlib.BACKEND_FN(backend)
    keep.append(cb)
    return lambda: lib.sbvh_verifier_new(1, 0, cb, None, 4096, 200, 1)


def noop(lib, keep):
    import ctypes

    def backend(_tuples, n, bitmap, _user):
        ctypes.memset(bitmap, 0xFF, (n + 7) // 8)        # every signature "verifies"
        return 0
    cb = hostlib.BACKEND_FN(backend)
    keep.append(cb)
    return lambda: lib.sbvh_verifier_new(1, 0, cb, None, 4096, 200, 1)


def test_fault_free_chain_delivers_the_same_blocks_as_a_noop_verifier_run(lib, oracle):
    keep = []
    real = hostlib.ChainRun(lib, checking(lib, oracle, keep), n_nodes=4, blocks=9, batch_size=1)
    ref = hostlib.ChainRun(lib, noop(lib, keep), n_nodes=4, blocks=9, batch_size=1)
    assert real.rc == 0 and ref.rc == 0 and real.unavailable == 0
    assert all(len(l) == 9 for l in real.ledgers)                      # chain_test.go: every node delivers every block ...
    assert all(l == real.ledgers[0] for l in real.ledgers)             # ... the same ones, in the same order
    assert real.ledgers == ref.ledgers                                 # and exactly what the no-op Verifier run delivers
    assert real.signers == ref.signers
    assert real.rejected_proposals == 0 and real.dropped_votes == 0
    assert all(len(s) == 3 for node in real.signers for s in node)     # quorum of 4 = 3 signatures per decision


def test_batched_blocks(lib, oracle):
    keep = []
    real = hostlib.ChainRun(lib, checking(lib, oracle, keep), n_nodes=7, blocks=4, batch_size=5)
    ref = hostlib.ChainRun(lib, noop(lib, keep), n_nodes=7, blocks=4, batch_size=5)
    assert real.rc == 0 and real.ledgers == ref.ledgers and real.signers == ref.signers
    assert all(len(l) == 4 for l in real.ledgers)
    assert all(len(s) == 5 for node in real.signers for s in node)     # n = 7: f = 2, q = 5


def test_forged_commit_votes_are_dropped_only_by_a_checking_verifier(lib, oracle):
    keep = []
    real = hostlib.ChainRun(lib, checking(lib, oracle, keep), blocks=5, byzantine_node=2)
    ref = hostlib.ChainRun(lib, noop(lib, keep), blocks=5, byzantine_node=2)
    assert real.rc == 0 and ref.rc == 0
    # view_test.go:466 TestBadCommit: the bad vote is dropped, the remaining honest votes still decide
    assert all(len(l) == 5 for l in real.ledgers)
    assert real.dropped_votes == 3 * 5                                 # nodes 1, 3, 4 each drop node 2's vote, every block
    for node in (0, 2, 3):
        assert all(2 not in s for s in real.signers[node])
    assert any(2 in s for s in ref.signers[0])                         # the no-op run hands the forged signature to Deliver
    assert ref.dropped_votes == 0


def test_forged_client_request_stops_the_proposal_only_under_a_checking_verifier(lib, oracle):
    keep = []
    real = hostlib.ChainRun(lib, checking(lib, oracle, keep), blocks=4, batch_size=3, bad_request_block=2)
    ref = hostlib.ChainRun(lib, noop(lib, keep), blocks=4, batch_size=3, bad_request_block=2)
    assert real.rejected_proposals == 1 and all(len(l) == 3 for l in real.ledgers)       # view.go:387-392
    assert ref.rejected_proposals == 0 and all(len(l) == 4 for l in ref.ledgers)
    assert real.ledgers[0][0] == ref.ledgers[0][0]                                       # identical until the forgery
