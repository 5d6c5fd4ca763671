// chain_emul.cc — SURVEY.md §8 row a12: the calls an N-node chain makes on its api.Verifier, in the order the reference
// makes them, so that a run over the product backend can be compared block for block with a run over a no-op Verifier.
//
// The reference's system test (examples/naive_chain/chain_test.go:71-98) orders transactions on a 4-node chain whose
// Node implements every Verifier method as a no-op (examples/naive_chain/node.go:64-100) and asserts that all nodes
// deliver the same blocks in the same order.  With a real Verifier plugged in, the same test must deliver the same
// blocks.  There is no Go toolchain in this image, so the protocol itself is not run here; what IS run is, per
// sequence, exactly the Verifier / Signer traffic of a view without faults, each node against its OWN Verifier instance:
//
//   leader     AssembleProposal: batch of signed client requests, header chained to the previous block,
//              metadata = (view, sequence, CommitSignaturesDigest of the previous decision)  — view.go:905-960
//   follower   verifyProposal: Verifier.VerifyProposal, VerificationSequence, verifyPrevCommitSignatures
//              (serial VerifyConsenterSig over the previous decision's signatures), digest check  — view.go:553-604, 606-644
//   all        Signer.SignProposal -> Commit vote to every other node  — view.go:486-517
//   all        processCommits: one VerifyConsenterSig per vote, concurrently, until Quorum-1 are valid  — view.go:519-551
//   all        decide -> Deliver(proposal, signatures): append to the ledger  — view.go:851-871
//
// Faults that make the Verifier matter (a no-op Verifier and a real one diverge on these, as they must):
//   byzantine_node    that node's commit votes are well-formed and name the right proposal but are signed with another key
//                     (only the curve arithmetic can tell)
//   bad_request_block at that sequence the leader's batch holds one request with a forged client signature; a real
//                     Verifier rejects the proposal (the reference then complains about the leader, view.go:387-392:
//                     here the sequence is skipped and counted), a no-op one commits it
//
// This is protocol-free by construction (no timers, no view change, no network): it is test and measurement scaffolding
// around the Verifier seam, not a consensus implementation.
#include <atomic>
#include <cstring>
#include <memory>
#include <thread>

#include "chain_emul.h"
#include "p256_host.h"

namespace sbvhost {

namespace {
bytes u64be(uint64_t v) {
    bytes s(8, '\0');
    for (int b = 0; b < 8; ++b) s[(size_t)b] = (char)(v >> (56 - 8 * b));
    return s;
}
bytes sha(const bytes& b) { return sha256(b); }
void key_from_label(const std::string& label, uint8_t sk[32]) {
    sha256(label.data(), label.size(), sk);
    sk[0] &= 0x7f;                           // below the group order
}
struct Block {
    Proposal proposal;
    std::vector<Signature> signatures;      // what Deliver received with it
};
}  // namespace

int chain_emulate(const std::vector<Verifier*>& verifiers, const ChainEmulOptions& opt, ChainEmulResult* res) {
    const int N = (int)verifiers.size();
    if (N < 1 || opt.blocks < 0 || opt.batch_size < 1) return -1;
    int Q = 0, F = 0;
    compute_quorum((uint64_t)N, &Q, &F);
    res->ledgers.assign((size_t)N, std::vector<bytes>());
    res->signers.assign((size_t)N, std::vector<std::vector<uint64_t>>());
    res->rejected_proposals = 0;
    res->dropped_votes = 0;
    res->unavailable = 0;

    std::vector<std::unique_ptr<Signer>> nodes;
    for (int i = 0; i < N; ++i) {
        uint8_t sk[32];
        key_from_label("chain-node-" + std::to_string(i + 1), sk);
        nodes.emplace_back(new Signer((uint64_t)i + 1, sk));
    }
    uint8_t ask[32], msk[32];
    key_from_label("chain-client-alice", ask);
    key_from_label("chain-client-mallory", msk);
    Signer alice(0, ask), mallory(0, msk);
    for (Verifier* v : verifiers) {
        for (int i = 0; i < N; ++i) v->RegisterConsenter((uint64_t)i + 1, nodes[(size_t)i]->public_key());
        v->RegisterClient("alice", alice.public_key());
        v->SetVerificationSequence(0);
    }

    const int leader = 0;                       // node 1 leads the whole (fault-free) view
    std::vector<Block> last((size_t)N);         // every node's latest delivered block
    bytes prev_hash = bytes(32, '\0');
    uint64_t tx = 0;
    for (int seq = 1; seq <= opt.blocks; ++seq) {
        // ---- leader: assemble ------------------------------------------------------------------------------------
        std::vector<bytes> reqs;
        for (int k = 0; k < opt.batch_size; ++k, ++tx) {
            const bytes u = request_unsigned("alice", "tx" + std::to_string(tx), "pay " + std::to_string(tx));
            const bool forge = seq == opt.bad_request_block && k == opt.batch_size / 2;
            reqs.push_back(request_encode(u, forge ? mallory.Sign(u) : alice.Sign(u)));
        }
        Proposal p;
        p.set_payload(payload_encode(reqs));
        p.set_header(u64be((uint64_t)seq) + prev_hash + sha(p.payload()));                        // BlockHeader{Sequence, PrevHash, DataHash}
        const std::vector<Signature>& prev_sigs = last[(size_t)leader].signatures;
        p.set_metadata(u64be(0) + u64be((uint64_t)seq) + commit_signatures_digest(prev_sigs));  // view, sequence, prev commit digest

        // ---- followers: verifyProposal ---------------------------------------------------------------------------
        std::vector<int> accepted((size_t)N, 1);
        for (int i = 0; i < N; ++i) {
            if (i == leader) continue;
            Verifier& V = *verifiers[(size_t)i];
            std::vector<RequestInfo> infos;
            Status st = V.VerifyProposal(p, &infos);
            if (st.code == Status::UNAVAILABLE) { ++res->unavailable; return -2; }
            bool ok = st.ok() && (int)infos.size() == opt.batch_size && V.VerificationSequence() == 0;
            if (ok && seq > 1) {                                                              // verifyPrevCommitSignatures
                const Block& prev = last[(size_t)i];
                for (const Signature& s : prev_sigs) {
                    bytes aux;
                    st = V.VerifyConsenterSig(s, prev.proposal, &aux);
                    if (st.code == Status::UNAVAILABLE) { ++res->unavailable; return -2; }
                    if (!st.ok()) { ok = false; break; }
                }
                ok = ok && commit_signatures_digest(prev_sigs) == p.metadata().substr(16);
            }
            accepted[(size_t)i] = ok ? 1 : 0;
        }
        int accepting = 0;
        for (int i = 0; i < N; ++i) accepting += accepted[(size_t)i];
        if (accepting < Q) {                    // no quorum of prepares: the reference changes view; here the sequence is skipped
            ++res->rejected_proposals;
            continue;
        }

        // ---- every accepting node signs its commit vote ------------------------------------------------------------
        std::vector<Signature> votes((size_t)N);
        for (int i = 0; i < N; ++i) {
            if (!accepted[(size_t)i]) continue;
            if (i + 1 == opt.byzantine_node) {         // well-formed vote on the right proposal, signed with a key that is not the node's
                uint8_t sk[32];
                key_from_label("chain-node-impostor-" + std::to_string(i + 1), sk);
                Signer impostor((uint64_t)i + 1, sk);
                votes[(size_t)i] = impostor.SignProposal(p, "prepares-from-" + std::to_string(i + 1));
            } else {
                votes[(size_t)i] = nodes[(size_t)i]->SignProposal(p, "prepares-from-" + std::to_string(i + 1));
            }
        }

        // ---- processCommits at every node: one concurrent VerifyConsenterSig per received vote --------------------
        std::atomic<int> dropped(0), unavailable(0);
        std::vector<std::vector<Signature>> collected((size_t)N);
        std::vector<std::thread> th;
        std::vector<std::vector<int>> verdict((size_t)N, std::vector<int>((size_t)N, 0));
        for (int i = 0; i < N; ++i) {
            if (!accepted[(size_t)i]) continue;
            for (int j = 0; j < N; ++j) {
                if (j == i || !accepted[(size_t)j]) continue;
                th.emplace_back([&, i, j] {
                    bytes aux;
                    const Status st = verifiers[(size_t)i]->VerifyConsenterSig(votes[(size_t)j], p, &aux);
                    if (st.code == Status::UNAVAILABLE) ++unavailable;
                    else if (!st.ok()) ++dropped;
                    else verdict[(size_t)i][(size_t)j] = 1;
                });
            }
        }
        for (auto& t : th) t.join();
        if (unavailable.load()) { res->unavailable += (uint64_t)unavailable.load(); return -2; }
        res->dropped_votes += (uint64_t)dropped.load();

        // ---- decide: Quorum-1 valid votes of others + the node's own signature -------------------------------------
        std::vector<int> decided((size_t)N, 0);
        for (int i = 0; i < N; ++i) {
            if (!accepted[(size_t)i]) continue;
            std::vector<Signature> sigs;
            for (int j = 0; j < N && (int)sigs.size() < Q - 1; ++j)
                if (verdict[(size_t)i][(size_t)j]) sigs.push_back(votes[(size_t)j]);
            if ((int)sigs.size() < Q - 1) continue;
            sigs.push_back(votes[(size_t)i]);
            decided[(size_t)i] = 1;
            std::vector<uint64_t> ids;
            for (const Signature& s : sigs) ids.push_back(s.id);
            res->ledgers[(size_t)i].push_back(proposal_digest_raw(p));
            res->signers[(size_t)i].push_back(ids);
            last[(size_t)i].proposal = p;
            last[(size_t)i].signatures = sigs;
        }
        if (decided[(size_t)leader]) prev_hash = sha(p.header());
        // a node that did not decide would sync in the reference (out of scope: Synchronizer); keep it in step so that the
        // emulation can go on — its ledger simply lacks the block
        for (int i = 0; i < N; ++i)
            if (!decided[(size_t)i] && decided[(size_t)leader]) last[(size_t)i] = last[(size_t)leader];
    }
    return 0;
}

}  // namespace sbvhost
