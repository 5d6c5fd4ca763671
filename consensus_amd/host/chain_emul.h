// chain_emul.h — the Verifier / Signer traffic of an N-node chain, sequence by sequence (chain_emul.cc)
#pragma once
#include <vector>

#include "verifier.h"

namespace sbvhost {

struct ChainEmulOptions {
    int blocks = 9;               // chain_test.go:72 orders blockCount - 1 = 9 blocks
    int batch_size = 1;           // NetworkOptions.BatchSize (chain_test.go:79)
    int byzantine_node = 0;       // 1-based id of a node whose commit votes are signed with a key that is not its registered one; 0 = none
    int bad_request_block = 0;    // sequence at which the leader batches one request with a forged signature; 0 = never
};
struct ChainEmulResult {
    std::vector<std::vector<bytes>> ledgers;                        // [node][block] = Proposal.Digest() (raw 32 bytes)
    std::vector<std::vector<std::vector<uint64_t>>> signers;        // [node][block] = ids of the signatures Deliver received
    uint64_t rejected_proposals = 0, dropped_votes = 0, unavailable = 0;
};
// verifiers[i] = node i+1's Verifier.  0 ok; -1 bad arguments; -2 a backend could not answer (never reported as a reject)
int chain_emulate(const std::vector<Verifier*>& verifiers, const ChainEmulOptions& opt, ChainEmulResult* res);

}  // namespace sbvhost
