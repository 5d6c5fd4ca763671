// Package gpuverifier implements SmartBFT's api.Verifier, api.Signer and api.RequestInspector
// (pkg/api/dependencies.go:46-83) with ECDSA P-256 signatures, verifying them in batches on AMD MI355X GPUs through
// libsbv.so (C-ABI: include/sbv.h of the consensus_amd repository).
//
// Drop this directory into the SmartBFT tree as pkg/gpuverifier.  Two build flavours:
//
//	go test ./pkg/gpuverifier/                      pure Go: every batch is verified with crypto/ecdsa (cpuBackend).
//	                                                This is also the route a GPU build takes for batches below GPUMin
//	                                                and whenever the device reports a fault.
//	go test -tags sbvgpu ./pkg/gpuverifier/         cgo: batches of >= GPUMin signatures go to libsbv.so
//	  CGO_CFLAGS=-I<consensus_amd>/include  CGO_LDFLAGS="-L<consensus_amd>/consensus_amd -lsbv"
//
// What the package does above the C-ABI (all of it is exercised by verifier_test.go without a GPU):
//
//   - wire formats for signed client requests and consenter signature messages (the reference defines none:
//     examples/naive_chain/chain.go:41-58 has unsigned transactions) — formats.go;
//   - a key registry: types.Signature.ID selects the consenter key, Request.ClientID the client key; P-256 keys are also
//     registered with the device (Backend.RegisterKey -> sbv_p256_register_keys) and keep their comb slot;
//   - routing inside the device backend (backend_cgo.go), the same four routes as the C++ mirror's SbvBackend: every
//     item slotted -> raw messages + DER + slots (sbv_p256_verify_msgs_keyed: SHA-256 and DER on the device) or, for
//     small batches, r|s|hash records + slots (sbv_p256_verify_batch_keyed); otherwise generic tuples over all GPUs of
//     the node (sbv_p256_verify_batch_sharded); Options.Scheme selects Ed25519 (sbv_ed25519_verify_msgs) or secp256k1
//     (sbv_secp256k1_verify_batch);
//   - a coalescer for backends that take bursts: the <= N-1 goroutines of View.processCommits
//     (internal/bft/view.go:537-541) each call VerifyConsenterSig with one signature; they are merged into one backend
//     batch by the first of them (the leader of the burst: no dispatcher goroutine to wake), which polls (spin, then
//     yield — no timer) until the expected burst of N-1 is in, a quiet period passes or the window closes.  With the
//     default GPUMin (8) the 15 votes of a 16-node cluster go to the device in one launch (73-75 us through the C++
//     mirror, against ~100 us for one crypto/ecdsa verification per goroutine), the 3 votes of a 4-node cluster are
//     verified on their own goroutines' cores;
//   - VerifyProposal ships all K request signatures of a proposal as ONE batch (internal/bft/view.go:553-559);
//   - a verified-signature cache with an injective key (commit signatures of sequence s come back as
//     prev_commit_signatures at s+1: internal/bft/view.go:376, 630);
//   - Proposal.Digest() computed once per proposal, also for concurrent first callers;
//   - LoadgenSigner.SignBatch (build tag sbv_loadgen only — the device signer is not constant-time): the batch form of api.Signer.Sign over sbv_p256_sign_batch (load generators, replay tools).
//
// A device fault is never reported as an invalid signature: VerifyProposal returning an error deposes the leader
// (internal/bft/view.go:387-392), so on any backend error the batch is re-verified with crypto/ecdsa.
//
// This code was written without a Go toolchain at hand (the build image of consensus_amd has none); the C++ mirror
// consensus_amd/host/verifier.{h,cc} has the same structure, is compiled and tested there, and is the executable
// specification of what follows.
package gpuverifier
