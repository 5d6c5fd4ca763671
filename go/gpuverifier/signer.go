package gpuverifier

import (
	"crypto/ecdsa"
	"crypto/rand"
	"crypto/sha256"
	"encoding/asn1"
	"math/big"

	bft "github.com/hyperledger-labs/SmartBFT/pkg/types"
)

// Signer implements api.Signer (pkg/api/dependencies.go:46-52) for one node: ECDSA P-256, ASN.1 DER signatures over
// SHA-256 of the message — the pair of Verifier above.  Signing stays on the CPU: one signature per sequence per node
// (internal/bft/view.go:481).
type Signer struct {
	ID  uint64
	Key *ecdsa.PrivateKey
}

// Sign: internal/bft/viewchanger.go:445, 1260.
func (s *Signer) Sign(msg []byte) []byte {
	h := sha256.Sum256(msg)
	sig, err := ecdsa.SignASN1(rand.Reader, s.Key, h[:])
	if err != nil {
		panic(err)
	}
	return sig
}

// SignBatch is the batch form of Sign for load generators and replay tools (a consensus node signs once per sequence and
// has no use for it): every message is signed with this node's key through the backend's batch signer
// (sbv_p256_sign_batch: deterministic RFC 6979 nonces, DER-encoded here); a backend without one, a device fault or an
// unusable key sends the affected messages through Sign.  Counterpart: consensus_amd/host (Signer over p256_host.cc) and
// the device entry it is tested against (tests/test_gpu_sign.py).
func (s *Signer) SignBatch(be Backend, msgs [][]byte) [][]byte {
	out := make([][]byte, len(msgs))
	if be != nil && len(msgs) > 0 && s.Key != nil && s.Key.D != nil && s.Key.D.Sign() > 0 && s.Key.D.BitLen() <= 256 {
		var key [32]byte
		s.Key.D.FillBytes(key[:])
		digests := make([][32]byte, len(msgs))
		for i, m := range msgs {
			digests[i] = sha256.Sum256(m)
		}
		sigs, ok, err := be.SignBatch([][32]byte{key}, make([]uint32, len(msgs)), digests)
		for i := range key {
			key[i] = 0
		}
		if err == nil && len(sigs) == len(msgs) {
			for i := range msgs {
				if !ok[i] {
					continue
				}
				der, e := asn1.Marshal(struct{ R, S *big.Int }{new(big.Int).SetBytes(sigs[i][:32]), new(big.Int).SetBytes(sigs[i][32:])})
				if e == nil {
					out[i] = der
				}
			}
		}
	}
	for i, m := range msgs {
		if out[i] == nil {
			out[i] = s.Sign(m)
		}
	}
	return out
}

// SignProposal: internal/bft/view.go:481.  Msg binds the proposal and carries auxiliaryInput verbatim.
func (s *Signer) SignProposal(p bft.Proposal, auxiliaryInput []byte) *bft.Signature {
	msg := ConsenterMsg(proposalDigestRaw(p), auxiliaryInput)
	return &bft.Signature{ID: s.ID, Value: s.Sign(msg), Msg: msg}
}

// SignRequest is the client side: a request VerifyRequest / VerifyProposal accept.
func SignRequest(clientID, id string, payload []byte, key *ecdsa.PrivateKey) []byte {
	u := RequestUnsigned(clientID, id, payload)
	h := sha256.Sum256(u)
	sig, err := ecdsa.SignASN1(rand.Reader, key, h[:])
	if err != nil {
		panic(err)
	}
	return RequestEncode(u, sig)
}
