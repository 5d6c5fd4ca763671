package gpuverifier

import (
	"crypto/ecdsa"
	"crypto/rand"
	"crypto/sha256"

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
