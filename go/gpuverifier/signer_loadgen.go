//go:build sbv_loadgen

package gpuverifier

import (
	"crypto/sha256"
	"encoding/asn1"
	"math/big"
)

// LoadgenSigner is a Signer whose key may go to the device's batch signer.  It is its own type, in a file that only a
// build with -tags sbv_loadgen contains, because sbv_p256_sign_batch is NOT constant-time (secret-indexed table lookups in
// HBM, include/sbv.h): a consensus node's long-term key must not be able to reach it by accident, so the production Signer
// has no such method.  For load generators and replay tools over throw-away keys.
type LoadgenSigner struct{ Signer }

// SignBatch is the batch form of Sign (a consensus node signs once per sequence and
// has no use for it): every message is signed with this key through the backend's batch signer
// (sbv_p256_sign_batch: deterministic RFC 6979 nonces, DER-encoded here); a backend without one, a device fault or an
// unusable key sends the affected messages through Sign.  Counterpart: consensus_amd/host (Signer over p256_host.cc) and
// the device entry it is tested against (tests/test_gpu_sign.py).
func (s *LoadgenSigner) SignBatch(be Backend, msgs [][]byte) [][]byte {
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
