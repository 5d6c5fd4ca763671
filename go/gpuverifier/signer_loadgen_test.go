//go:build sbv_loadgen

package gpuverifier

import (
	"testing"

	"github.com/stretchr/testify/assert"

	bft "github.com/hyperledger-labs/SmartBFT/pkg/types"
)

func TestSignBatchFallsBackToSign(t *testing.T) {
	h, be := newKeyedHarness(t, 4, DefaultOptions)
	defer h.v.Close()
	msgs := [][]byte{[]byte("a"), []byte("b"), []byte("c")}
	sigs := (&LoadgenSigner{Signer: *h.nodes[0]}).SignBatch(be, msgs)
	for i, m := range msgs {
		assert.NoError(t, h.v.VerifySignature(bft.Signature{ID: 1, Value: sigs[i], Msg: m}))
	}
}
