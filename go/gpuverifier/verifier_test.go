package gpuverifier

import (
	"crypto/ecdsa"
	"crypto/elliptic"
	"crypto/rand"
	"fmt"
	"sync"
	"testing"

	bft "github.com/hyperledger-labs/SmartBFT/pkg/types"
	"github.com/stretchr/testify/assert"
)

// The seam tests of the reference use mocks below the interface (internal/bft/view_test.go:466, 533;
// internal/bft/controller_test.go:548; internal/bft/requestpool_test.go:264); these use real signatures and whatever
// backend the build selects (crypto/ecdsa without the sbvgpu tag).

type harness struct {
	v       *Verifier
	nodes   []*Signer
	clients map[string]*ecdsa.PrivateKey
}

func newHarness(t *testing.T, n int, opt Options) *harness {
	be, err := NewDeviceBackend()
	assert.NoError(t, err)
	h := &harness{v: New(be, opt), clients: map[string]*ecdsa.PrivateKey{}}
	for i := 1; i <= n; i++ {
		k, _ := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
		h.nodes = append(h.nodes, &Signer{ID: uint64(i), Key: k})
		h.v.RegisterConsenter(uint64(i), &k.PublicKey)
	}
	for i := 0; i < 3; i++ {
		k, _ := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
		id := fmt.Sprintf("alice%d", i)
		h.clients[id] = k
		h.v.RegisterClient(id, &k.PublicKey)
	}
	return h
}

func (h *harness) request(client, id string, corrupt bool) []byte {
	r := SignRequest(client, id, []byte("tx"), h.clients[client])
	if corrupt {
		r[len(r)-1] ^= 1
	}
	return r
}

func TestSignThenVerify(t *testing.T) {
	h := newHarness(t, 4, DefaultOptions)
	defer h.v.Close()
	msg := []byte("raw view data")
	sig := h.nodes[0].Sign(msg)
	assert.NoError(t, h.v.VerifySignature(bft.Signature{ID: 1, Value: sig, Msg: msg}))
	assert.Error(t, h.v.VerifySignature(bft.Signature{ID: 1, Value: sig, Msg: append(msg, '!')}))
	assert.Error(t, h.v.VerifySignature(bft.Signature{ID: 2, Value: sig, Msg: msg}))
	assert.Error(t, h.v.VerifySignature(bft.Signature{ID: 9, Value: sig, Msg: msg}))
}

func TestCacheKeyIsInjective(t *testing.T) {
	h := newHarness(t, 4, DefaultOptions)
	defer h.v.Close()
	msg := []byte("view-data:0123456789abcdefghijklmnopqrstuvwxyz")
	sig := h.nodes[0].Sign(msg)
	assert.NoError(t, h.v.VerifySignature(bft.Signature{ID: 1, Value: sig, Msg: msg}))
	for _, k := range []int{1, 7, len(msg) - 1} {
		shifted := bft.Signature{ID: 1, Value: append(append([]byte(nil), sig...), msg[:k]...), Msg: msg[k:]}
		assert.Error(t, h.v.VerifySignature(shifted), "a verified (sig, msg) must not vouch for a shifted boundary")
	}
}

func TestConsenterSigBindsProposalAndReturnsAux(t *testing.T) {
	h := newHarness(t, 4, DefaultOptions)
	defer h.v.Close()
	prop := bft.Proposal{Payload: PayloadEncode(nil), Header: []byte("h"), Metadata: []byte("m")}
	sig := h.nodes[2].SignProposal(prop, []byte("prepares-from"))
	aux, err := h.v.VerifyConsenterSig(*sig, prop)
	assert.NoError(t, err)
	assert.Equal(t, []byte("prepares-from"), aux)
	assert.Equal(t, []byte("prepares-from"), h.v.AuxiliaryData(sig.Msg))
	other := prop
	other.Header = []byte("h2")
	_, err = h.v.VerifyConsenterSig(*sig, other)
	assert.Error(t, err)
}

func TestCommitVotesAreCoalesced(t *testing.T) {
	h := newHarness(t, 16, DefaultOptions)
	defer h.v.Close()
	prop := bft.Proposal{Payload: PayloadEncode(nil), Header: []byte("h"), Metadata: []byte("m")}
	var wg sync.WaitGroup
	errs := make([]error, 15)
	for i := 0; i < 15; i++ { // N - 1 concurrent votes (internal/bft/view.go:537-541)
		wg.Add(1)
		go func(i int) {
			defer wg.Done()
			_, errs[i] = h.v.VerifyConsenterSig(*h.nodes[i+1].SignProposal(prop, nil), prop)
		}(i)
	}
	wg.Wait()
	for _, e := range errs {
		assert.NoError(t, e)
	}
}

func TestLeaderAndPool(t *testing.T) {
	h := newHarness(t, 4, DefaultOptions)
	defer h.v.Close()
	good := h.request("alice0", "req-1", false)
	info, err := h.v.VerifyRequest(good)
	assert.NoError(t, err)
	assert.Equal(t, bft.RequestInfo{ClientID: "alice0", ID: "req-1"}, info)
	assert.Equal(t, info, h.v.RequestID(good))
	_, err = h.v.VerifyRequest(h.request("alice1", "req-2", true))
	assert.Error(t, err)
	_, err = h.v.VerifyRequest(good[:len(good)-3])
	assert.Error(t, err)
	// Pool.Prune pattern (internal/bft/requestpool.go:335-371): exactly the rejects leave
	kept := 0
	for i := 0; i < 40; i++ {
		if _, err := h.v.VerifyRequest(h.request(fmt.Sprintf("alice%d", i%3), fmt.Sprintf("r%d", i), i%4 == 1)); err == nil {
			kept++
		}
	}
	assert.Equal(t, 30, kept)
}

func TestVerifyProposalBatch(t *testing.T) {
	for _, gpuMin := range []int{0, 1 << 30} { // through the backend, and through the crypto/ecdsa route
		opt := DefaultOptions
		opt.GPUMin = gpuMin
		h := newHarness(t, 4, opt)
		var reqs [][]byte
		var want []bft.RequestInfo
		for i := 0; i < 100; i++ {
			c := fmt.Sprintf("alice%d", i%3)
			reqs = append(reqs, h.request(c, fmt.Sprintf("r%d", i), false))
			want = append(want, bft.RequestInfo{ClientID: c, ID: fmt.Sprintf("r%d", i)})
		}
		prop := bft.Proposal{Payload: PayloadEncode(reqs), Header: []byte("h"), Metadata: []byte("m")}
		infos, err := h.v.VerifyProposal(prop)
		assert.NoError(t, err)
		assert.Equal(t, want, infos)
		assert.Equal(t, want, h.v.RequestsFromProposal(prop)) // internal/bft/view.go:395, 419
		reqs[57] = h.request("alice0", "r57", true)
		_, err = h.v.VerifyProposal(bft.Proposal{Payload: PayloadEncode(reqs), Header: []byte("h"), Metadata: []byte("m")})
		assert.Error(t, err)
		prop.VerificationSequence = 7
		_, err = h.v.VerifyProposal(prop)
		assert.Error(t, err)
		h.v.Close()
	}
}

func TestVerifyDecisionsQuorum(t *testing.T) {
	h := newHarness(t, 16, DefaultOptions)
	defer h.v.Close()
	var ds []bft.Decision
	for d := 0; d < 20; d++ {
		p := bft.Proposal{Payload: []byte(fmt.Sprintf("decision-%d", d)), Header: []byte("h"), Metadata: []byte("m")}
		dec := bft.Decision{Proposal: p}
		for j := 0; j < 11; j++ {
			s := h.nodes[(d+j)%16].SignProposal(p, nil)
			if d%5 == 0 && j < 2 { // two bad signatures: 9 < Q = 11 valid
				s.Value[len(s.Value)-1] ^= 1
			}
			if d%7 == 3 && j == 1 { // the same signer twice must not count twice
				s = h.nodes[d%16].SignProposal(p, nil)
			}
			dec.Signatures = append(dec.Signatures, *s)
		}
		ds = append(ds, dec)
	}
	got := h.v.VerifyDecisions(ds, 11)
	for d := range ds {
		assert.Equal(t, !(d%5 == 0 || d%7 == 3), got[d], "decision %d", d)
	}
}
