package gpuverifier

import (
	"bytes"
	"crypto/ecdsa"
	"crypto/ed25519"
	"crypto/elliptic"
	"crypto/rand"
	"errors"
	"fmt"
	"sync"
	"testing"
	"time"

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

// ---- the registered-key ("keyed") route ---------------------------------------------------------------------------------
// recordingBackend stands below the seam the way mocks.VerifierMock stands above it in the reference's tests
// (internal/bft/mocks/verifier_mock.go): it hands out key slots like sbv_p256_register_keys, records which route a batch
// would take in backend_cgo.go (every item slotted -> the keyed entries; otherwise generic tuples) and judges with
// crypto/ecdsa — resolving the key FROM THE SLOT when there is one, as the device does, so a wrong slot is a wrong verdict.
type recordingBackend struct {
	mu       sync.Mutex
	keys     []*ecdsa.PublicKey
	keyed    int // batches in which every item had a slot
	generic  int
	sizes    []int
	widened  []int32
	failNext bool
	failAll  bool
}

func (b *recordingBackend) RegisterKey(pub *ecdsa.PublicKey) int32 {
	b.mu.Lock()
	defer b.mu.Unlock()
	for i, k := range b.keys {
		if k.X.Cmp(pub.X) == 0 && k.Y.Cmp(pub.Y) == 0 {
			return int32(i)
		}
	}
	b.keys = append(b.keys, pub)
	return int32(len(b.keys) - 1)
}

func (b *recordingBackend) WidenKey(slot int32) {
	b.mu.Lock()
	defer b.mu.Unlock()
	b.widened = append(b.widened, slot)
}

func (b *recordingBackend) Verify(scheme Scheme, items []Item) ([]bool, error) {
	b.mu.Lock()
	defer b.mu.Unlock()
	if b.failNext || b.failAll {
		b.failNext = false
		return nil, errors.New("device fault")
	}
	all := true
	for _, it := range items {
		all = all && it.Slot >= 0
	}
	if all {
		b.keyed++
	} else {
		b.generic++
	}
	b.sizes = append(b.sizes, len(items))
	resolved := make([]Item, len(items))
	for i, it := range items {
		resolved[i] = it
		if it.Slot >= 0 {
			resolved[i].Pub = nil
			if int(it.Slot) < len(b.keys) {
				resolved[i].Pub = b.keys[it.Slot]
			}
		}
	}
	return cpuBackend{}.Verify(scheme, resolved)
}

func (b *recordingBackend) SignBatch([][32]byte, []uint32, [][32]byte) ([][64]byte, []bool, error) {
	return nil, nil, ErrNoBatchSigner
}
func (b *recordingBackend) Close() {}

func newKeyedHarness(t *testing.T, n int, opt Options) (*harness, *recordingBackend) {
	be := &recordingBackend{}
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
	return h, be
}

// TestBadCommit mirrors internal/bft/view_test.go:466 below the seam: there the VerifierMock answers every
// VerifyConsenterSig with an error and the view must log "Couldn't verify 2's signature"; here the commit signatures are
// real — one bound to a wrong digest, one with a damaged signature value — and the Verifier must be the one to refuse
// them, through the registered-key route.
func TestBadCommit(t *testing.T) {
	opt := DefaultOptions
	opt.GPUMin = 0
	h, be := newKeyedHarness(t, 4, opt)
	defer h.v.Close()
	prop := bft.Proposal{Payload: PayloadEncode(nil), Header: []byte("h"), Metadata: []byte("m"), VerificationSequence: 0}
	wrong := prop
	wrong.Metadata = []byte("another proposal")
	commitWrongDigest := h.nodes[0].SignProposal(wrong, nil)
	_, err := h.v.VerifyConsenterSig(*commitWrongDigest, prop)
	assert.Error(t, err) // "Got wrong digest"
	commit2 := h.nodes[1].SignProposal(prop, nil)
	commit2.Value[len(commit2.Value)-1] ^= 0x40
	_, err = h.v.VerifyConsenterSig(*commit2, prop)
	assert.Error(t, err) // "Couldn't verify 2's signature"
	good := h.nodes[2].SignProposal(prop, nil)
	_, err = h.v.VerifyConsenterSig(*good, prop)
	assert.NoError(t, err)
	stolen := *good
	stolen.ID = 4 // node 3's signature presented as node 4's: the slot decides the key
	_, err = h.v.VerifyConsenterSig(stolen, prop)
	assert.Error(t, err)
	assert.True(t, be.keyed >= 3 && be.generic == 0, "consenter signatures must take the keyed route: %d keyed, %d generic", be.keyed, be.generic)
	// RegisterConsenter names every consenter's slot for a wide comb (Backend.WidenKey -> sbv_p256_widen_keys); clients' slots are not named
	assert.Equal(t, []int32{0, 1, 2, 3}, be.widened)
}

// TestNormalPath mirrors internal/bft/view_test.go:533: one view taken through pre-prepare, prepare and commit to a
// decision, twice — the Verifier traffic of a follower: VerifyProposal (view.go:555), the previous sequence's commit
// signatures one by one (view.go:630-644), then N - 1 concurrent commit votes of which Q - 1 suffice (view.go:531-541).
func TestNormalPath(t *testing.T) {
	opt := DefaultOptions
	opt.GPUMin = 0
	h, be := newKeyedHarness(t, 4, opt)
	defer h.v.Close()
	var prev []*bft.Signature
	var prevProp bft.Proposal
	for seq := 0; seq < 2; seq++ {
		var reqs [][]byte
		for i := 0; i < 10; i++ {
			reqs = append(reqs, h.request(fmt.Sprintf("alice%d", i%3), fmt.Sprintf("s%d-r%d", seq, i), false))
		}
		prop := bft.Proposal{Payload: PayloadEncode(reqs), Header: []byte{byte(seq)}, Metadata: []byte("md")}
		infos, err := h.v.VerifyProposal(prop)
		assert.NoError(t, err)
		assert.Len(t, infos, 10)
		for _, s := range prev { // verifyPrevCommitSignatures
			_, err := h.v.VerifyConsenterSig(*s, prevProp)
			assert.NoError(t, err)
		}
		var wg sync.WaitGroup
		sigs := make([]*bft.Signature, 3)
		errs := make([]error, 3)
		for i := 0; i < 3; i++ { // nodes 2, 3, 4 vote; this node is 1
			sigs[i] = h.nodes[i+1].SignProposal(prop, []byte("prepares"))
			wg.Add(1)
			go func(i int) {
				defer wg.Done()
				_, errs[i] = h.v.VerifyConsenterSig(*sigs[i], prop)
			}(i)
		}
		wg.Wait()
		for _, e := range errs {
			assert.NoError(t, e)
		}
		prev, prevProp = sigs[:2], prop // Q - 1 = 2 signatures travel with the next pre-prepare
	}
	assert.Equal(t, 0, be.generic, "every key of this run is registered: nothing may take the generic route")
	assert.True(t, be.keyed >= 4)
	maxBatch := 0
	for _, n := range be.sizes {
		if n > maxBatch {
			maxBatch = n
		}
	}
	assert.Equal(t, 10, maxBatch, "the proposal's request signatures go as ONE batch")
}

// TestControllerLeaderRequestHandling mirrors internal/bft/controller_test.go:548 ("bad request" / "good request"): the
// leader verifies a forwarded request before pooling it (controller.go:233-246); VerifyRequest's error is what keeps an
// unauthorized request out of the pool.
func TestControllerLeaderRequestHandling(t *testing.T) {
	opt := DefaultOptions
	opt.GPUMin = 0
	h, be := newKeyedHarness(t, 4, opt)
	defer h.v.Close()
	for _, tc := range []struct {
		description   string
		req           []byte
		shouldEnqueue bool
	}{
		{"bad request: damaged signature", h.request("alice0", "r1", true), false},
		{"bad request: unauthorized user", SignRequest("mallory", "r2", []byte("tx"), h.clients["alice0"]), false},
		{"bad request: someone else's key", SignRequest("alice1", "r3", []byte("tx"), h.clients["alice0"]), false},
		{"good request", h.request("alice2", "r4", false), true},
	} {
		t.Run(tc.description, func(t *testing.T) {
			info, err := h.v.VerifyRequest(tc.req)
			if tc.shouldEnqueue {
				assert.NoError(t, err)
				assert.Equal(t, h.v.RequestID(tc.req), info)
			} else {
				assert.Error(t, err)
			}
		})
	}
	assert.Equal(t, 0, be.generic)
}

// TestReqPoolPrune mirrors internal/bft/requestpool_test.go:264: Pool.Prune keeps exactly the requests its predicate
// accepts (requestpool.go:335-371); the predicate the Controller passes is VerifyRequest (controller.go:742-745).  Here a
// client's key is revoked (re-registered with another key, as after a configuration change): its pooled requests go, the
// others stay.
func TestReqPoolPrune(t *testing.T) {
	opt := DefaultOptions
	opt.GPUMin = 0
	h, _ := newKeyedHarness(t, 4, opt)
	defer h.v.Close()
	byteReq1 := h.request("alice0", "1", false)
	byteReq2 := h.request("alice1", "2", false)
	pool := [][]byte{byteReq1, byteReq2}
	prune := func() [][]byte {
		var kept [][]byte
		for _, r := range pool {
			if _, err := h.v.VerifyRequest(r); err == nil {
				kept = append(kept, r)
			}
		}
		return kept
	}
	assert.Len(t, prune(), 2)
	fresh, _ := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
	h.v.RegisterClient("alice0", &fresh.PublicKey) // revoked
	kept := prune()
	assert.Len(t, kept, 1)
	assert.True(t, bytes.Equal(byteReq2, kept[0]))
}

// A device fault must never look like an invalid signature (internal/bft/view.go:387-392: a rejected proposal deposes the
// leader): the batch is judged again by crypto/ecdsa.
func TestDeviceFaultFallsBackToCPU(t *testing.T) {
	opt := DefaultOptions
	opt.GPUMin = 0
	h, be := newKeyedHarness(t, 4, opt)
	defer h.v.Close()
	var reqs [][]byte
	for i := 0; i < 40; i++ {
		reqs = append(reqs, h.request(fmt.Sprintf("alice%d", i%3), fmt.Sprintf("r%d", i), false))
	}
	be.mu.Lock()
	be.failNext = true
	be.mu.Unlock()
	_, err := h.v.VerifyProposal(bft.Proposal{Payload: PayloadEncode(reqs), Header: []byte("h"), Metadata: []byte("m")})
	assert.NoError(t, err)
}

// Default routing (GPUMin = 8): the 3 commit votes of a 4-node cluster never reach the device — each is verified on its own
// goroutine's core; the 15 votes of a 16-node cluster are merged by the first of them and go to the backend when at least 8
// made one batch (how a burst is cut depends on the scheduler, so only the verdicts are asserted there); a K = 100 proposal
// goes to the backend, slotted.
func TestDefaultRouting(t *testing.T) {
	for _, nodes := range []int{4, 16} {
		h, be := newKeyedHarness(t, nodes, DefaultOptions)
		prop := bft.Proposal{Payload: PayloadEncode(nil), Header: []byte("h"), Metadata: []byte("m")}
		var wg sync.WaitGroup
		for i := 0; i < nodes-1; i++ {
			wg.Add(1)
			go func(i int) {
				defer wg.Done()
				_, err := h.v.VerifyConsenterSig(*h.nodes[i+1].SignProposal(prop, nil), prop)
				assert.NoError(t, err)
			}(i)
		}
		wg.Wait()
		if nodes == 4 {
			assert.Equal(t, 0, be.keyed+be.generic)
		} else {
			assert.LessOrEqual(t, be.keyed+be.generic, 1) // at most one batch of >= 8 out of 15
		}
		h.v.Close()
	}
	h, be := newKeyedHarness(t, 16, DefaultOptions)
	defer h.v.Close()
	var reqs [][]byte
	for i := 0; i < 100; i++ {
		reqs = append(reqs, h.request(fmt.Sprintf("alice%d", i%3), fmt.Sprintf("r%d", i), false))
	}
	_, err := h.v.VerifyProposal(bft.Proposal{Payload: PayloadEncode(reqs), Header: []byte("h"), Metadata: []byte("m")})
	assert.NoError(t, err)
	assert.Equal(t, 1, be.keyed)
	assert.Equal(t, 0, be.generic)
}

// Unregistered client keys (Options.DeviceClientKeys = false): request signatures travel as generic tuples.
func TestOpenClientPopulationTakesTheGenericRoute(t *testing.T) {
	opt := DefaultOptions
	opt.GPUMin = 0
	opt.DeviceClientKeys = false
	h, be := newKeyedHarness(t, 4, opt)
	defer h.v.Close()
	_, err := h.v.VerifyRequest(h.request("alice0", "r", false))
	assert.NoError(t, err)
	assert.Equal(t, 1, be.generic)
	assert.Equal(t, 0, be.keyed)
}

// SchemeEd25519: the same seam over crypto/ed25519 (BASELINE.json configs[4]).
func TestEd25519Scheme(t *testing.T) {
	opt := DefaultOptions
	opt.Scheme = SchemeEd25519
	v := New(nil, opt)
	defer v.Close()
	pub, priv, _ := ed25519.GenerateKey(rand.Reader)
	v.RegisterConsenterRaw(1, pub)
	msg := []byte("view data")
	sig := ed25519.Sign(priv, msg)
	assert.NoError(t, v.VerifySignature(bft.Signature{ID: 1, Value: sig, Msg: msg}))
	assert.Error(t, v.VerifySignature(bft.Signature{ID: 1, Value: sig, Msg: append(msg, 'x')}))
	assert.Error(t, v.VerifySignature(bft.Signature{ID: 1, Value: sig[:63], Msg: msg}))
}

// SchemeSecp256k1 without a device: nobody can judge, and that is reported as an error, never as a verdict.
func TestSecp256k1WithoutDeviceCannotJudge(t *testing.T) {
	opt := DefaultOptions
	opt.Scheme = SchemeSecp256k1
	v := New(nil, opt)
	defer v.Close()
	v.RegisterConsenterRaw(1, make([]byte, 64))
	assert.Error(t, v.VerifySignature(bft.Signature{ID: 1, Value: []byte{0x30, 0x06, 2, 1, 1, 2, 1, 1}, Msg: []byte("m")}))
}

// A batch nobody could judge (here: every Verify fails, and the CPU side has no verifier for the scheme) must reach the
// caller as an error WITHOUT being cached: the same signature verifies once the device is back.
func TestCoalescedDeviceFaultIsNotCachedAsInvalid(t *testing.T) {
	opt := DefaultOptions
	opt.GPUMin = 1 // every single call goes through the coalescer (submit / serve)
	opt.CacheVerified = true
	h, be := newKeyedHarness(t, 4, opt)
	defer h.v.Close()
	h.v.cpu = failingBackend{}
	msg := []byte("view data")
	sig := h.nodes[1].Sign(msg)
	be.mu.Lock()
	be.failAll = true
	be.mu.Unlock()
	assert.Error(t, h.v.VerifySignature(bft.Signature{ID: 2, Value: sig, Msg: msg}))
	be.mu.Lock()
	be.failAll = false
	be.mu.Unlock()
	assert.NoError(t, h.v.VerifySignature(bft.Signature{ID: 2, Value: sig, Msg: msg}))
}

type failingBackend struct{ cpuBackend } // RegisterKey, SignBatch, Close from the embedded value

func (failingBackend) Verify(Scheme, []Item) ([]bool, error) { return nil, errors.New("no verifier") }

// ADVICE r4 (high): the reference recomputes Proposal.Digest() on every call; a memo must therefore never answer for bytes
// it did not hash.  A caller that decodes the next proposal into the SAME buffers (same slice identity, other contents) must
// get a fresh digest: the vote over the old contents is refused, a vote over the new contents accepted.
func TestDigestMemoIsValidatedAgainstTheBytes(t *testing.T) {
	opt := DefaultOptions
	opt.CacheVerified = false
	h := newHarness(t, 4, opt)
	defer h.v.Close()
	payload := PayloadEncode(nil)
	buf := append([]byte(nil), payload...)
	prop := bft.Proposal{Payload: buf, Header: []byte("h"), Metadata: []byte("m")}
	vote := h.nodes[1].SignProposal(prop, nil)
	_, err := h.v.VerifyConsenterSig(*vote, prop)
	assert.NoError(t, err) // the digest of prop is memoised from here on
	// copy-then-mutate: another Proposal value over its own buffer is a different entry
	q := prop
	q.Payload = append(append([]byte(nil), payload...), 't')
	_, err = h.v.VerifyConsenterSig(*vote, q)
	assert.Error(t, err)
	// in-place rewrite of the decode buffer: same slices, other bytes
	buf[len(buf)-1] ^= 0x40
	_, err = h.v.VerifyConsenterSig(*vote, prop)
	assert.Error(t, err, "a vote over the OLD contents must not be accepted for the rewritten buffer")
	fresh := h.nodes[1].SignProposal(bft.Proposal{Payload: append([]byte(nil), buf...), Header: []byte("h"), Metadata: []byte("m")}, nil)
	_, err = h.v.VerifyConsenterSig(*fresh, prop)
	assert.NoError(t, err, "a vote over the NEW contents is accepted: the digest was recomputed")
	buf[len(buf)-1] ^= 0x40 // and back
	_, err = h.v.VerifyConsenterSig(*vote, prop)
	assert.NoError(t, err)
}

// slowBackend delays every batch: the queue of a busy Verifier is then never empty when its leader looks.
type slowBackend struct {
	Backend
	delay time.Duration
}

func (s slowBackend) Verify(scheme Scheme, items []Item) ([]bool, error) {
	time.Sleep(s.delay)
	return s.Backend.Verify(scheme, items)
}

// ADVICE r4 (medium): the caller that becomes leader returns once ITS job is done; sustained concurrent traffic must not hold it.
func TestLeaderIsNotHeldBySustainedTraffic(t *testing.T) {
	be, err := NewDeviceBackend()
	assert.NoError(t, err)
	opt := DefaultOptions
	opt.CacheVerified = false
	opt.GPUMin = 1 // every single-signature call takes the coalescer (verifyOne)
	v := New(slowBackend{Backend: be, delay: 3 * time.Millisecond}, opt)
	defer v.Close()
	k, _ := ecdsa.GenerateKey(elliptic.P256(), rand.Reader)
	s := &Signer{ID: 1, Key: k}
	v.RegisterConsenter(1, &k.PublicKey)
	msg := []byte("view data")
	sig := bft.Signature{ID: 1, Value: s.Sign(msg), Msg: msg}
	deadline := time.Now().Add(400 * time.Millisecond)
	var wg sync.WaitGroup
	calls := make([]int, 6)
	worst := make([]time.Duration, 6)
	for g := range calls {
		wg.Add(1)
		go func(g int) {
			defer wg.Done()
			for time.Now().Before(deadline) {
				t0 := time.Now()
				assert.NoError(t, v.VerifySignature(sig))
				if d := time.Since(t0); d > worst[g] {
					worst[g] = d
				}
				calls[g]++
			}
		}(g)
	}
	wg.Wait()
	for g := range calls {
		assert.GreaterOrEqual(t, calls[g], 20, "a held leader completes ONE call in the whole run")
		assert.Less(t, worst[g], 60*time.Millisecond)
	}
}
