package gpuverifier

import (
	"bytes"
	"crypto/ecdsa"
	"crypto/sha256"
	"encoding/binary"
	"errors"
	"runtime"
	"sync"
	"sync/atomic"
	"time"
	"unsafe"

	bft "github.com/hyperledger-labs/SmartBFT/pkg/types"
)

// Options tune the adapter.  Counterpart of consensus_amd/host/verifier.h: struct VerifierOptions.
type Options struct {
	// Scheme: SchemeP256 (default), SchemeEd25519 or SchemeSecp256k1.  One Verifier = one scheme.
	Scheme Scheme
	// GPUMin: batches with fewer signatures are verified with the standard library on the calling goroutines' cores.
	// Through the registered-key route a K = 100 proposal takes 0.27 ms on the device (16 cores need ~0.6 ms); a burst of
	// 15 commit votes 73-75 us through the C++ mirror since round 4 (one launch of the prepared-scalars latency kernel;
	// 15 idle cores: ~100 us, one crypto/ecdsa verification) — DESIGN.md section 5.  The default sends everything from 8
	// signatures on to the device (a burst at N >= 10; the 3 votes of a 4-node cluster stay on the CPU); 0 = always
	// use the backend.
	GPUMin int
	// CoalesceWait / CoalesceMax: the longest the leader of a burst holds the first single-signature call back while it
	// waits for the rest of the burst, and how many calls it ships at once.
	CoalesceWait time.Duration
	CoalesceMax  int
	// CacheVerified keeps verdicts of single-signature calls (commit signatures of sequence s reappear at s+1).
	CacheVerified bool
	// DeviceClientKeys: register client keys with the device (comb slots) as consenter keys always are.  Switch it off
	// for an open client population: request signatures then carry their key inline and the device groups them per batch.
	DeviceClientKeys bool
}

// DefaultOptions are sized for a node with a GPU backend.
var DefaultOptions = Options{GPUMin: 8, CoalesceWait: 50 * time.Microsecond, CoalesceMax: 4096, CacheVerified: true, DeviceClientKeys: true}

// verdict of one coalesced job: a batch nobody could judge (device fault with no CPU verifier for the scheme) is its own
// state — it must reach the caller as an error and must never be cached as "invalid".
type verdict uint8

const (
	verdictInvalid verdict = iota
	verdictValid
	verdictUnknown
)

type job struct {
	item Item
	done chan verdict
}

// Verifier implements api.Verifier and api.RequestInspector.
type Verifier struct {
	opt     Options
	backend Backend
	cpu     Backend // the standard library (cpuBackend); an interface so that tests can take it away
	qmu     sync.Mutex // the queue of pending single-signature jobs and the leader flag
	queue   []*job
	leader  bool  // some caller is serving the queue (serve)
	pending int32 // len(queue), readable without the lock by the collecting leader
	burst   int32 // expected size of a burst of single-signature calls: N - 1 commit votes (internal/bft/view.go:537-541); 0 = unknown

	mu         sync.RWMutex
	consenters map[uint64]regKey
	clients    map[string]regKey
	seq        uint64

	cacheMu sync.Mutex
	cache   map[[32]byte]bool

	digestMu sync.Mutex
	digests  []*digestSlot
}

type regKey struct {
	pub  *ecdsa.PublicKey // SchemeP256
	raw  []byte           // the key bytes: Qx|Qy (P-256, secp256k1) or the 32-byte Ed25519 key
	slot int32
}

type digestSlot struct {
	p     bft.Proposal // the slices the caller handed in: their identity (data pointer + length) is the cheap first filter
	once  sync.Once
	value [32]byte
	// private copies made when the digest was computed: every hit is validated against them, so a decode buffer that was
	// rewritten in place can never answer with the digest of what it held before (ADVICE r4, high)
	payload, header, metadata []byte
}

// New creates a Verifier over the given backend (NewDeviceBackend(), or nil for crypto/ecdsa only).
func New(backend Backend, opt Options) *Verifier {
	if backend == nil {
		backend = cpuBackend{}
	}
	if opt.CoalesceMax <= 0 {
		opt.CoalesceMax = 4096
	}
	return &Verifier{opt: opt, backend: backend, cpu: cpuBackend{}, consenters: map[uint64]regKey{}, clients: map[string]regKey{},
		cache: map[[32]byte]bool{}}
}

// Close releases the backend (there is no goroutine to stop: concurrent calls are merged by the callers themselves, serve).
func (v *Verifier) Close() {
	v.backend.Close()
}

// RegisterConsenter / RegisterClient: the key registry (types.Signature.ID and Request.ClientID select the key).
// P-256 keys are also given to the backend (RegisterKey -> sbv_p256_register_keys): a verification against a registered
// key runs 46 comb additions and no doublings, and the batch can take the raw-messages route (SHA-256 and DER on the device).
func (v *Verifier) RegisterConsenter(id uint64, pub *ecdsa.PublicKey) {
	k := regKey{pub: pub, raw: p256Raw(pub), slot: -1}
	if v.opt.Scheme == SchemeP256 {
		k.slot = v.backend.RegisterKey(pub)
		if k.slot >= 0 {
			v.backend.WidenKey(k.slot) // consenters sign every vote of the epoch: the wide comb halves their u2*Q
		}
	}
	v.mu.Lock()
	v.consenters[id] = k
	n := len(v.consenters)
	v.mu.Unlock()
	if n > 1 {
		atomic.StoreInt32(&v.burst, int32(n-1))
	}
}

func (v *Verifier) RegisterClient(clientID string, pub *ecdsa.PublicKey) {
	k := regKey{pub: pub, raw: p256Raw(pub), slot: -1}
	if v.opt.Scheme == SchemeP256 && v.opt.DeviceClientKeys {
		k.slot = v.backend.RegisterKey(pub)
	}
	v.mu.Lock()
	v.clients[clientID] = k
	v.mu.Unlock()
}

// RegisterConsenterRaw / RegisterClientRaw: the registry for SchemeEd25519 (32-byte keys) and SchemeSecp256k1 (64 bytes
// Qx|Qy big-endian); no device slots for these schemes (the device groups by key inside each batch).
func (v *Verifier) RegisterConsenterRaw(id uint64, key []byte) {
	v.mu.Lock()
	v.consenters[id] = regKey{raw: append([]byte(nil), key...), slot: -1}
	n := len(v.consenters)
	v.mu.Unlock()
	if n > 1 {
		atomic.StoreInt32(&v.burst, int32(n-1))
	}
}

func (v *Verifier) RegisterClientRaw(clientID string, key []byte) {
	v.mu.Lock()
	v.clients[clientID] = regKey{raw: append([]byte(nil), key...), slot: -1}
	v.mu.Unlock()
}

func p256Raw(pub *ecdsa.PublicKey) []byte {
	if pub == nil || pub.X == nil || pub.Y == nil || pub.X.Sign() < 0 || pub.Y.Sign() < 0 || pub.X.BitLen() > 256 || pub.Y.BitLen() > 256 {
		return nil
	}
	kb := make([]byte, 64)
	pub.X.FillBytes(kb[:32])
	pub.Y.FillBytes(kb[32:])
	return kb
}

func (k regKey) item(msg, sig []byte) Item { return Item{Pub: k.pub, Key: k.raw, Slot: k.slot, Msg: msg, Sig: sig} }

// SetVerificationSequence is called by the application when its verification rules change (epoch / config update).
func (v *Verifier) SetVerificationSequence(s uint64) {
	v.mu.Lock()
	v.seq = s
	v.mu.Unlock()
}

func (v *Verifier) consenter(id uint64) (regKey, bool) {
	v.mu.RLock()
	defer v.mu.RUnlock()
	k, ok := v.consenters[id]
	return k, ok
}

func (v *Verifier) client(id string) (regKey, bool) {
	v.mu.RLock()
	defer v.mu.RUnlock()
	k, ok := v.clients[id]
	return k, ok
}

// verifyBatch judges every item: the backend for batches of at least GPUMin, the standard library otherwise — and the
// standard library again whenever the backend fails, so that a device fault can never look like an invalid signature.
// (secp256k1 has no standard-library verifier: when neither side can judge, ok is nil and the callers report
// "could not verify", never "invalid".)
func (v *Verifier) verifyBatch(items []Item) []bool {
	if len(items) >= v.opt.GPUMin {
		if ok, err := v.backend.Verify(v.opt.Scheme, items); err == nil && len(ok) == len(items) {
			return ok
		}
	}
	ok, err := v.cpu.Verify(v.opt.Scheme, items)
	if err != nil {
		if ok2, err2 := v.backend.Verify(v.opt.Scheme, items); err2 == nil && len(ok2) == len(items) {
			return ok2
		}
		return nil
	}
	return ok
}

// serve merges concurrent single-signature calls into one batch WITHOUT a dispatcher goroutine: the first caller that finds
// no leader becomes the leader (it is running already; a parked dispatcher would have to be woken first, and at this scale the
// wake-up is a large part of the round trip — the C++ mirror measured 191 -> 73 us for a burst of 15 votes with this and a
// faster kernel, profiles/r04/m2_trace_r04h.txt).  The leader polls the queue — spin, then yield; no timer: Go's timers fire
// 50-100 us late at this scale, which is the whole budget — until
//
//	the expected burst is in (N - 1 votes: the goroutines of View.processCommits arrive within microseconds of one
//	another, internal/bft/view.go:537-541), or
//	nothing has arrived for a quiet period: CoalesceWait / 6 while the leader is still alone (a lone call — VerifyRequest
//	from HandleRequest, the serial loop of verifyPrevCommitSignatures, internal/bft/view.go:630-644 — is not held back for
//	the whole window), CoalesceWait / 4 otherwise, or
//	CoalesceWait has passed since it took over, or CoalesceMax jobs are queued,
//
// ships the batch and hands the verdicts out.  The CALLER that became leader returns after that first batch (its own job is in
// it); whatever queued up meanwhile is served by a goroutine it starts (at once, no second window), which steps down when it
// finds the queue empty, under the same lock a new job is appended with, so no job is ever left without a leader.
// Counterpart of consensus_amd/host/verifier.cc: Coalescer::submit / serve_as_leader.
func (v *Verifier) serve(stepDown bool) {
	first := stepDown // the continuation goroutine ships what is queued at once: its window has been sat out already
	for {
		if first {
			start := time.Now()
			lastArrival := start
			seen := 1
			hint := int(atomic.LoadInt32(&v.burst))
			for {
				have := int(atomic.LoadInt32(&v.pending))
				if have >= v.opt.CoalesceMax || (hint > 0 && have >= hint) {
					break
				}
				now := time.Now()
				if have != seen {
					seen, lastArrival = have, now
				}
				quiet := v.opt.CoalesceWait / 4
				if have == 1 {
					quiet = v.opt.CoalesceWait / 6
				}
				if now.Sub(start) >= v.opt.CoalesceWait || now.Sub(lastArrival) >= quiet {
					break
				}
				runtime.Gosched()
			}
			first = false
		}
		v.qmu.Lock()
		n := len(v.queue)
		if n > v.opt.CoalesceMax {
			n = v.opt.CoalesceMax
		}
		if n == 0 {
			v.leader = false
			v.qmu.Unlock()
			return
		}
		batch := make([]*job, n)
		copy(batch, v.queue[:n])
		v.queue = append(v.queue[:0], v.queue[n:]...)
		atomic.StoreInt32(&v.pending, int32(len(v.queue)))
		v.qmu.Unlock()
		items := make([]Item, len(batch))
		for i, j := range batch {
			items[i] = j.item
		}
		ok := v.verifyBatch(items)
		for i, j := range batch {
			switch {
			case ok == nil:
				j.done <- verdictUnknown // nobody could judge: never a verdict, never cached (verifyOne)
			case ok[i]:
				j.done <- verdictValid
			default:
				j.done <- verdictInvalid
			}
		}
		if stepDown {
			// The caller that became leader has its verdict (its job was the first of this batch: it found the queue empty
			// when it queued it).  It must not be held here by whatever arrived during the backend call — under sustained
			// VerifyRequest / VerifySignature traffic the queue may never drain, and this may be the consensus goroutine
			// verifying a commit vote (ADVICE r4, medium).  Leadership goes to a goroutine of its own, which keeps serving
			// until the queue is empty; v.leader stays set, so no job is ever without a leader.
			go v.serve(false)
			return
		}
	}
}

// submit queues one job, serves the queue if nobody else does, and waits for the job's verdict.
func (v *Verifier) submit(it Item) verdict {
	j := &job{item: it, done: make(chan verdict, 1)}
	v.qmu.Lock()
	v.queue = append(v.queue, j)
	atomic.StoreInt32(&v.pending, int32(len(v.queue)))
	lead := !v.leader
	if lead {
		v.leader = true
	}
	v.qmu.Unlock()
	if lead {
		v.serve(true) // returns after the first batch, which holds this job: its verdict is in its channel
	}
	return <-j.done
}

// cacheKey must be injective in (key, signature, message): both variable-length fields are length-prefixed.  With a plain
// concatenation a verified (sig, msg) would vouch for (sig + msg[:k], msg[k:]) — a message nobody signed.
func cacheKey(key, msg, sig []byte) [32]byte {
	h := sha256.New()
	var l [8]byte
	binary.LittleEndian.PutUint64(l[:], uint64(len(key)))
	h.Write(l[:])
	h.Write(key)
	binary.LittleEndian.PutUint64(l[:], uint64(len(sig)))
	h.Write(l[:])
	h.Write(sig)
	binary.LittleEndian.PutUint64(l[:], uint64(len(msg)))
	h.Write(l[:])
	h.Write(msg)
	var out [32]byte
	copy(out[:], h.Sum(nil))
	return out
}

func (v *Verifier) verifyOne(k regKey, msg, sig []byte) bool {
	var key [32]byte
	if v.opt.CacheVerified {
		key = cacheKey(k.raw, msg, sig)
		v.cacheMu.Lock()
		ok, hit := v.cache[key]
		v.cacheMu.Unlock()
		if hit {
			return ok
		}
	}
	var ok bool
	if hint := int(atomic.LoadInt32(&v.burst)); v.opt.GPUMin > 1 && hint < v.opt.GPUMin {
		// A burst of single-signature calls can never reach GPUMin (it is at most N - 1 votes): verify right here, on the
		// calling goroutine's core — the reference's goroutines already are the parallelism (internal/bft/view.go:537-541).
		verdict := v.verifyBatch([]Item{k.item(msg, sig)})
		if verdict == nil {
			return false // nobody could judge (secp256k1 without a device): an error for the caller, and nothing to remember
		}
		ok = verdict[0]
	} else {
		switch v.submit(k.item(msg, sig)) {
		case verdictUnknown:
			return false // as on the direct path: a device fault is an error for the caller and must not be remembered as "invalid"
		case verdictValid:
			ok = true
		}
	}
	if v.opt.CacheVerified {
		v.cacheMu.Lock()
		if len(v.cache) > 1<<20 {
			v.cache = map[[32]byte]bool{}
		}
		v.cache[key] = ok
		v.cacheMu.Unlock()
	}
	return ok
}

// digest returns SHA-256 of the proposal's ASN.1 form, computed once per proposal even when the first callers arrive
// together (the reference recomputes it three times per sequence: internal/bft/view.go:435, 443, 524).  An entry is FOUND by
// the identity of the proposal's byte slices (data pointer + length of Payload, Header, Metadata, and the sequence): the
// reference hands the same v.inFlightProposal to VerifyProposal (view.go:555) and to every VerifyConsenterSig of the
// sequence (view.go:834).  It is TRUSTED only after its private copies of the three fields compare equal to what the caller
// holds now (three memcmp: ~1 us for a 100-request proposal, ~0.1 ms for a 10 000-request one — against ~1 ms of ASN.1 +
// SHA-256): identity alone is not enough, a caller may decode the next proposal into the same buffer, and the reference
// recomputes Digest() on every call.  An entry whose copies no longer match is dropped and the new contents are hashed.
// The digest is computed FROM the copies, so the two can never disagree.  Counterpart of consensus_amd/host/formats.h
// (there the fields are private and every mutator drops the memo).
func (v *Verifier) digest(p bft.Proposal) [32]byte {
	same := func(a, b []byte) bool { return len(a) == len(b) && unsafe.SliceData(a) == unsafe.SliceData(b) }
	for attempt := 0; attempt < 3; attempt++ {
		v.digestMu.Lock()
		var s *digestSlot
		for _, d := range v.digests {
			if d.p.VerificationSequence == p.VerificationSequence && same(d.p.Payload, p.Payload) && same(d.p.Header, p.Header) &&
				same(d.p.Metadata, p.Metadata) {
				s = d
				break
			}
		}
		if s == nil {
			s = &digestSlot{p: p}
			if len(v.digests) >= 4 {
				v.digests = v.digests[1:]
			}
			v.digests = append(v.digests, s)
		}
		v.digestMu.Unlock()
		s.once.Do(func() {
			s.payload = append([]byte(nil), p.Payload...)
			s.header = append([]byte(nil), p.Header...)
			s.metadata = append([]byte(nil), p.Metadata...)
			s.value = proposalDigestRaw(bft.Proposal{Payload: s.payload, Header: s.header, Metadata: s.metadata,
				VerificationSequence: p.VerificationSequence})
		})
		if bytes.Equal(s.payload, p.Payload) && bytes.Equal(s.header, p.Header) && bytes.Equal(s.metadata, p.Metadata) {
			return s.value
		}
		// same slices, other bytes: the caller's buffers were rewritten since the entry was made
		v.digestMu.Lock()
		for i, d := range v.digests {
			if d == s {
				v.digests = append(v.digests[:i:i], v.digests[i+1:]...)
				break
			}
		}
		v.digestMu.Unlock()
	}
	// buffers that keep changing under the call: no memo, hash a private snapshot
	return proposalDigestRaw(bft.Proposal{Payload: append([]byte(nil), p.Payload...), Header: append([]byte(nil), p.Header...),
		Metadata: append([]byte(nil), p.Metadata...), VerificationSequence: p.VerificationSequence})
}

// ---- api.Verifier ------------------------------------------------------------------------------------------------------

var (
	errInvalidSig = errors.New("invalid signature")
	errUnknown    = errors.New("unknown signer")
)

// VerifySignature: internal/bft/viewchanger.go:598, 660, 983, 1022, 1076.
func (v *Verifier) VerifySignature(s bft.Signature) error {
	k, ok := v.consenter(s.ID)
	if !ok {
		return errUnknown
	}
	if !v.verifyOne(k, s.Msg, s.Value) {
		return errInvalidSig
	}
	return nil
}

// VerifyConsenterSig: internal/bft/view.go:631, 834; internal/bft/viewchanger.go:681-727.
func (v *Verifier) VerifyConsenterSig(s bft.Signature, prop bft.Proposal) ([]byte, error) {
	bound, aux, ok := ConsenterMsgSplit(s.Msg)
	if !ok {
		return nil, errors.New("malformed signature message")
	}
	if bound != v.digest(prop) {
		return nil, errors.New("signature message does not match proposal")
	}
	if err := v.VerifySignature(s); err != nil {
		return nil, err
	}
	return aux, nil
}

// AuxiliaryData: internal/bft/view.go:1029, 1071 — no verification.
func (v *Verifier) AuxiliaryData(msg []byte) []byte {
	_, aux, _ := ConsenterMsgSplit(msg)
	return aux
}

// RequestID implements api.RequestInspector (internal/bft/requestpool.go:192).
func (v *Verifier) RequestID(raw []byte) bft.RequestInfo {
	r, err := ParseRequest(raw)
	if err != nil {
		return bft.RequestInfo{}
	}
	return bft.RequestInfo{ClientID: r.ClientID, ID: r.ID}
}

// VerifyRequest: internal/bft/controller.go:233-246 (leader, before pooling) and :733-746 (Pool.Prune).
func (v *Verifier) VerifyRequest(raw []byte) (bft.RequestInfo, error) {
	r, err := ParseRequest(raw)
	if err != nil {
		return bft.RequestInfo{}, err
	}
	k, ok := v.client(r.ClientID)
	if !ok {
		return bft.RequestInfo{}, errors.New("unknown client")
	}
	if !v.verifyOne(k, r.Signed, r.Sig) {
		return bft.RequestInfo{}, errors.New("invalid request signature")
	}
	return bft.RequestInfo{ClientID: r.ClientID, ID: r.ID}, nil
}

func (v *Verifier) parseProposal(p bft.Proposal) ([]*Request, []bft.RequestInfo, error) {
	raws, err := PayloadSplit(p.Payload)
	if err != nil {
		return nil, nil, err
	}
	reqs := make([]*Request, len(raws))
	infos := make([]bft.RequestInfo, len(raws))
	for i, raw := range raws {
		r, err := ParseRequest(raw)
		if err != nil {
			return nil, nil, err
		}
		reqs[i] = r
		infos[i] = bft.RequestInfo{ClientID: r.ClientID, ID: r.ID}
	}
	return reqs, infos, nil
}

// VerifyProposal: internal/bft/view.go:553-559 — all K request signatures as ONE batch.
func (v *Verifier) VerifyProposal(p bft.Proposal) ([]bft.RequestInfo, error) {
	v.mu.RLock()
	seq := v.seq
	v.mu.RUnlock()
	if uint64(p.VerificationSequence) != seq {
		return nil, errors.New("verification sequence mismatch")
	}
	reqs, infos, err := v.parseProposal(p)
	if err != nil {
		return nil, err
	}
	// Proposal.Digest() starts now, beside the backend call below and the prepare round after it: the first commit vote on
	// this proposal finds it ready instead of marshalling and hashing the whole payload (1.7 MB at K = 10 000).
	go v.digest(p)
	items := make([]Item, len(reqs))
	for i, r := range reqs {
		k, ok := v.client(r.ClientID)
		if !ok {
			return nil, errors.New("unknown client")
		}
		items[i] = k.item(r.Signed, r.Sig)
	}
	verdicts := v.verifyBatch(items)
	if verdicts == nil && len(items) > 0 {
		return nil, errors.New("could not verify the proposal's request signatures (no backend for this scheme)")
	}
	for _, ok := range verdicts {
		if !ok {
			return nil, errors.New("invalid request signature in proposal")
		}
	}
	return infos, nil
}

// RequestsFromProposal: internal/bft/view.go:395, 419; internal/bft/controller.go:892-898 — parsing only, and the same
// list VerifyProposal returns (the pool evicts delivered requests by it).
func (v *Verifier) RequestsFromProposal(p bft.Proposal) []bft.RequestInfo {
	_, infos, err := v.parseProposal(p)
	if err != nil {
		return nil
	}
	return infos
}

// VerificationSequence: internal/bft/view.go:614-618.
func (v *Verifier) VerificationSequence() uint64 {
	v.mu.RLock()
	defer v.mu.RUnlock()
	return v.seq
}

// VerifyDecisions is the batch form for a Synchronizer that replays decisions (pkg/types/types.go:31-34): every
// signature of every decision in one backend call; decided[i] reports whether decision i carries at least quorum valid
// signatures by distinct consenters.
func (v *Verifier) VerifyDecisions(decisions []bft.Decision, quorum int) (decided []bool) {
	var items []Item
	type ref struct{ d int; id uint64 }
	var refs []ref
	for di, d := range decisions {
		dg := v.digest(d.Proposal)
		for _, s := range d.Signatures {
			bound, _, ok := ConsenterMsgSplit(s.Msg)
			k, known := v.consenter(s.ID)
			if !ok || !known || bound != dg {
				continue
			}
			items = append(items, k.item(s.Msg, s.Value))
			refs = append(refs, ref{di, s.ID})
		}
	}
	ok := v.verifyBatch(items)
	seen := make([]map[uint64]bool, len(decisions))
	for i, r := range refs {
		if ok != nil && ok[i] {
			if seen[r.d] == nil {
				seen[r.d] = map[uint64]bool{}
			}
			seen[r.d][r.id] = true
		}
	}
	decided = make([]bool, len(decisions))
	for i := range decisions {
		decided[i] = len(seen[i]) >= quorum
	}
	return decided
}
