package gpuverifier

import (
	"crypto/ecdsa"
	"crypto/ed25519"
	"crypto/sha256"
	"errors"
	"runtime"
	"sync"
)

// Scheme is the signature scheme of a Verifier / Signer pair (consensus_amd/host/verifier.h: enum class Scheme).
//
//	SchemeP256       ECDSA over SHA-256, ASN.1 DER signatures, crypto/ecdsa.VerifyASN1 semantics (the default)
//	SchemeEd25519    crypto/ed25519.Verify semantics, 64-byte R|S signatures, 32-byte keys (BASELINE.json configs[4])
//	SchemeSecp256k1  ECDSA over SHA-256 with DER signatures like P-256, on y^2 = x^3 + 7; keys are 64 raw bytes Qx|Qy.
//	                 Go's standard library has no secp256k1: without a device backend such a batch cannot be judged.
type Scheme int

const (
	SchemeP256 Scheme = iota
	SchemeEd25519
	SchemeSecp256k1
)

// Item is one signature to verify: the signer's key, the signed bytes and the signature.
//
//	Pub   the P-256 key (SchemeP256)
//	Key   the raw key bytes of the other schemes: 32-byte Ed25519 key, or 64 bytes Qx|Qy big-endian for secp256k1
//	Slot  >= 0 when the key is registered with the device (sbv_p256_register_keys); -1 otherwise
type Item struct {
	Pub  *ecdsa.PublicKey
	Key  []byte
	Slot int32
	Msg  []byte
	Sig  []byte
}

// ErrNoCPUPath is returned by the pure-Go backend for a scheme the standard library cannot verify.
var ErrNoCPUPath = errors.New("gpuverifier: no CPU implementation of this scheme in the standard library")

// Backend verifies a batch; ok[i] reports item i.  An error means the BATCH could not be judged (device fault): the
// caller re-verifies it on the CPU and never turns the error into "invalid signature".
//
// File-by-file counterpart of consensus_amd/host/verifier.h: class Backend (verify / verify_keyed / verify_msgs_keyed /
// verify_ed25519 / verify_k256 are folded into Verify here: the route is chosen inside the backend from the items' slots).
type Backend interface {
	Verify(scheme Scheme, items []Item) (ok []bool, err error)
	// RegisterKey gives the device a P-256 key it will see again (consenters, clients) and returns its comb slot;
	// -1 when there is no registry (then items carry Slot = -1 and travel as generic tuples with the key inline).
	RegisterKey(pub *ecdsa.PublicKey) int32
	// WidenKey marks a registered slot as a consenter's: a key that signs every vote of every decision of the epoch
	// (internal/bft/view.go:531-541, 631, 834).  The device gives it a second, wide comb (sbv_p256_widen_keys: built on the
	// device in milliseconds; 20-bit windows = 436 MB of HBM and u2*Q in 13 additions instead of 32 while at most 16 slots are
	// wide, 16-bit windows beyond); a no-op for a backend without a registry.
	WidenKey(slot int32)
	// SignBatch is the batch form of api.Signer.Sign for P-256 (sbv_p256_sign_batch: RFC 6979 nonces): signature i =
	// ECDSA(keys[keyIndex[i]], digests[i]) as r|s, 64 bytes; ok[i] = false when the key or index is unusable.
	// A backend without batch signing returns ErrNoBatchSigner and the Signer signs one by one with crypto/ecdsa.
	SignBatch(keys [][32]byte, keyIndex []uint32, digests [][32]byte) (sigs [][64]byte, ok []bool, err error)
	Close()
}

// ErrNoBatchSigner: the backend has no batch signing entry (the pure-Go backend).
var ErrNoBatchSigner = errors.New("gpuverifier: backend has no batch signer")

// cpuBackend is the standard library: the reference semantics themselves (ecdsa.VerifyASN1 over SHA-256(Msg),
// ed25519.Verify).
type cpuBackend struct{}

func (cpuBackend) Verify(scheme Scheme, items []Item) ([]bool, error) {
	if scheme != SchemeP256 && scheme != SchemeEd25519 {
		return nil, ErrNoCPUPath
	}
	ok := make([]bool, len(items))
	one := func(i int) {
		it := &items[i]
		if scheme == SchemeP256 {
			h := sha256.Sum256(it.Msg)
			ok[i] = it.Pub != nil && ecdsa.VerifyASN1(it.Pub, h[:], it.Sig)
		} else {
			ok[i] = len(it.Key) == ed25519.PublicKeySize && ed25519.Verify(ed25519.PublicKey(it.Key), it.Msg, it.Sig)
		}
	}
	workers := runtime.GOMAXPROCS(0)
	if workers > len(items) {
		workers = len(items)
	}
	if workers <= 1 {
		for i := range items {
			one(i)
		}
		return ok, nil
	}
	var wg sync.WaitGroup // a batch on the CPU route (a proposal below GPUMin, a device fault) uses every core
	for w := 0; w < workers; w++ {
		wg.Add(1)
		go func(w int) {
			defer wg.Done()
			for i := w; i < len(items); i += workers {
				one(i)
			}
		}(w)
	}
	wg.Wait()
	return ok, nil
}
func (cpuBackend) RegisterKey(*ecdsa.PublicKey) int32 { return -1 }
func (cpuBackend) WidenKey(int32)                      {}
func (cpuBackend) SignBatch([][32]byte, []uint32, [][32]byte) ([][64]byte, []bool, error) {
	return nil, nil, ErrNoBatchSigner
}
func (cpuBackend) Close() {}
