package gpuverifier

import (
	"crypto/ecdsa"
	"crypto/sha256"
)

// Item is one signature to verify: the signer's key, the signed bytes and the ASN.1 DER signature.
// Slot >= 0 when the key is registered with the device (sbv_p256_register_keys).
type Item struct {
	Pub  *ecdsa.PublicKey
	Slot int32
	Msg  []byte
	Sig  []byte
}

// Backend verifies a batch; ok[i] reports item i.  An error means the BATCH could not be judged (device fault): the
// caller re-verifies it on the CPU and never turns the error into "invalid signature".
type Backend interface {
	Verify(items []Item) (ok []bool, err error)
	// RegisterKey gives the device a key it will see again (consenters, clients); -1 when there is no registry.
	RegisterKey(pub *ecdsa.PublicKey) int32
	Close()
}

// cpuBackend is stock crypto/ecdsa: the reference semantics themselves (ecdsa.VerifyASN1 over SHA-256(Msg)).
type cpuBackend struct{}

func (cpuBackend) Verify(items []Item) ([]bool, error) {
	ok := make([]bool, len(items))
	for i, it := range items {
		h := sha256.Sum256(it.Msg)
		ok[i] = it.Pub != nil && ecdsa.VerifyASN1(it.Pub, h[:], it.Sig)
	}
	return ok, nil
}
func (cpuBackend) RegisterKey(*ecdsa.PublicKey) int32 { return -1 }
func (cpuBackend) Close()                           {}
