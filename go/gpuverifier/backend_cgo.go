//go:build sbvgpu

package gpuverifier

/*
#cgo LDFLAGS: -lsbv
#include <stdlib.h>
#include "sbv.h"
*/
import "C"

import (
	"crypto/ecdsa"
	"crypto/sha256"
	"errors"
	"runtime"
	"sync"
	"unsafe"
)

const (
	tupleBytes   = 160     // r | s | hash | Qx | Qy   (include/sbv.h)
	rshBytes     = 96      // r | s | hash             (registered-key form)
	frontEndMax  = 1 << 21 // sbv_p256_verify_msgs_keyed / sbv_ed25519_verify_msgs take one chunk of at most 2^21
	frontEndFrom = 64      // below this many signatures the host hashes and parses itself (one upload instead of five)
	shardedFrom  = 32768   // above this many registered-key signatures: every GPU of the node, uploads in pieces beside the kernels
)

// gpuBackend talks to libsbv.so.  All device work is funnelled through ONE goroutine locked to its OS thread
// (HIP binds devices per thread; a blocked cgo call pins an OS thread anyway), fed by a channel.
//
// Counterpart of consensus_amd/host/verifier.cc: class SbvBackend — the same four routes:
//
//	every item has a slot, n > shardedFrom     sbv_p256_verify_msgs_keyed_sharded  the same inputs over every GPU of the node (decision replay:
//	                                           50 000 decisions x 11 commit signatures, internal/bft/controller.go:587-633), in pieces
//	every item has a slot, n >= frontEndFrom   sbv_p256_verify_msgs_keyed   raw messages + DER + slots (SHA-256 and DER on the device)
//	every item has a slot                      sbv_p256_verify_batch_keyed  96-byte r|s|hash records + slots
//	otherwise                                  sbv_p256_verify_batch_sharded generic tuples, all GPUs of the node
//	SchemeEd25519 / SchemeSecp256k1            sbv_ed25519_verify_msgs / sbv_secp256k1_verify_batch
type gpuBackend struct {
	jobs chan *gpuJob
	mu   sync.Mutex
	regs map[[64]byte]int32
}

type gpuJob struct {
	run  func()
	done chan struct{}
}

// NewDeviceBackend initialises every visible MI355X (sbv_init_all).
func NewDeviceBackend() (Backend, error) {
	b := &gpuBackend{jobs: make(chan *gpuJob, 64), regs: map[[64]byte]int32{}}
	ready := make(chan error, 1)
	go b.loop(ready)
	if err := <-ready; err != nil {
		return nil, err
	}
	return b, nil
}

func lastError() error { return errors.New("libsbv: " + C.GoString(C.sbv_last_error())) }

func (b *gpuBackend) loop(ready chan<- error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if n := C.sbv_init_all(); n <= 0 {
		ready <- lastError()
		return
	}
	ready <- nil
	for j := range b.jobs {
		j.run()
		close(j.done)
	}
	C.sbv_shutdown()
}

// on runs f on the device goroutine and waits for it.
func (b *gpuBackend) on(f func()) {
	j := &gpuJob{run: f, done: make(chan struct{})}
	b.jobs <- j
	<-j.done
}

func keyBytes(pub *ecdsa.PublicKey) [64]byte {
	var k [64]byte
	pub.X.FillBytes(k[0:32]) // keys are validated at registration (non-negative, <= 256 bits)
	pub.Y.FillBytes(k[32:64])
	return k
}

func bitmapToBools(bitmap []byte, n int) []bool {
	ok := make([]bool, n)
	for i := range ok {
		ok[i] = bitmap[i>>3]>>(uint(i)&7)&1 == 1
	}
	return ok
}

func u8(p []byte) *C.uint8_t { return (*C.uint8_t)(unsafe.Pointer(&p[0])) }

// fillRSH writes r | s | hash of one item at t[0:96]: strict DER (cryptobyte rules; on failure r = s = 0, which every
// verify rejects) and SHA-256 of the message.
func fillRSH(t []byte, it *Item) {
	if len(it.Sig) > 0 {
		C.sbv_p256_parse_der(u8(it.Sig), C.size_t(len(it.Sig)), u8(t))
	}
	h := sha256.Sum256(it.Msg)
	copy(t[64:96], h[:])
}

func allSlotted(items []Item) bool {
	for i := range items {
		if items[i].Slot < 0 {
			return false
		}
	}
	return true
}

// packMsgs lays variable-length byte strings back to back with an offset table (n + 1 entries), as the device front
// ends take them.
func packMsgs(get func(i int) []byte, n int) ([]byte, []uint64) {
	off := make([]uint64, n+1)
	total := 0
	for i := 0; i < n; i++ {
		off[i] = uint64(total)
		total += len(get(i))
	}
	off[n] = uint64(total)
	buf := make([]byte, total+1) // never empty: &buf[0] must exist
	for i := 0; i < n; i++ {
		copy(buf[off[i]:], get(i))
	}
	return buf, off
}

// verifyP256 runs on the device goroutine.  Flat []byte / []uint32 / []uint64 buffers only cross the boundary (cgo
// pointer rules); the C side has copied them to the device when the call returns.
func (b *gpuBackend) verifyP256(items []Item) ([]bool, error) {
	n := len(items)
	bitmap := make([]byte, (n+7)/8)
	if allSlotted(items) {
		slots := make([]uint32, n)
		for i := range items {
			slots[i] = uint32(items[i].Slot)
		}
		if n > shardedFrom {
			// the whole batch in one call: the library splits it by device and by piece (every device holds a replica of the key
			// registry and of the consenters' wide combs; no 2^21 limit on this entry)
			msgs, moff := packMsgs(func(i int) []byte { return items[i].Msg }, n)
			sigs, soff := packMsgs(func(i int) []byte { return items[i].Sig }, n)
			rc := C.sbv_p256_verify_msgs_keyed_sharded(u8(msgs), (*C.uint64_t)(unsafe.Pointer(&moff[0])), u8(sigs),
				(*C.uint64_t)(unsafe.Pointer(&soff[0])), (*C.uint32_t)(unsafe.Pointer(&slots[0])), C.size_t(n), 0, 0, u8(bitmap), nil, nil)
			if rc != 0 {
				return nil, lastError()
			}
			return bitmapToBools(bitmap, n), nil
		}
		if n >= frontEndFrom {
			for lo := 0; lo < n; lo += frontEndMax {
				hi := lo + frontEndMax
				if hi > n {
					hi = n
				}
				m := hi - lo
				msgs, moff := packMsgs(func(i int) []byte { return items[lo+i].Msg }, m)
				sigs, soff := packMsgs(func(i int) []byte { return items[lo+i].Sig }, m)
				rc := C.sbv_p256_verify_msgs_keyed(u8(msgs), (*C.uint64_t)(unsafe.Pointer(&moff[0])), u8(sigs),
					(*C.uint64_t)(unsafe.Pointer(&soff[0])), (*C.uint32_t)(unsafe.Pointer(&slots[lo])), C.size_t(m),
					u8(bitmap[lo/8:])) // lo is a multiple of 2^21: whole bitmap bytes
				if rc != 0 {
					return nil, lastError()
				}
			}
			return bitmapToBools(bitmap, n), nil
		}
		rsh := make([]byte, n*rshBytes)
		for i := range items {
			fillRSH(rsh[i*rshBytes:(i+1)*rshBytes], &items[i])
		}
		if rc := C.sbv_p256_verify_batch_keyed(u8(rsh), (*C.uint32_t)(unsafe.Pointer(&slots[0])), C.size_t(n), u8(bitmap)); rc != 0 {
			return nil, lastError()
		}
		return bitmapToBools(bitmap, n), nil
	}
	buf := make([]byte, n*tupleBytes)
	for i := range items {
		t := buf[i*tupleBytes : (i+1)*tupleBytes]
		fillRSH(t, &items[i])
		// Only the VALIDATED raw key (RegisterKey / p256Raw refuse nil or > 256-bit coordinates and leave Key empty) ever
		// reaches the tuple: big.Int.FillBytes panics on an oversized coordinate, and this runs on the device goroutine.
		// Without one the key field stays zero: (0, 0) is not on the curve, so the kernel rejects the signature, which is
		// what crypto/ecdsa does with such a key.
		if len(items[i].Key) == 64 {
			copy(t[96:160], items[i].Key)
		}
	}
	// all GPUs of the node; a batch that is not worth splitting goes whole to one device, round-robin
	rc := C.sbv_p256_verify_batch_sharded(u8(buf), C.size_t(n), 0, 0, u8(bitmap), nil, nil)
	if rc != 0 {
		return nil, lastError()
	}
	return bitmapToBools(bitmap, n), nil
}

// verifyEd25519: signatures n x 64, keys n x 32, messages packed; SHA-512 and the reduction mod L run on the device.
// A signature or key of the wrong length is this layer's reject (crypto/ed25519.Verify returns false / panics on them).
func (b *gpuBackend) verifyEd25519(items []Item) ([]bool, error) {
	n := len(items)
	ok := make([]bool, n)
	idx := make([]int, 0, n)
	for i := range items {
		if len(items[i].Sig) == 64 && len(items[i].Key) == 32 {
			idx = append(idx, i)
		}
	}
	for lo := 0; lo < len(idx); lo += frontEndMax {
		hi := lo + frontEndMax
		if hi > len(idx) {
			hi = len(idx)
		}
		m := hi - lo
		sigs := make([]byte, 64*m)
		pks := make([]byte, 32*m)
		for k := 0; k < m; k++ {
			copy(sigs[64*k:], items[idx[lo+k]].Sig)
			copy(pks[32*k:], items[idx[lo+k]].Key)
		}
		msgs, moff := packMsgs(func(k int) []byte { return items[idx[lo+k]].Msg }, m)
		bitmap := make([]byte, (m+7)/8)
		rc := C.sbv_ed25519_verify_msgs(u8(sigs), u8(pks), u8(msgs), (*C.uint64_t)(unsafe.Pointer(&moff[0])), C.size_t(m), u8(bitmap))
		if rc != 0 {
			return nil, lastError()
		}
		for k := 0; k < m; k++ {
			ok[idx[lo+k]] = bitmap[k>>3]>>(uint(k)&7)&1 == 1
		}
	}
	return ok, nil
}

// verifySecp256k1: generic tuples with the 64-byte key inline (no registered keys for this curve yet).
func (b *gpuBackend) verifySecp256k1(items []Item) ([]bool, error) {
	n := len(items)
	buf := make([]byte, n*tupleBytes)
	for i := range items {
		t := buf[i*tupleBytes : (i+1)*tupleBytes]
		fillRSH(t, &items[i])
		if len(items[i].Key) == 64 {
			copy(t[96:160], items[i].Key)
		} // else: the all-zero key, which is not on the curve
	}
	bitmap := make([]byte, (n+7)/8)
	if rc := C.sbv_secp256k1_verify_batch(u8(buf), C.size_t(n), u8(bitmap)); rc != 0 {
		return nil, lastError()
	}
	return bitmapToBools(bitmap, n), nil
}

func (b *gpuBackend) Verify(scheme Scheme, items []Item) (ok []bool, err error) {
	if len(items) == 0 {
		return nil, nil
	}
	b.on(func() {
		switch scheme {
		case SchemeEd25519:
			ok, err = b.verifyEd25519(items)
		case SchemeSecp256k1:
			ok, err = b.verifySecp256k1(items)
		default:
			ok, err = b.verifyP256(items)
		}
	})
	return ok, err
}

// RegisterKey builds the key's comb on the device once (sbv_p256_register_keys: 270 KiB of HBM per key) and remembers
// the slot; registering a key again returns the same slot.  -1 when the device refuses (then the key travels inline).
func (b *gpuBackend) RegisterKey(pub *ecdsa.PublicKey) int32 {
	if pub == nil || pub.X == nil || pub.Y == nil || pub.X.Sign() < 0 || pub.Y.Sign() < 0 || pub.X.BitLen() > 256 || pub.Y.BitLen() > 256 {
		return -1
	}
	k := keyBytes(pub)
	b.mu.Lock()
	if s, hit := b.regs[k]; hit {
		b.mu.Unlock()
		return s
	}
	b.mu.Unlock()
	slot := int32(-1)
	b.on(func() {
		var out C.uint32_t
		if rc := C.sbv_p256_register_keys(u8(k[:]), 1, &out); rc == 0 {
			slot = int32(out)
		}
	})
	if slot >= 0 {
		b.mu.Lock()
		b.regs[k] = slot
		b.mu.Unlock()
	}
	return slot
}

// WidenKey: sbv_p256_widen_keys for one slot (best effort: a slot that gets no wide comb keeps its 8-bit one, verdicts are the same).
func (b *gpuBackend) WidenKey(slot int32) {
	if slot < 0 {
		return
	}
	b.on(func() {
		var s C.uint32_t
		s = C.uint32_t(slot)
		C.sbv_p256_widen_keys(&s, C.size_t(1))
	})
}

// PoolStats is what the grouped P-256 step really holds on the default device (sbv_p256_pool_stats; round 6).  The pools are sized for a
// 288 GB MI355X; a shared or smaller device gets halved pools instead of an error, and a batch whose pools cannot be allocated at all is
// verified by the one-lane kernel — an operator reads here which of the two happened (NomemFallbacks > 0: give the replica more HBM).
type PoolStats struct {
	CacheKeys, GroupsPerBatch uint32 // 0 before the first grouped batch
	Shrunk                    bool   // smaller than asked for
	NomemFallbacks            uint32 // grouped batches that took the one-lane kernel for lack of memory
	HotPool                   uint32 // 16-bit combs of the hot-key pool
	GPUShare                  uint32 // contexts sharing the GPU (SBV_LOGICAL_DEVICES)
}

// PoolStats: sbv_p256_pool_stats.
func (b *gpuBackend) PoolStats() (st PoolStats, err error) {
	b.on(func() {
		var out [6]C.uint32_t
		if rc := C.sbv_p256_pool_stats((*C.uint32_t)(unsafe.Pointer(&out[0]))); rc != 0 {
			err = lastError()
			return
		}
		st = PoolStats{CacheKeys: uint32(out[0]), GroupsPerBatch: uint32(out[1]), Shrunk: out[2] != 0, NomemFallbacks: uint32(out[3]),
			HotPool: uint32(out[4]), GPUShare: uint32(out[5])}
	})
	return st, err
}

// SignBatch: sbv_p256_sign_batch (RFC 6979 nonces on the device; not constant-time — see include/sbv.h).
func (b *gpuBackend) SignBatch(keys [][32]byte, keyIndex []uint32, digests [][32]byte) (sigs [][64]byte, ok []bool, err error) {
	n := len(digests)
	if n == 0 || len(keys) == 0 || len(keyIndex) != n {
		return nil, nil, errors.New("gpuverifier: SignBatch needs keys and one key index per digest")
	}
	kb := make([]byte, 32*len(keys))
	for i := range keys {
		copy(kb[32*i:], keys[i][:])
	}
	db := make([]byte, 32*n)
	for i := range digests {
		copy(db[32*i:], digests[i][:])
	}
	out := make([]byte, 64*n)
	flags := make([]byte, n)
	b.on(func() {
		rc := C.sbv_p256_sign_batch(u8(kb), C.uint32_t(len(keys)), (*C.uint32_t)(unsafe.Pointer(&keyIndex[0])), u8(db), C.size_t(n), u8(out), u8(flags))
		if rc != 0 {
			err = lastError()
		}
	})
	for i := range kb {
		kb[i] = 0
	}
	if err != nil {
		return nil, nil, err
	}
	sigs = make([][64]byte, n)
	ok = make([]bool, n)
	for i := 0; i < n; i++ {
		copy(sigs[i][:], out[64*i:64*i+64])
		ok[i] = flags[i] != 0
	}
	return sigs, ok, nil
}

func (b *gpuBackend) Close() { close(b.jobs) }
