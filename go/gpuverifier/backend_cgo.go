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

const tupleBytes = 160

// gpuBackend talks to libsbv.so.  All device work is funnelled through ONE goroutine locked to its OS thread
// (HIP binds devices per thread; a blocked cgo call pins an OS thread anyway), fed by a channel.
type gpuBackend struct {
	jobs chan *gpuJob
	mu   sync.Mutex
	regs map[string]int32
}

type gpuJob struct {
	items []Item
	ok    []bool
	err   error
	done  chan struct{}
}

// NewDeviceBackend initialises every visible MI355X (sbv_init_all).
func NewDeviceBackend() (Backend, error) {
	b := &gpuBackend{jobs: make(chan *gpuJob, 64), regs: map[string]int32{}}
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
		j.ok, j.err = b.verify(j.items)
		close(j.done)
	}
	C.sbv_shutdown()
}

func keyBytes(pub *ecdsa.PublicKey) [64]byte {
	var k [64]byte
	pub.X.FillBytes(k[0:32]) // keys are validated at registration (non-negative, <= 256 bits)
	pub.Y.FillBytes(k[32:64])
	return k
}

// verify runs on the device goroutine.  Flat []byte buffers only cross the boundary (cgo pointer rules); the C side
// has copied them to the device when the call returns.
func (b *gpuBackend) verify(items []Item) ([]bool, error) {
	n := len(items)
	if n == 0 {
		return nil, nil
	}
	buf := make([]byte, n*tupleBytes)
	for i, it := range items {
		t := buf[i*tupleBytes : (i+1)*tupleBytes]
		if len(it.Sig) > 0 { // strict DER (cryptobyte rules); on failure r = s = 0, which every verify rejects
			C.sbv_p256_parse_der((*C.uint8_t)(unsafe.Pointer(&it.Sig[0])), C.size_t(len(it.Sig)), (*C.uint8_t)(unsafe.Pointer(&t[0])))
		}
		h := sha256.Sum256(it.Msg)
		copy(t[64:96], h[:])
		if it.Pub != nil {
			k := keyBytes(it.Pub)
			copy(t[96:160], k[:])
		}
	}
	bitmap := make([]byte, (n+7)/8)
	// all GPUs of the node; batches below 2 x 2^18 signatures go whole to one device, round-robin
	rc := C.sbv_p256_verify_batch_sharded((*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(n), 0, 0,
		(*C.uint8_t)(unsafe.Pointer(&bitmap[0])), nil, nil)
	if rc != 0 {
		return nil, lastError()
	}
	ok := make([]bool, n)
	for i := range ok {
		ok[i] = bitmap[i>>3]>>(uint(i)&7)&1 == 1
	}
	return ok, nil
}

func (b *gpuBackend) Verify(items []Item) ([]bool, error) {
	j := &gpuJob{items: items, done: make(chan struct{})}
	b.jobs <- j
	<-j.done
	return j.ok, j.err
}

// RegisterKey: in-step key grouping on the device already exploits repeated keys inside a batch, so the generic entry
// above needs no slots; the registry is kept for callers that use sbv_p256_verify_batch_keyed directly.
func (b *gpuBackend) RegisterKey(pub *ecdsa.PublicKey) int32 { return -1 }

func (b *gpuBackend) Close() { close(b.jobs) }
