//go:build !sbvgpu

package gpuverifier

// NewDeviceBackend without the sbvgpu build tag: there is no device; every batch takes the crypto/ecdsa route.
func NewDeviceBackend() (Backend, error) { return cpuBackend{}, nil }
