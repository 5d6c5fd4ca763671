// ed25519_host.h — host-side Ed25519 helpers for the Signer half of the plugin pair in the Ed25519
// Verifier variant (BASELINE.json configs[4]; api.Signer, pkg/api/dependencies.go:46-52).  RFC 8032 key
// expansion and signing, built from the product's own field code (consensus_amd/csrc/ed25519_*.h
// compiled for the host).  Signing is NOT verification: nothing here can accept or reject a signature.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace sbvhost {

void sha512(const void* msg, size_t len, uint8_t out[64]);
// A = [a]B for a = clamp(SHA-512(seed)[0..32)), RFC 8032 §5.1.5
void ed25519_public_key(const uint8_t seed[32], uint8_t a_enc[32]);
// sig = R | S, RFC 8032 §5.1.6 (deterministic)
void ed25519_sign(const uint8_t seed[32], const void* msg, size_t len, uint8_t sig[64]);
// k = SHA-512(R | A | msg) mod L, little-endian (the tuple's fourth field, include/sbv.h)
void ed25519_hram(const uint8_t r_enc[32], const uint8_t a_enc[32], const void* msg, size_t len, uint8_t k[32]);

}  // namespace sbvhost
