// p256_host.h — host-side P-256 helpers for the Signer half of the plugin pair
// (api.Signer, pkg/api/dependencies.go:46-52: Sign / SignProposal; SURVEY.md §8 row a10 "not
// accelerated; CPU").  Built from the product's own field code (consensus_amd/csrc/p256_*.h
// compiled for the host).  Signing is NOT verification: nothing here can accept or reject a
// signature, and libsbv's verify path never calls it.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace sbvhost {

typedef std::string bytes;

void sha256(const void* msg, size_t len, uint8_t out[32]);
bytes sha256(const bytes& msg);
// public key Q = d*G as Qx|Qy (32+32 bytes big-endian); false if d is 0 or >= N
bool pubkey_from_private(const uint8_t d[32], uint8_t q[64]);
// deterministic ECDSA (RFC 6979, HMAC-SHA-256) over a 32-byte digest -> r|s; false on bad key
bool sign_rfc6979(const uint8_t d[32], const uint8_t digest[32], uint8_t rs[64]);
// same with an explicit nonce (test vectors)
bool sign_with_nonce(const uint8_t d[32], const uint8_t k[32], const uint8_t digest[32], uint8_t rs[64]);
// minimal DER ECDSA-Sig-Value, as Go's ecdsa.SignASN1 emits
bytes der_encode_sig(const uint8_t rs[64]);

}  // namespace sbvhost
