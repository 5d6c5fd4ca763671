// k256_host.h — host-side secp256k1 key derivation and RFC 6979 signing for Scheme::SECP256K1 (verifier.h).
#pragma once
#include <stdint.h>

namespace sbvhost {

bool k256_pubkey_from_private(const uint8_t d_be[32], uint8_t q[64]);      // false: d = 0 or d >= n
bool k256_sign_with_nonce(const uint8_t d_be[32], const uint8_t k_be[32], const uint8_t digest[32], uint8_t rs[64]);
bool k256_sign_rfc6979(const uint8_t d_be[32], const uint8_t digest[32], uint8_t rs[64]);

}  // namespace sbvhost
