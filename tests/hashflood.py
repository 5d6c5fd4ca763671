"""Test helper: keys that collide in the open-addressing tables of the grouped steps (consensus_amd/csrc/p256_group.h) under a
KNOWN hash seed — what an adversary could precompute against the unkeyed hash of rounds 1-4 (seed 0).  Used by the emulator test of
the probe bound and by the GPU tests of the defence (random per-context seed + SBV_GROUP_MAX_PROBES)."""


def grouping_hash(words, seed):
    """Python twin of group_insert_lane_t's hash over the key words (little-endian u32 as the device loads them)."""
    M = 0xFFFFFFFF
    h = 0x9E3779B1 ^ seed
    for w in words:
        h = ((h ^ w) * 0x85EBCA77) & M
        h ^= h >> 15
    h = ((h ^ ((seed * 0x27D4EB2F) & M)) * 0xC2B2AE3D) & M
    return h ^ (h >> 16)


def colliding_keys(count, ht_bits, seed, rng, target=None, nwords=16):
    """`count` distinct keys of `nwords` 32-bit words (16: the 64-byte ECDSA keys, 8: Ed25519) whose grouping hash has the same low
    `ht_bits` bits under `seed`: the last key word is solved for (every step of the hash is a bijection of the state)."""
    M = 0xFFFFFFFF
    inv = lambda a: pow(a, -1, 1 << 32)                                   # noqa: E731
    def unshift(h, s):                                                    # inverse of h ^= h >> s
        x = h
        for _ in range(32 // s + 1):
            x = h ^ (x >> s)
        return x
    target = rng.getrandbits(ht_bits) if target is None else target
    out = []
    while len(out) < count:
        prefix = [rng.getrandbits(32) for _ in range(nwords - 1)]
        h15 = 0x9E3779B1 ^ seed
        for w in prefix:
            h15 = ((h15 ^ w) * 0x85EBCA77) & M
            h15 ^= h15 >> 15
        for hi in range(min(1 << (32 - ht_bits), count - len(out))):
            final = (hi << ht_bits) | target                              # wanted h ^ (h >> 16)
            h = unshift(final, 16)
            h = ((h * inv(0xC2B2AE3D)) & M) ^ ((seed * 0x27D4EB2F) & M)
            h = unshift(h, 15)
            x = (h * inv(0x85EBCA77)) & M
            words = prefix + [x ^ h15]
            assert grouping_hash(words, seed) & ((1 << ht_bits) - 1) == target
            out.append(b"".join(w.to_bytes(4, "little") for w in words))
    return out


