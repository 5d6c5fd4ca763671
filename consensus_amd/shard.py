"""Sharding one batch of tuples over the GPUs of a node (one process per GPU, torch.distributed).

SURVEY.md §8e: every tuple is independent, so a batch that outgrows one GPU is split
contiguously by tuple; rank g verifies [lo_g, hi_g) on its own MI355X and the only exchange
step is an all-gather of the per-rank accept-bitmap shards (ceil(B/8G) bytes each — latency,
not bandwidth, on xGMI).  Shard sizes are multiples of 512 tuples so every rank writes whole
wavefront ballots / bitmap bytes.  Batches that fit one GPU are never sharded ("replicas").

The per-rank verification is `verify_fn(tuples: bytes-like, n) -> bitmap bytes`; the product
default is the HIP path (consensus_amd.verify_batch).  Tests on CPU inject a stand-in and use
the `gloo` backend (tests/test_shard_gloo.py); on GPUs the backend is `nccl` (= RCCL).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

GRANULE = 512
TUPLE_BYTES = 160


def shard_bounds(n: int, world: int, rank: int, granule: int = GRANULE) -> Tuple[int, int]:
    """Contiguous split: rank g gets [g*per, min(n, (g+1)*per)), per = ceil(n/world) rounded up to granule."""
    per = -(-n // world)
    per = -(-per // granule) * granule
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def shard_capacity_bytes(n: int, world: int, granule: int = GRANULE) -> int:
    per = -(-n // world)
    per = -(-per // granule) * granule
    return per // 8


def sharded_verify(tuples, n: int, verify_fn: Optional[Callable] = None, device: str = "cpu") -> bytes:
    """Every rank passes the same (tuples, n); returns the full ceil(n/8)-byte bitmap on every rank."""
    import torch
    import torch.distributed as dist

    if verify_fn is None:
        import consensus_amd
        verify_fn = consensus_amd.verify_batch
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return verify_fn(bytes(tuples[:n * TUPLE_BYTES]), n)
    lo, hi = shard_bounds(n, world, rank)
    cap = shard_capacity_bytes(n, world)
    local = bytes(verify_fn(bytes(tuples[lo * TUPLE_BYTES:hi * TUPLE_BYTES]), hi - lo)) if hi > lo else b""
    send = torch.zeros(cap, dtype=torch.uint8, device=device)
    if local:
        send[:len(local)] = torch.frombuffer(bytearray(local), dtype=torch.uint8).to(device)
    recv = torch.zeros(cap * world, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send)
    # shards start at multiples of `granule` tuples = whole bytes, so concatenation is the bitmap
    return bytes(recv.cpu().numpy().tobytes())[:(n + 7) // 8]


# ---- key-affine partition ------------------------------------------------------------------------------------------------
# The contiguous split above hands every rank signatures of every signer, so every rank builds every key's comb tables — the
# part of a cold step that does not shrink with the shard.  Partitioning by a hash of the public key
# gives rank g the tuples of "its" keys only: K / G tables, n / G tuples.  The exchange step becomes an all-reduce of
# full-size bitmaps with disjoint bits (OR = byte-wise sum) instead of an all-gather of shards.  The hash is the one
# libsbv.so uses on the device (sbv_api.hip: part_of_key), restated in numpy; tests pin the two against each other.
def key_parts(tuples, n: int, parts: int):
    """part id (0 .. parts-1) of every tuple, from its 64 key bytes — numpy twin of the device's part_of_key."""
    import numpy as np
    t = np.frombuffer(tuples, dtype=np.uint8, count=n * TUPLE_BYTES).reshape(n, TUPLE_BYTES)
    k = np.ascontiguousarray(t[:, 96:160]).view("<u4").astype(np.uint64)          # 16 little-endian words, as the device loads them
    mask = np.uint64(0xFFFFFFFF)
    h = np.full(n, 0x2545F491, dtype=np.uint64)
    for j in range(16):
        h = ((h ^ k[:, j]) * np.uint64(0x9E3779B1)) & mask
        h ^= h >> np.uint64(13)
    h = (h * np.uint64(0x85EBCA77)) & mask
    return ((h >> np.uint64(11)) % np.uint64(parts)).astype(np.int64)


def quorum_bits(tuples, bitmap: bytes, n: int, group: int, quorum: int) -> bytes:
    """Bit p = proposal p (tuples [p*group, (p+1)*group)) carries >= quorum accepted signatures by DISTINCT keys
    (internal/bft/viewchanger.go:681-727); LSB-first like the accept bitmap.  A ragged tail gets no bit."""
    import numpy as np
    props = n // group
    bits = np.unpackbits(np.frombuffer(bitmap, dtype=np.uint8), bitorder="little")[:props * group].reshape(props, group)
    t = np.frombuffer(tuples, dtype=np.uint8, count=n * TUPLE_BYTES).reshape(n, TUPLE_BYTES)[:props * group, 96:160].reshape(props, group, 64)
    out = np.zeros(props, dtype=np.uint8)
    for p in range(props):
        seen = {t[p, i].tobytes() for i in range(group) if bits[p, i]}
        out[p] = 1 if len(seen) >= quorum else 0
    return np.packbits(out, bitorder="little").tobytes()


def sharded_verify_by_key(tuples, n: int, verify_fn: Optional[Callable] = None, group: int = 0, quorum: int = 0, device: str = "cpu"):
    """Every rank passes the same (tuples, n); rank g verifies the tuples whose key hashes to part g.  Returns
    (full accept bitmap, per-proposal quorum bitmap or None) on every rank."""
    import numpy as np
    import torch
    import torch.distributed as dist

    if verify_fn is None:
        import consensus_amd
        verify_fn = consensus_amd.verify_batch
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = np.nonzero(key_parts(tuples, n, world) == rank)[0] if world > 1 else np.arange(n)
    t = np.frombuffer(tuples, dtype=np.uint8, count=n * TUPLE_BYTES).reshape(n, TUPLE_BYTES)
    bits = np.zeros(n, dtype=np.uint8)
    if len(mine):
        local = verify_fn(np.ascontiguousarray(t[mine]).tobytes(), len(mine))
        bits[mine] = np.unpackbits(np.frombuffer(bytes(local), dtype=np.uint8), bitorder="little")[:len(mine)]
    packed = np.packbits(bits, bitorder="little")
    if world > 1:
        buf = torch.from_numpy(packed.copy()).to(device)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)           # disjoint bits: the sum IS the OR
        packed = buf.cpu().numpy()
    full = packed.tobytes()[:(n + 7) // 8]
    q = quorum_bits(tuples, full, n, group, quorum) if group and quorum else None
    return full, q


# ---- registered-key form (libsbv.so: sbv_p256_verify_batch_keyed_sharded) -------------------------------------------------
# configs[3] as BASELINE.json words it: the consenters' commit signatures travel as 96-byte records r | s | hash plus a 4-byte key
# slot (types.Signature.ID -> registry slot), every rank holds a replica of the registry, shards are whole proposals (granule
# lcm(512, 8 * group)), and the only exchange is the all-gather of the bitmap shards.  Quorum bits count DISTINCT accepted slots.
RSH_BYTES = 96


def keyed_granule(group: int) -> int:
    import math
    g8 = 8 * (group or 1)
    return 512 // math.gcd(512, g8) * g8


def quorum_bits_slots(slots, bitmap: bytes, n: int, group: int, quorum: int) -> bytes:
    """numpy twin of the device's k_quorum_bits_slots: bit p = proposal p has >= quorum accepted signatures by distinct slots."""
    import numpy as np
    props = n // group
    bits = np.unpackbits(np.frombuffer(bitmap, dtype=np.uint8), bitorder="little")[:props * group].reshape(props, group).astype(bool)
    sl = np.asarray(slots, dtype=np.int64)[:props * group].reshape(props, group)
    out = np.zeros(props, dtype=np.uint8)
    for p in range(props):
        out[p] = 1 if len(set(sl[p][bits[p]].tolist())) >= quorum else 0
    return np.packbits(out, bitorder="little").tobytes()


def sharded_verify_keyed(rsh, slots, n: int, verify_fn: Optional[Callable] = None, group: int = 0, quorum: int = 0, device: str = "cpu"):
    """Every rank passes the same (rsh, slots, n); rank g verifies the contiguous range of whole proposals the plan gives it
    against ITS replica of the key registry (`verify_fn(rsh_bytes, slot_list, m) -> bitmap`; default: the HIP registered-key
    entry).  Returns (full accept bitmap, per-proposal quorum bitmap or None) on every rank."""
    import numpy as np
    import torch
    import torch.distributed as dist

    if verify_fn is None:
        import consensus_amd
        verify_fn = consensus_amd.verify_batch_keyed
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    gran = keyed_granule(group)
    lo, hi = shard_bounds(n, world, rank, gran)
    cap = shard_capacity_bytes(n, world, gran)
    sl = np.asarray(slots, dtype=np.uint32)
    local = bytes(verify_fn(bytes(rsh[lo * RSH_BYTES:hi * RSH_BYTES]), sl[lo:hi].tolist(), hi - lo)) if hi > lo else b""
    if world == 1:
        full = local
    else:
        send = torch.zeros(cap, dtype=torch.uint8, device=device)
        if local:
            send[:len(local)] = torch.frombuffer(bytearray(local), dtype=torch.uint8).to(device)
        recv = torch.zeros(cap * world, dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(recv, send)
        full = bytes(recv.cpu().numpy().tobytes())[:(n + 7) // 8]
    q = quorum_bits_slots(sl, full, n, group, quorum) if group and quorum else None
    return full, q
