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
