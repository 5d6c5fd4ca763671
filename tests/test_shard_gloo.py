"""N > 1 path on CPU: world_size-2 gloo run of the shard + all-gather logic.
The per-rank verifier is a stand-in (the oracle) because the container has no GPU; on the GPU
box the same code runs over RCCL with the HIP path (bench.py --gpus N)."""
import ctypes
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from consensus_amd import shard  # noqa: E402


def test_shard_bounds_cover_and_align():
    for n in [0, 1, 511, 512, 513, 1000, 4096, 550000, 1 << 20, (1 << 20) + 7]:
        for world in [1, 2, 3, 4, 8]:
            prev = 0
            for r in range(world):
                lo, hi = shard.shard_bounds(n, world, r)
                assert lo == prev or lo == n
                assert lo % 512 == 0 or lo == n
                assert hi - lo <= shard.shard_capacity_bytes(n, world) * 8
                prev = hi
            assert prev == n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
    lib.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.sbvo_p256_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    lib.sbvo_gen_batch(0xBEEF, n, 16, 3, tup, exp, 1)

    calls = []

    def stand_in(tuples, m):
        calls.append(m)
        bm = ctypes.create_string_buffer(max(1, (m + 7) // 8))
        lib.sbvo_p256_verify_batch(tuples, m, bm, 1)
        return bm.raw[:(m + 7) // 8]

    full = shard.sharded_verify(tup.raw, n, verify_fn=stand_in)
    lo, hi = shard.shard_bounds(n, world, rank)
    out_q.put((rank, full == exp.raw[:(n + 7) // 8], calls == ([hi - lo] if hi > lo else [])))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1500, 300])
def test_world_size_2_gloo_gathers_full_bitmap(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), "gathered bitmap differs from the single-process result"
    assert all(r[2] for r in res), "a rank verified something other than exactly its shard"


# ---- key-affine partition (consensus_amd/shard.py: sharded_verify_by_key) -------------------------------------------------
def _worker_by_key(rank, world, port, n, group, quorum, out_q):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
    lib.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.sbvo_p256_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    lib.sbvo_gen_batch(0xC0DE, n, 16, 5, tup, exp, 1)
    raw = bytearray(tup.raw)
    raw[160 * 3 + 96:160 * 3 + 160] = raw[160 * 4 + 96:160 * 4 + 160]      # proposal 0: tuple 3 now carries tuple 4's key (its signature no longer verifies)
    raw = bytes(raw)
    want = ctypes.create_string_buffer((n + 7) // 8)
    lib.sbvo_p256_verify_batch(raw, n, want, 1)
    seen_keys = set()

    def stand_in(tuples, m):
        for i in range(m):
            seen_keys.add(tuples[160 * i + 96:160 * i + 160])
        bm = ctypes.create_string_buffer(max(1, (m + 7) // 8))
        lib.sbvo_p256_verify_batch(tuples, m, bm, 1)
        return bm.raw[:(m + 7) // 8]

    full, q = shard.sharded_verify_by_key(raw, n, verify_fn=stand_in, group=group, quorum=quorum)
    parts = shard.key_parts(raw, n, world)
    mine_keys = {raw[160 * i + 96:160 * i + 160] for i in range(n) if parts[i] == rank}
    # independent statement of the quorum rule
    bits = [(want.raw[i >> 3] >> (i & 7)) & 1 for i in range(n)]
    wq = []
    for p in range(n // group):
        keys = {raw[160 * i + 96:160 * i + 160] for i in range(p * group, (p + 1) * group) if bits[i]}
        wq.append(1 if len(keys) >= quorum else 0)
    got_q = [(q[p >> 3] >> (p & 7)) & 1 for p in range(n // group)]
    out_q.put((rank, full == want.raw[:(n + 7) // 8], seen_keys == mine_keys, got_q == wq, int(np.bincount(parts, minlength=world)[rank])))
    dist.destroy_process_group()


def test_world_size_2_gloo_key_affine_partition_and_quorum_bits():
    world, n, group, quorum = 2, 16 * 11 * 4 + 5, 11, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_by_key, args=(r, world, port, n, group, quorum, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), "combined bitmap differs from the single-process result"
    assert all(r[2] for r in res), "a rank saw a key of the other rank's part: the partition is not key-affine"
    assert all(r[3] for r in res), "per-proposal quorum bits differ from the rule"
    assert sum(r[4] for r in res) == n and all(r[4] > 0 for r in res)


# ---- registered-key form (consensus_amd/shard.py: sharded_verify_keyed; libsbv.so: sbv_p256_verify_batch_keyed_sharded) --------
def _worker_keyed(rank, world, port, props, group, quorum, out_q):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
    lib.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.sbvo_p256_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    n = props * group
    tup = ctypes.create_string_buffer(160 * n)
    exp = ctypes.create_string_buffer((n + 7) // 8)
    lib.sbvo_gen_batch(0xFACE, n, 16, 6, tup, exp, 1)
    raw = bytearray(tup.raw)
    # proposal 1: its second signature becomes a copy of its first (same signer twice: one distinct slot fewer)
    raw[160 * (group + 1):160 * (group + 2)] = raw[160 * group:160 * (group + 1)]
    raw = bytes(raw)
    # the registry every rank replicates: keys that sign often get a slot, a corrupted key (a one-off) is an unknown signer
    counts = {}
    for i in range(n):
        k = raw[160 * i + 96:160 * i + 160]
        counts[k] = counts.get(k, 0) + 1
    registry = sorted(k for k, c in counts.items() if c > 8)
    slot_of = {k: i for i, k in enumerate(registry)}
    slots = [slot_of.get(raw[160 * i + 96:160 * i + 160], 0xFFFFFFFF) for i in range(n)]
    rsh = b"".join(raw[160 * i:160 * i + 96] for i in range(n))
    want = ctypes.create_string_buffer((n + 7) // 8)
    lib.sbvo_p256_verify_batch(raw, n, want, 1)
    ranges = []

    def stand_in(rsh_part, slot_part, m):            # this rank's replica of the registry: slot -> key, unknown slot -> reject
        ranges.append(m)
        t = bytearray(160 * m)
        for i in range(m):
            t[160 * i:160 * i + 96] = rsh_part[96 * i:96 * i + 96]
            if slot_part[i] < len(registry):
                t[160 * i + 96:160 * i + 160] = registry[slot_part[i]]
        bm = ctypes.create_string_buffer(max(1, (m + 7) // 8))
        lib.sbvo_p256_verify_batch(bytes(t), m, bm, 1)
        return bm.raw[:(m + 7) // 8]

    full, q = shard.sharded_verify_keyed(rsh, slots, n, verify_fn=stand_in, group=group, quorum=quorum)
    lo, hi = shard.shard_bounds(n, world, rank, shard.keyed_granule(group))
    bits = [(want.raw[i >> 3] >> (i & 7)) & 1 for i in range(n)]
    wq = []
    for p in range(props):                           # the rule restated on KEYS: >= quorum accepted signatures by distinct signers
        keys = {raw[160 * i + 96:160 * i + 160] for i in range(p * group, (p + 1) * group) if bits[i]}
        wq.append(1 if len(keys) >= quorum else 0)
    got_q = [(q[p >> 3] >> (p & 7)) & 1 for p in range(props)]
    out_q.put((rank, full == want.raw[:(n + 7) // 8], ranges == ([hi - lo] if hi > lo else []), got_q == wq, lo % group == 0 and (hi - lo) > 0, sum(wq)))
    dist.destroy_process_group()


def test_world_size_2_gloo_registered_key_shards_and_quorum_bits_by_slot():
    world, props, group, quorum = 2, 640, 11, 10          # 7 040 signatures: rank 0 takes one granule of 5 632 (512 proposals), rank 1 the rest
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_keyed, args=(r, world, port, props, group, quorum, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), "gathered bitmap differs from the single-process result"
    assert all(r[2] for r in res), "a rank verified something other than exactly its shard"
    assert all(r[3] for r in res), "quorum bits by slot differ from the rule restated on keys"
    assert all(r[4] for r in res), "a shard does not start at a proposal boundary, or a rank got nothing"
    assert 0 < res[0][5] < props
