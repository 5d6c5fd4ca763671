#!/usr/bin/env python3
"""One grouped batch whose keys were crafted to collide in the grouping table of the UNKEYED hash (seed 0: what rounds 1-4 shipped)
against the same batch with random keys — verdicts and step times.  usage: hashflood_run.py <p256|k256|ed25519> [log2 junk keys]
Run it with SBV_HASH_SEED=0 to see the probe bound alone at work, without it for the library's random per-context seed.
Prints one JSON object.  Used by tests/test_gpu_hashflood.py; VERDICT r4 #4."""
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import consensus_amd as sbv  # noqa: E402
import hashflood  # noqa: E402
import hostlib  # noqa: E402

scheme = sys.argv[1] if len(sys.argv) > 1 else "p256"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 17
junk = 1 << lg
honest = junk
n = junk + honest
tb, koff, kwords = (128, 64, 8) if scheme == "ed25519" else (160, 96, 16)
THREADS = os.cpu_count() or 1

base = np.zeros(honest * tb, dtype=np.uint8)
exp = np.zeros((honest + 7) // 8, dtype=np.uint8)
if scheme == "p256":
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
    orc.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    orc.sbvo_gen_batch(0xF100D, honest, 64, 8, base.ctypes.data, exp.ctypes.data, THREADS)
    call = sbv.verify_batch_ptr
else:
    h = hostlib.load()
    gen = h.sbvh_ed25519_gen_batch if scheme == "ed25519" else h.sbvh_k256_gen_batch
    gen(0xF100D, honest, 64, 8, base.ctypes.data, exp.ctypes.data, THREADS)
    lib = sbv.load()
    fn = lib.sbv_ed25519_verify_batch if scheme == "ed25519" else lib.sbv_secp256k1_verify_batch

    def call(p, m, o):
        sbv._check(fn(p, m, o))
want = np.concatenate([np.unpackbits(exp, bitorder="little")[:honest], np.zeros(junk, dtype=np.uint8)])

rng = random.Random(2026)
t0 = time.time()
# low 22 bits: the table of the largest launch (2^21 tuples) has 2^22 entries, every smaller table uses a subset of these bits
collide = hashflood.colliding_keys(junk, 22, 0, rng, nwords=kwords)
craft_s = time.time() - t0
rand_keys = [rng.randbytes(4 * kwords) for _ in range(junk)]


def batch(keys):
    t = np.zeros((n, tb), dtype=np.uint8)
    hb = base.reshape(honest, tb)
    t[:honest] = hb
    t[honest:] = hb[np.arange(junk) % honest]                      # real signatures ...
    t[honest:, koff:koff + 4 * kwords] = np.frombuffer(b"".join(keys), dtype=np.uint8).reshape(junk, 4 * kwords)   # ... under crafted keys
    perm = np.random.default_rng(7).permutation(n)                 # interleave honest and crafted tuples
    return np.ascontiguousarray(t[perm]).reshape(-1), want[perm]


sbv.init(0)
out = {"scheme": scheme, "tuples": n, "crafted_keys": junk, "hash_seed_env": os.environ.get("SBV_HASH_SEED"), "craft_s": round(craft_s, 2)}
for name, keys in (("random", rand_keys), ("colliding", collide), ("random_again", rand_keys)):
    tup, w = batch(keys)
    got = np.zeros((n + 7) // 8, dtype=np.uint8)
    call(tup.ctypes.data, n, got.ctypes.data)                      # warm-up (buffers, tables of the honest keys)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        call(tup.ctypes.data, n, got.ctypes.data)
        ts.append(time.perf_counter() - t0)
    bits = np.unpackbits(got, bitorder="little")[:n]
    out[name] = {"ms": round(1e3 * sorted(ts)[2], 3), "worst_ms": round(1e3 * max(ts), 3), "verdicts_ok": bool((bits == w).all()),
                 "accepted": int(bits.sum())}
    if scheme == "p256":
        out[name]["group_stats"] = list(sbv.last_group_stats())
print(json.dumps(out))
