#!/usr/bin/env python3
"""Which earlier small batch poisons the warm/cold verification of the headline batch?  (GPU session e)"""
import collections, ctypes, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import consensus_amd as gpu
import synth
oracle = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
oracle.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
n = 1 << 20
tuples, valid = synth.gen_batch(0x5B7F2026, n)
want = np.unpackbits(valid, bitorder="little")[:n]
t2 = tuples.reshape(n, 160)
def big(label):
    got = np.zeros(n // 8, dtype=np.uint8)
    gpu.verify_batch_ptr(tuples.ctypes.data, n, got.ctypes.data)
    bits = np.unpackbits(got, bitorder="little")[:n]
    bad = np.nonzero(bits != want)[0]
    keys = collections.Counter(bytes(t2[i, 96:160]) for i in bad[:5000])
    print(label, "mismatches", len(bad), "first", bad[:6].tolist(), "distinct keys", len(keys), "cache", gpu.key_cache_stats(), "groups", gpu.last_group_stats(), flush=True)
def gen(seed, m, nk, inv):
    tup = ctypes.create_string_buffer(160 * m); exp = ctypes.create_string_buffer((m + 7) // 8)
    oracle.sbvo_gen_batch(seed, m, nk, inv, tup, exp, 8)
    return tup.raw
def reset():
    gpu.key_cache(False); gpu.key_cache(True, 4096)
gpu.init(0)
rng = random.Random(99)
scen = {
    "nothing": lambda: None,
    "zeros100": lambda: gpu.verify_batch(bytes(160 * 100), 100),
    "ones100": lambda: gpu.verify_batch(b"\xff" * 16000, 100),
    "junk3000": lambda: gpu.verify_batch(bytes(rng.getrandbits(8) for _ in range(160 * 3000)), 3000),
    "onekey512": lambda: gpu.verify_batch(gen(7, 512, 1, 0), 512),
    "ragged": lambda: [gpu.verify_batch(gen(0xA000 + m, m, 33, 3), m) for m in (1, 7, 64, 65, 255, 1000, 4097, 20000)],
}
for name in sys.argv[1:] or list(scen):
    reset()
    scen[name]()
    big("after " + name + " (cold big)")
    big("after " + name + " (warm big)")
