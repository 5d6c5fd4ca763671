#!/usr/bin/env python3
"""Diagnostic for the key-cache GPU test: which tuples of batch c get a wrong verdict, and what they have in common."""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import consensus_amd as gpu
oracle = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
oracle.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
n = 1 << 18
def batch(seed, nkeys):
    tup = ctypes.create_string_buffer(160 * n); exp = ctypes.create_string_buffer(n // 8)
    oracle.sbvo_gen_batch(seed, n, nkeys, 7, tup, exp, os.cpu_count() or 1)
    return tup, exp.raw
def run(name, tup, exp):
    got = ctypes.create_string_buffer(n // 8)
    gpu.verify_batch_ptr(ctypes.addressof(tup), n, ctypes.addressof(got))
    g, raw = got.raw, tup.raw
    bad = [i for i in range(n) if ((g[i >> 3] >> (i & 7)) & 1) != ((exp[i >> 3] >> (i & 7)) & 1)]
    keys = collections.Counter(raw[160 * i + 96:160 * i + 160] for i in range(n))
    info = []
    for i in bad[:12]:
        k = raw[160 * i + 96:160 * i + 160]
        info.append((i, (exp[i >> 3] >> (i & 7)) & 1, keys[k]))
    badkeys = collections.Counter(raw[160 * i + 96:160 * i + 160] for i in bad)
    print(name, "mismatches", len(bad), "distinct keys among them", len(badkeys), "cache", gpu.key_cache_stats(), "groups", gpu.last_group_stats(),
          "first (idx, expected, uses of its key)", info, "per-key (bad, uses)", [(v, keys[k]) for k, v in badkeys.most_common(6)], flush=True)
gpu.init(0)
a, ea = batch(0xE1, 300); c, ec = batch(0xE2, 300)
mode = sys.argv[1] if len(sys.argv) > 1 else "abc"
gpu.key_cache(False); gpu.key_cache(True, 4096)
for ch in mode:
    if ch == "a": run("a", a, ea)
    if ch == "c": run("c", c, ec)
    if ch == "o": gpu.key_cache(False); run("c(cache off)", c, ec); gpu.key_cache(True, 4096)
