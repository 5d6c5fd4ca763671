#!/usr/bin/env python3
"""Diagnostics: table classes of consecutive batches over the same keys (cache on): groups / classes / cache stats per run."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import consensus_amd as sbv
o = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
o.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
def gen(seed, n, nkeys, inv=7):
    t = np.zeros(n * 160, dtype=np.uint8); e = np.zeros((n + 7) // 8, dtype=np.uint8)
    o.sbvo_gen_batch(seed, n, nkeys, inv, t.ctypes.data, e.ctypes.data, os.cpu_count() or 1)
    return t, e
sbv.init(0)
for nkeys, n, inv in ((8, 8 * 16384, 7), (8, 8 * 16384, 0), (64, 64 * 4096, 7)):
    t, e = gen(0x51, n, nkeys, inv)
    got = np.zeros((n + 7) // 8, dtype=np.uint8)
    sbv.key_cache(False); sbv.key_cache(True)
    for run in range(3):
        sbv.verify_batch_ptr(t.ctypes.data, n, got.ctypes.data)
        print(nkeys, n, inv, "run", run, "ok", bool((got == e).all()), "groups", sbv.last_group_stats(), "classes", sbv.last_table_classes(), "cache", sbv.key_cache_stats(), flush=True)
