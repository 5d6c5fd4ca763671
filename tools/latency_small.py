#!/usr/bin/env python3
"""Latency of the registered-key host entry for quorum-sized batches (n = 1, 15, 64): median wall time of sbv_p256_verify_batch_keyed
over 300 calls, through ctypes.  SBV_SMALL=0 in the environment selects the staged path (copies + two launches) for the A/B,
SBV_LAT_WIDE_BITS=16|18|20 gives the 16 keys their wide combs first (sbv_p256_widen_keys).
With arguments: one fresh process per environment variant, e.g. `latency_small.py default SBV_SMALL=0`."""
import ctypes, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:        # A/B: one fresh process per variant (NAME=VAL+NAME=VAL or "default"), two passes
    for rep in (1, 2):
        for var in sys.argv[1:]:
            env = dict(os.environ)
            if var != "default":
                env.update(kv.split("=", 1) for kv in var.split("+"))
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
            print(json.dumps({"variant": var, "pass": rep, "result": json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else r.stderr[-400:]}), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import consensus_amd as sbv
import synth
sbv.init(0)
tuples, valid = synth.gen_batch(0x5B7F2026, 1 << 16, 16, 8)
t2 = tuples.reshape(-1, 160)
keys = [bytes(t2[i, 96:160]) for i in range(16)]
slots_of = dict(zip(keys, sbv.register_keys(keys)))
wide_bits = int(os.environ.get("SBV_LAT_WIDE_BITS", "0"))       # the consenters' wide combs (sbv_p256_widen_keys); 0 = the 8-bit combs
if wide_bits:
    sbv.wide_keys(wide_bits, 64)
    sbv.widen_keys(list(slots_of.values()))
lib = sbv.load()
lib.sbv_p256_verify_batch_keyed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
out = {"small_path": os.environ.get("SBV_SMALL", "1") != "0", "wide_keys": sbv.wide_key_stats()[0], "wide_bits": wide_bits}
for n in (1, 15, 64, 65):
    rsh = np.ascontiguousarray(t2[:n, :96]).reshape(-1)
    slots = np.array([slots_of.get(bytes(t2[i, 96:160]), 0xFFFFFFFF) for i in range(n)], dtype=np.uint32)
    bm = np.zeros(16, dtype=np.uint8)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter()
        rc = lib.sbv_p256_verify_batch_keyed(rsh.ctypes.data, slots.ctypes.data, n, bm.ctypes.data)
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    want = np.unpackbits(valid, bitorder="little")[:n]
    got = np.unpackbits(bm, bitorder="little")[:n]
    ts.sort()
    out[f"n={n}"] = {"median_us": round(1e6 * ts[len(ts) // 2], 1), "p10_us": round(1e6 * ts[30], 1), "p90_us": round(1e6 * ts[270], 1), "correct": bool((got == want).all())}
print(json.dumps(out))
