#!/usr/bin/env python3
"""Start-up stress of the Ed25519 hot keys: init -> a few batches (promotion in the first, the wide pass from the second on) -> every bitmap
against the generator's, one comb against the host builder -> shutdown, again and again (the P-256 pools' start-up race of round 6 was
found this way: tools/stress_logical.py).  usage: stress_ed_hot.py [init cycles] [batches per cycle]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import consensus_amd as sbv
    import hostlib
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n, keys = 1 << 17, 64
    h = hostlib.load()
    tuples = np.zeros(n * 128, dtype=np.uint8)
    expect = np.zeros((n + 7) // 8, dtype=np.uint8)
    h.sbvh_ed25519_gen_batch(0xE5D0, n, keys, 8, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)
    lib = sbv.load()
    bad = 0
    for cycle in range(cycles):
        sbv.init(0)
        sbv.ed_hot_keys(64, 1024)
        wide = []
        for call in range(calls):
            got = np.zeros((n + 7) // 8, dtype=np.uint8)
            sbv._check(lib.sbv_ed25519_verify_batch(ctypes.c_void_p(tuples.ctypes.data), n, ctypes.c_void_p(got.ctypes.data)))
            ok = bool((got == expect).all())
            st = sbv.ed_hot_key_stats()
            wide.append(st[2])
            if not ok:
                bad += 1
                diff = np.nonzero(np.unpackbits(got ^ expect, bitorder="little")[:n])[0]
                print(json.dumps({"cycle": cycle, "call": call, "bad_tuples": int(len(diff)), "first": [int(x) for x in diff[:8]], "stats": list(st)}), flush=True)
        st = sbv.ed_hot_key_stats()
        check = bool(sbv.ed_hot_selfcheck(cycle % max(1, st[0]))) if st[0] else None
        if check is False:
            bad += 1
        print(json.dumps({"cycle": cycle, "promoted": st[0], "wide_tuples_per_call": wide, "comb_equals_host_builder": check}), flush=True)
        sbv.shutdown()
    print(json.dumps({"init_cycles": cycles, "calls_per_cycle": calls, "bad": bad}))


if __name__ == "__main__":
    main()
