#!/usr/bin/env python3
"""Stress of the multi-context path on one GPU: SBV_LOGICAL_DEVICES contexts, the sharded generic entry over configs[3]-shaped batches again
and again, every bitmap compared; prints where a mismatch sits (shard, tuple range, kind).  usage: stress_logical.py [contexts] [calls per init cycle] [init cycles]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import numpy as np
    import consensus_amd as sbv
    import synth
    contexts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    group, quorum, props = 11, 10, 50000
    n = group * props
    tuples, valid = synth.gen_batch(0x5B7F2026 + 0x300, n, 16, 8)
    want = np.unpackbits(valid, bitorder="little")[:n]
    os.environ["SBV_LOGICAL_DEVICES"] = str(contexts)
    os.environ["SBV_SHARD_MIN"] = str(1 << 16)
    bad_calls = 0
    qgot = np.zeros((props + 7) // 8, dtype=np.uint8)
    all_entries = os.environ.get("STRESS_ENTRIES", "generic") == "all"       # also the registered-key and the message entries, alternating
    t2 = tuples.reshape(n, 160)
    keys, counts = np.unique(t2[:, 96:160], axis=0, return_counts=True)
    signer_keys = [bytes(k) for k in keys[counts > 1000]]
    rsh = np.ascontiguousarray(t2[:, :96]).reshape(-1)
    for rnd in range(rounds):
        sbv.shutdown()
        assert sbv.init_all() == contexts
        sbv.key_cache(True)
        slots = None
        if all_entries:
            slot_of = dict(zip(signer_keys, sbv.register_keys(signer_keys)))
            sbv.widen_keys(list(slot_of.values()))
            slots = np.fromiter((slot_of.get(bytes(k), 0xFFFFFFFF) for k in t2[:, 96:160]), dtype=np.uint32, count=n)
        for call in range(calls):
            got = np.zeros((n + 7) // 8, dtype=np.uint8)
            t0 = time.perf_counter()
            if all_entries and call % 2 == 1:
                info = sbv.verify_batch_keyed_sharded(rsh.ctypes.data, slots.ctypes.data, n, got.ctypes.data, group, quorum, qgot.ctypes.data)
            else:
                info = sbv.verify_batch_sharded(tuples.ctypes.data, n, got.ctypes.data, group, quorum, qgot.ctypes.data)
            ms = 1e3 * (time.perf_counter() - t0)
            bits = np.unpackbits(got, bitorder="little")[:n]
            diff = np.nonzero(bits != want)[0]
            if len(diff):
                bad_calls += 1
                per = info.tuples_per_shard
                print(json.dumps({"round": rnd, "call": call, "ms": round(ms, 2), "mismatches": int(len(diff)), "first": int(diff[0]), "last": int(diff[-1]),
                                  "shards_hit": sorted(set(int(d // per) for d in diff)), "false_accepts": int((bits[diff] == 1).sum()),
                                  "false_rejects": int((bits[diff] == 0).sum()), "hot": sbv.hot_key_stats()[0], "pool": sbv.pool_stats()}), flush=True)
                shard = int(diff[0] // per)
                os.environ["SBV_DEBUG_HOT_DUMP"] = "1"
                keys_hit = sorted(set(bytes(tuples[160 * int(d) + 96:160 * int(d) + 104]).hex() for d in diff[::97]))
                print(json.dumps({"shard": shard, "distinct_keys_hit": len(keys_hit), "hot_check_of_that_context": sbv.debug_hot_check(shard),
                                  "hot_check_ctx0": sbv.debug_hot_check(0)}), flush=True)
                break
            elif ms > 30:
                print(json.dumps({"round": rnd, "call": call, "ms": round(ms, 2), "slow": True, "pool": sbv.pool_stats()}), flush=True)
    print(json.dumps({"contexts": contexts, "calls": rounds * calls, "init_cycles": rounds, "bad_calls": bad_calls}), flush=True)
    sbv.shutdown()


if __name__ == "__main__":
    main()
