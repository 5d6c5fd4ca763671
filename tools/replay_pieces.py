#!/usr/bin/env python3
"""configs[3] (50 000 proposals x 11 signatures, 16 signers) through the sharded GENERIC entry, PCIe-inclusive, as a function of the
upload piece (SBV_SHARD_PIECE) and of the number of logical devices (SBV_LOGICAL_DEVICES): one JSON line per setting.
usage: replay_pieces.py [piece,piece,...] [logical,logical,...]"""
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
    pieces = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "131072,262144,524288,1048576").split(",")]
    logicals = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1").split(",")]
    group, quorum, props = 11, 10, 50000
    n = group * props
    tuples, valid = synth.gen_batch(0x5B7F2026 + 0x300, n, 16, 8)
    got = np.zeros((n + 7) // 8, dtype=np.uint8)
    qgot = np.zeros((props + 7) // 8, dtype=np.uint8)
    for logical in logicals:
        for piece in pieces:
            os.environ["SBV_SHARD_PIECE"] = str(piece)
            if logical > 1:
                os.environ["SBV_LOGICAL_DEVICES"] = str(logical)
                os.environ["SBV_SHARD_MIN"] = str(1 << 16)
            else:
                os.environ.pop("SBV_LOGICAL_DEVICES", None)
            sbv.shutdown()
            sbv.init_all()
            sbv.key_cache(True)
            for _ in range(3):
                info = sbv.verify_batch_sharded(tuples.ctypes.data, n, got.ctypes.data, group, quorum, qgot.ctypes.data)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter()
                info = sbv.verify_batch_sharded(tuples.ctypes.data, n, got.ctypes.data, group, quorum, qgot.ctypes.data)
                ts.append(1e3 * (time.perf_counter() - t0))
            ok = bool((got == valid).all())
            if not ok:
                bits, want = np.unpackbits(got, bitorder="little")[:n], np.unpackbits(valid, bitorder="little")[:n]
                diff = np.nonzero(bits != want)[0]
                print(json.dumps({"MISMATCH": int(len(diff)), "first": int(diff[0]), "last": int(diff[-1]), "per_shard": int(info.tuples_per_shard),
                                  "shards_hit": sorted(set(int(d // info.tuples_per_shard) for d in diff)), "false_accepts": int((bits[diff] == 1).sum()),
                                  "runs": [[int(a), int(b)] for a, b in zip(diff[np.r_[True, np.diff(diff) > 1]][:6], diff[np.r_[np.diff(diff) > 1, True]][:6])]}), flush=True)
            print(json.dumps({"logical_devices": logical, "piece": piece, "ms_median": sorted(ts)[3], "ms_min": min(ts), "shards": info.shards,
                              "h2d_us": info.h2d_us, "kernels_us": info.kernels_us, "gather_us": info.gather_us, "ok": ok, "hot": sbv.hot_key_stats()[0]}), flush=True)
    sbv.shutdown()


if __name__ == "__main__":
    main()
