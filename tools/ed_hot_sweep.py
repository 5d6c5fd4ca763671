#!/usr/bin/env python3
"""Ed25519 hot keys (sbv_ed25519_hot_keys) against the cached 8-bit combs alone, by batch size, signer count and pool size: JSON lines
{n, keys, pool, promoted, wide_tuples, ms}.  pool 0 = the feature off.  usage: ed_hot_sweep.py [log2n,...] [keys,...] [pool as a percentage of the keys,...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import bench
    import consensus_amd as sbv
    import hostlib
    logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "18,20").split(",")]
    keyset = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "96,1024").split(",")]
    h = hostlib.load()
    sbv.init(0)
    stream = torch.cuda.current_stream()
    for lg in logs:
        n = 1 << lg
        for keys in keyset:
            tuples = np.zeros(n * 128, dtype=np.uint8)
            expect = np.zeros((n + 7) // 8, dtype=np.uint8)
            h.sbvh_ed25519_gen_batch(bench.SEED + keys, n, keys, 8, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)
            d_t = torch.from_numpy(tuples).cuda()
            d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
            call = lambda: sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)      # noqa: E731
            for pool in [keys * int(x) // 100 for x in (sys.argv[3] if len(sys.argv) > 3 else "0,50,100").split(",")]:
                sbv.ed_hot_keys(pool, 64)
                promoted = 0
                for _ in range(40):
                    call()
                    torch.cuda.synchronize()
                    if pool == 0:
                        break
                    promoted, cap = sbv.ed_hot_key_stats()[:2]
                    if promoted >= min(pool, cap):
                        break
                call()
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(8):
                        call()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) / 8)
                st = sbv.ed_hot_key_stats()
                print(json.dumps({"n": n, "keys": keys, "pool": pool, "promoted": st[0], "wide_tuples": st[2], "ms": round(1e3 * min(ts), 4),
                                  "bitmap_correct": bool((d_b.cpu().numpy() == expect).all())}), flush=True)
    sbv.ed_hot_keys(1024, 4096)


if __name__ == "__main__":
    main()
