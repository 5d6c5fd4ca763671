#!/usr/bin/env python3
"""One process, one 2^20 batch: time the device-pointer entry on the first n tuples for several n, with the key-table cache
cold (off: every step rebuilds what it groups), warm (on, tables resident after the first call) and with grouping off (the
one-lane doubling kernel).  One JSON line per size.  usage: sweep_sizes.py [log2 sizes ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import numpy as np
    import torch
    import consensus_amd as sbv
    import synth
    logs = [int(a) for a in sys.argv[1:]] or [10, 12, 14, 16, 17, 18, 19, 20]
    nmax = 1 << max(logs)
    tuples, valid = synth.gen_batch(0x5B7F2026, max(nmax, 1 << 20))
    sbv.init(0)
    stream = torch.cuda.current_stream()
    d_all = torch.from_numpy(tuples).cuda()
    d_big = torch.zeros(1 << 17, dtype=torch.uint8, device="cuda")
    for lg in logs:
        n = 1 << lg
        d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
        want = valid[:n // 8]
        row = {"log2_tuples": lg}

        def timed(label, steps):
            sbv.verify_batch_dev(d_all.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                sbv.verify_batch_dev(d_all.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            ok = bool((d_b.cpu().numpy()[:n // 8] == want).all())
            g = sbv.last_group_stats()
            row[label] = {"ms": round(1e3 * dt, 4), "M_per_s": round(n / dt / 1e6, 2), "ok": ok, "groups": g[0], "grouped": g[1], "generic": g[2]}

        steps = 20 if lg <= 16 else 8
        sbv.set_grouping(True)
        sbv.key_cache(False)
        timed("cold", steps)
        sbv.key_cache(True)             # warm = every signer's comb already resident: one full batch fills the cache first
        sbv.verify_batch_dev(d_all.data_ptr(), 1 << 20, d_big.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        timed("warm", steps)
        sbv.key_cache(False)
        sbv.set_grouping(False)
        timed("one_lane", max(3, steps // 4))
        sbv.set_grouping(True)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
