#!/usr/bin/env python3
"""A/B of the consenters' wide combs (sbv_p256_widen_keys) on the registered-key entry: 550 000 records (configs[3]: 50 000
proposals x 11 signatures, 16 keys) and 2^16 records, device-resident, 8-bit combs vs every width named on the command line
(default 16 18 20), in ONE process; per width: host build time, ms per call (median of 7), verdicts against the generator."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import consensus_amd as sbv
import synth

widths = [int(a) for a in sys.argv[1:]] or [16, 18, 20]
sbv.init(0)
stream = torch.cuda.Stream()
for n, nkeys in ((550000, 16), (1 << 16, 16), (1 << 20, 4)):
    tuples, valid = synth.gen_batch(0x5B7F2026 + 0x300, n, nkeys, 8)
    t2 = tuples.reshape(n, 160)
    keys, counts = np.unique(t2[:, 96:160], axis=0, return_counts=True)
    keys = keys[counts >= 64]
    sbv.clear_keys()
    sbv.wide_keys(16, 64)
    reg = sbv.register_keys([bytes(k) for k in keys])
    slots_of = dict(zip((bytes(k) for k in keys), reg))
    slots = np.fromiter((slots_of.get(bytes(k), 0xFFFFFFFF) for k in t2[:, 96:160]), dtype=np.uint32, count=n)
    d_rsh = torch.from_numpy(np.ascontiguousarray(t2[:, :96]).reshape(-1)).cuda()
    d_slots = torch.from_numpy(slots).cuda()
    d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")

    def run():
        ts = []
        for _ in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sbv.verify_batch_keyed_dev(d_rsh.data_ptr(), d_slots.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[2:])
        return round(1e3 * ts[len(ts) // 2], 4), bool((d_b.cpu().numpy() == valid).all())

    ms, ok = run()
    print(json.dumps({"n": n, "keys": int(len(keys)), "combs": "8-bit", "ms": ms, "M_per_s": round(n / ms / 1e3, 1), "ok": ok}), flush=True)
    for bits in widths:
        sbv.wide_keys(bits, 64)
        t0 = time.perf_counter()
        sbv.widen_keys(reg)
        build = time.perf_counter() - t0
        ms, ok = run()
        w = sbv.wide_key_stats()
        print(json.dumps({"n": n, "keys": int(len(keys)), "combs": f"{bits}-bit", "wide_keys": w[0], "MiB_per_key": round(w[3] / 1024.0, 1), "build_s": round(build, 2),
                          "ms": ms, "M_per_s": round(n / ms / 1e3, 1), "ok": ok}), flush=True)
    sbv.wide_keys(0, 0)
sbv.wide_keys()
sbv.clear_keys()
