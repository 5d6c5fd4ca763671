#!/usr/bin/env python3
"""A/B of environment-selected variants of the grouped step in ONE process (one box, one batch): for every variant
(NAME=VAL,NAME=VAL or "default") the library is shut down and initialised again — init_context re-reads the environment —
and the device-pointer entry is timed cold (key cache off) and warm on the first n tuples of the 2^20 batch.
usage: ab_env.py <log2 sizes, comma separated> <variant> [<variant> ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import consensus_amd as sbv
    import synth
    logs = [int(a) for a in sys.argv[1].split(",")]
    variants = sys.argv[2:] or ["default"]
    tuples, valid = synth.gen_batch(0x5B7F2026, 1 << 20)
    d_all = torch.from_numpy(tuples).cuda()
    d_big = torch.zeros(1 << 17, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    touched = set()
    for rep in range(2):                  # every variant twice, interleaved: box drift shows up as a difference between the passes
        for var in variants:
            for k in touched:
                os.environ.pop(k, None)
            if var != "default":
                for kv in var.split(","):
                    k, v = kv.split("=")
                    os.environ[k] = v
                    touched.add(k)
            sbv.shutdown()
            sbv.init(0)
            row = {"variant": var, "pass": rep}
            for lg in logs:
                n = 1 << lg
                d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
                res = {}
                for label, warm in (("cold", False), ("warm", True)):
                    sbv.key_cache(warm)
                    if warm:
                        sbv.verify_batch_dev(d_all.data_ptr(), 1 << 20, d_big.data_ptr(), stream.cuda_stream)
                    steps = 10 if lg >= 19 else 20
                    sbv.verify_batch_dev(d_all.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        sbv.verify_batch_dev(d_all.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / steps
                    ok = bool((d_b.cpu().numpy()[:n // 8] == valid[:n // 8]).all())
                    res[label] = round(1e3 * dt, 4) if ok else "WRONG"
                sbv.key_cache(False)
                row[f"2^{lg}_ms"] = res
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
