#!/usr/bin/env python3
"""Hot-key promotion on the device (dev sessions): the headline batch (2^20 tuples over K keys, default 1024) cold, warm on 8-bit
tables, batch by batch while the promotions run, and with every signer on a 16-bit comb.  usage: hot_keys_run.py [K] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
import consensus_amd as sbv  # noqa: E402
import synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << 20
sbv.init(0)
tuples, valid = synth.gen_batch(bench.SEED + 0x900 + K, n, K, 8) if K != 1024 else synth.gen_batch(bench.SEED, n)
d_t = torch.from_numpy(tuples).cuda()
d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
out = {"keys": K, "tuples": n}


def step():
    t0 = time.perf_counter()
    sbv.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


def med(k):
    ts = sorted(step() for _ in range(k))
    return ts[len(ts) // 2]


with torch.cuda.stream(stream):
    sbv.key_cache(False)
    out["first_batch_ms_with_pool_allocation"] = step()
    out["cold_ms"] = med(steps)
    sbv.key_cache(True)
    sbv.hot_keys(1024, 0xFFFFFFFF)
    step()
    out["warm_8bit_ms"] = med(steps)
    sbv.hot_keys(1024, 4096)
    ramp = []
    for _ in range(300):
        ramp.append(step())
        if sbv.hot_key_stats()[0] >= min(K, 1024):
            break
    out["ramp_batches"] = len(ramp)
    out["ramp_ms"] = [round(x, 2) for x in ramp]
    step()
    out["hot_ms"] = med(steps)
    out["hot_stats"] = sbv.hot_key_stats()
    out["bitmap_correct"] = bool((d_b.cpu().numpy() == valid).all())
    p = out["hot_stats"][0]
    out["selfcheck"] = [bool(sbv.hot_selfcheck(i)) for i in sorted({0, p // 2, max(0, p - 1)})] if p else []
    for k in ("cold_ms", "warm_8bit_ms", "hot_ms"):
        out[k.replace("_ms", "_M_per_s")] = n / out[k] / 1e3
print(json.dumps(out))
