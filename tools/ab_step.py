#!/usr/bin/env python3
"""One JSON line: the 2^20-tuple headline step cold (key cache off), warm (8-bit tables cached) and hot (16-bit combs promoted), ms per
step, with the dominant kernel's HIP-event duration of the cold step.  No arguments; SBV_LIB selects the library build (tools/ab_lib.sh
alternates two builds in fresh processes).  AB_SIZES=20,18 adds other sizes (cold / warm only)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def timed(sbv, torch, d, n, d_b, stream, steps):
    sbv.verify_batch_dev(d.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sbv.verify_batch_dev(d.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    import torch
    import consensus_amd as sbv
    import synth
    steps = int(os.environ.get("AB_STEPS", "20"))
    tuples, valid = synth.gen_batch(0x5B7F2026, 1 << 20)
    d = torch.from_numpy(tuples).cuda()
    stream = torch.cuda.current_stream()
    sbv.init(0)
    out = {"lib": os.path.basename(sbv.LIB_PATH)}
    for lg in [int(x) for x in os.environ.get("AB_SIZES", "20").split(",")]:
        n = 1 << lg
        d_b = torch.zeros(n // 8, dtype=torch.uint8, device="cuda")
        sbv.key_cache(False)
        for _ in range(3):
            sbv.verify_batch_dev(d.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        sbv.profile_enable(2)
        cold = timed(sbv, torch, d, n, d_b, stream, steps)
        dom_us, dom_n = sbv.profile_read_dominant()
        sbv.profile_enable(False)
        ok = bool((d_b.cpu().numpy() == valid[:n // 8]).all())
        sbv.key_cache(True)
        sbv.hot_keys(1024, 0xFFFFFFFF)
        warm = timed(sbv, torch, d, n, d_b, stream, steps)
        ok = ok and bool((d_b.cpu().numpy() == valid[:n // 8]).all())
        row = {"cold_ms": round(cold, 4), "q_launch_us": round(dom_us / max(1, dom_n), 1), "warm_ms": round(warm, 4)}
        if lg == 20 and os.environ.get("AB_HOT", "1") != "0":
            sbv.hot_keys(1024, 4096)
            last, still = -1, 0
            for _ in range(60):
                sbv.verify_batch_dev(d.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                promoted = sbv.hot_key_stats()[0]
                still = still + 1 if promoted == last and promoted else 0
                last = promoted
                if still >= 3:
                    break
            row["hot_ms"] = round(timed(sbv, torch, d, n, d_b, stream, steps), 4)
            row["promoted"] = last
            ok = ok and bool((d_b.cpu().numpy() == valid[:n // 8]).all())
        row["ok"] = ok
        out[f"2^{lg}"] = row
        sbv.key_cache(False)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
