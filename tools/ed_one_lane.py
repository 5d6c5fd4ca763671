#!/usr/bin/env python3
"""One JSON line: the Ed25519 one-lane kernel over the whole 2^20 batch of bench.py's Ed25519 leg (grouping off), ms per step (SBV_LIB selects the library build)."""
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
    n = 1 << 20
    h = hostlib.load()
    tuples = np.zeros(n * 128, dtype=np.uint8)
    expect = np.zeros((n + 7) // 8, dtype=np.uint8)
    h.sbvh_ed25519_gen_batch(bench.SEED, n, 1024, 8, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)
    sbv.init(0)
    stream = torch.cuda.current_stream()
    d_t = torch.from_numpy(tuples).cuda()
    d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
    sbv.set_grouping(False, 0, 0, 0)
    call = lambda: sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)      # noqa: E731
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(json.dumps({"lib": os.path.basename(sbv.LIB_PATH), "one_lane_ms": 1e3 * dt, "verifies_per_s": n / dt,
                      "bitmap_correct": bool((d_b.cpu().numpy() == expect).all())}), flush=True)


if __name__ == "__main__":
    main()
