#!/usr/bin/env python3
"""bench.py's key_count_sweep leg on its own (dev sessions): 2^20 tuples over 256 ... all-distinct keys, cold and warm.
usage: key_sweep.py [log2 n] [K,K,...]   Prints the leg's JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
import consensus_amd as sbv  # noqa: E402
import synth  # noqa: E402

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
sbv.init(0)
sbv.key_cache(False)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    res = bench.leg_key_count_sweep(sbv, synth, torch, n, stream, ks)
for p in res["points"]:
    print("keys %8d  cold %7.1f M/s %7.2f ms (groups %6d tables %8d one-lane %8d)  warm %7.1f M/s %7.2f ms (groups %6d tables %8d)  ok %s" % (
        p["keys"], p["cold"]["verifies_per_s"] / 1e6, p["cold"]["ms"], p["cold"]["groups"], p["cold"]["tuples_through_tables"], p["cold"]["tuples_one_lane_kernel"],
        p["warm"]["verifies_per_s"] / 1e6, p["warm"]["ms"], p["warm"]["groups"], p["warm"]["tuples_through_tables"],
        p["cold"]["bitmap_correct"] and p["warm"]["bitmap_correct"]), file=sys.stderr)
    if "hot" in p:
        print("              hot  %7.1f M/s %7.2f ms (promoted %5d, wide-pass tuples %8d)  ok %s" % (
            p["hot"]["verifies_per_s"] / 1e6, p["hot"]["ms"], p["hot"]["promoted_keys"], p["hot"]["tuples_through_the_wide_pass"], p["hot"]["bitmap_correct"]), file=sys.stderr)
print(json.dumps(res))
