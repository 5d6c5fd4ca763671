#!/usr/bin/env python3
"""Kernel timeline of ONE cold Ed25519 grouped step (key-table cache off) from a rocprofv3 kernel trace of this very script.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python tools/ed_cold_timeline.py run ; python tools/ed_cold_timeline.py show DIR/.../p_kernel_trace.csv"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def run():
    import torch
    import bench
    import consensus_amd as sbv
    sbv.init(0)
    r = bench.leg_ed25519(sbv, torch, 1 << 20, 4, torch.cuda.current_stream(), False)
    print(r["ms_per_step"], r["bitmap_correct"])


def show(path):
    rows = [r for r in csv.DictReader(open(path)) if "sbv::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    inserts = [i for i, r in enumerate(rows) if "k_ed_group_insert" in r["Kernel_Name"]]
    # the cold steps come first (the leg's warm part follows): take the third cold step
    start = inserts[2]
    end = inserts[3]
    t0 = int(rows[start]["Start_Timestamp"])
    # stage-A-less scheme: the G phase of this step may have started before the insert kernel
    for r in rows[max(0, start - 3):end]:
        name = r["Kernel_Name"].split("(")[0].replace("sbv::", "").replace("void ", "")
        print("%-34s start %8.3f ms  end %8.3f ms" % (name, (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        show(sys.argv[2])
