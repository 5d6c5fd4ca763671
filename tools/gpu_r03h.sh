#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03h
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
( SBV_SMALL=1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st" -o p -- python "$ROOT/tools/latency_small.py" > "$OUT/lat.log" 2>&1 )
grep small_path "$OUT/lat.log"
python3 - "$OUT/st/p_kernel_trace.csv" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sbv::" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"].split("(")[0], r.get("Grid_Size") or r.get("Grid_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    v.sort()
    print(k, "launches", len(v), "median us", round(v[len(v) // 2], 1), "min", round(v[0], 1), "max", round(v[-1], 1))
PY
rm -rf "$OUT/st"
