#!/bin/bash
# round 3, session l: fresh-process cold / warm steps of the two table builders (session k saw 3.83 ms in a fresh process and
# 3.47 ms after a re-init for the wide form), and the kernel trace of a cold 2^18 / 2^20 step of the wide form
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03l
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for rep in 1 2; do for w in 1 0; do
  echo "# SBV_GROUP_WIDE=$w rep $rep" >> "$OUT/fresh.jsonl"
  SBV_GROUP_WIDE=$w timeout 120 python tools/sweep_sizes.py 18 20 >> "$OUT/fresh.jsonl" 2>> "$OUT/fresh.err"
done; done
python3 - "$OUT/fresh.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"): print(l.strip()); continue
    d = json.loads(l); print(d["log2_tuples"], d["cold"]["ms"], d["warm"]["ms"])
PY
for w in 1 0; do
( cd /tmp && SBV_GROUP_WIDE=$w timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_w$w" -o p -- python "$ROOT/tools/sweep_sizes.py" 18 20 > "$OUT/trace_w$w.jsonl" 2> "$OUT/trace_w$w.err" )
f=$(find "$OUT/trace_w$w" -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > "$OUT/timeline_w$w.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the cold steps are the first timed loops of each size: print two consecutive steps starting at the 3rd k_p256_prep, for each grid size of prep
seen = {}
out = []
for i, r in enumerate(rows):
    if "k_p256_prep" in r["Kernel_Name"]:
        key = r.get("Grid_Size") or r.get("Grid_Size_X")
        seen[key] = seen.get(key, 0) + 1
        if seen[key] == 3:
            t0 = int(r["Start_Timestamp"])
            j = i
            preps = 0
            while j < len(rows):
                if "k_p256_prep" in rows[j]["Kernel_Name"]:
                    preps += 1
                    if preps == 3: break
                out.append("%s %9.3f %9.3f  %s" % (key, (int(rows[j]["Start_Timestamp"]) - t0) / 1e6, (int(rows[j]["End_Timestamp"]) - t0) / 1e6, rows[j]["Kernel_Name"].split("(")[0][:50]))
                j += 1
            out.append("")
print("\n".join(out))
PY
find "$OUT/trace_w$w" -type f ! -name "*.txt" -delete 2>/dev/null
done
head -70 "$OUT/timeline_w1.txt"
