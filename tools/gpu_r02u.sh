#!/bin/bash
# round 2, step u: Ed25519 kernel profile (configs[4])
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02u5
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
cat > /tmp/ed_prof.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench, consensus_amd as sbv
sbv.init(0)
print(bench.leg_ed25519(sbv, torch, 1 << 20, 4, torch.cuda.Stream()))
PY
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/edprof" -o ed -- python /tmp/ed_prof.py > "$OUT/edprof.log" 2>&1; echo "rc=$?" >> "$OUT/edprof.log" )
f=$(find "$OUT/edprof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-220
t=$(find "$OUT/edprof" -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "ed" in r["Kernel_Name"].lower() or "sbv" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: the final 12 kernels
last = rows[-14:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-40s start %8.3f ms  end %8.3f ms  grid %s wg %s" % (r["Kernel_Name"].split("(")[0][-40:], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
PY
find "$OUT/edprof" -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
