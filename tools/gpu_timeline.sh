#!/bin/bash
# rocprofv3 kernel trace of one command; prints the kernel timeline of its LAST step (from the last launch of
# $ANCHOR on) and the per-kernel stats.  usage: gpu_timeline.sh <tag> <anchor-kernel-substring> <command...>
set -u
TAG=$1; ANCHOR=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
( timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- "$@" > "$OUT/stats.log" 2>&1; echo "rc=$?" >> "$OUT/stats.log" )
cp "$OUT/stats/p_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
python3 - "$OUT/stats/p_kernel_trace.csv" "$ANCHOR" <<'PY'
import csv, sys
try:
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sbv::" in r["Kernel_Name"]]
except Exception as e:
    print("no trace", e); sys.exit(0)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
if not idx:
    print("anchor not found"); sys.exit(0)
# last step = from the last anchor launch that is followed by at least 3 kernels
last = idx[-1]
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%-28s start %8.3f ms  end %8.3f ms" % (r["Kernel_Name"].split("(")[0].replace("sbv::", ""), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
PY
rm -rf "$OUT/stats"
cut -c1-150 "$OUT/kernel_stats.csv" | head -16
