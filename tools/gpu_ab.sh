#!/bin/bash
# A/B of the grouped step's pipeline knobs (chunks of windows x lanes per window); every step bounded.
set -u
TAG=${1:-ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "group" > "$OUT/pytest_group.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_group.log" ); tail -3 "$OUT/pytest_group.log"
for cfg in "2 4 8" "3 4 8" "2 8 8" "2 4 4" "2 4 16" "1 4 8"; do
  set -- $cfg
  ( SBV_GROUP_CHUNKS=$1 SBV_GROUP_PARTS=$2 SBV_PREP_T=${3:-32} SBV_GENERIC_STREAM=${4:-0} GPU_MAX_HW_QUEUES=${5:-4} timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/bench_c$1_p$2_t${3:-32}_g${4:-0}_q${5:-4}.log" 2>&1; echo "rc=$?" >> "$OUT/bench_c$1_p$2_t${3:-32}_g${4:-0}_q${5:-4}.log" )
  python - "$OUT/bench_c$1_p$2_t${3:-32}_g${4:-0}_q${5:-4}.log" "$1" "$2/T${3:-32}/gs${4:-0}/hwq${5:-4}" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("chunks", sys.argv[2], "parts", sys.argv[3], "value %.1fM" % (d["value"] / 1e6), "ms %.2f" % d["ms_per_step"], "ok", d["bitmap_correct"], d["kernel_us"], d["key_grouping"]["groups"])
        break
else:
    print("chunks", sys.argv[2], "parts", sys.argv[3], "NO RESULT", open(sys.argv[1]).read()[-400:])
PY
done
cd /tmp
( SBV_GROUP_CHUNKS=2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/stats.log" 2>&1; echo "rc=$?" >> "$OUT/stats.log" )
cp "$OUT/stats/p_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
python - "$OUT/stats/p_kernel_trace.csv" <<'PY'
import csv, sys, collections
try:
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sbv::" in r["Kernel_Name"]]
except Exception as e:
    print("no trace", e); sys.exit(0)
# timeline of the LAST step: kernels after the last k_p256_prep start
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_p256_prep" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last - 3:]:
    print("%-28s start %8.3f ms  end %8.3f ms" % (r["Kernel_Name"].split("(")[0].replace("sbv::", ""), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
PY
rm -rf "$OUT/stats"
cat "$OUT/kernel_stats.csv" | cut -c1-160 | head -20
