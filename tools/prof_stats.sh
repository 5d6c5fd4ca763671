#!/bin/bash
# per-kernel stats of one bench invocation, parsed properly (kernel names contain commas)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-prof}; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"; timeout 100 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st" -o p -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/prof.log" 2>&1
python - "$OUT" <<'PY'
import csv, sys, os
out = sys.argv[1]
rows = list(csv.DictReader(open(os.path.join(out, "st", "p_kernel_stats.csv"))))
with open(os.path.join(out, "kernel_stats.txt"), "w") as f:
    for r in rows:
        line = f'{r["Name"].split("(")[0][:40]:40s} calls={int(r["Calls"]):3d} avg_us={float(r["AverageNs"])/1e3:10.1f} pct={float(r["Percentage"]):6.2f}'
        print(line); f.write(line + "\n")
PY
grep '^{' "$OUT/prof.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['kernel_us'], d['bitmap_correct'], 'ungrouped', round(d['without_key_grouping']['value']/1e6,2), 'keyed', d.get('registered_key_path',{}).get('value'))"
rm -rf "$OUT/st"
