#!/bin/bash
# round 2, step ac: two-pass split (classify + key check on the compacted candidates); Ed25519 key-sorted step (tuple-major
# accumulator records).  Parity subset, then A/B against SBV_GROUP_SORT=0 for both curves.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ac
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ed25519.py -m gpu -x -q -k "grouping or cache or grouped or ed25519" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log" ); tail -3 "$OUT/pytest.log"
grep -q "rc=0" "$OUT/pytest.log" || exit 1
for sort in 1 0 1 0; do
  SBV_GROUP_SORT=$sort timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --primary-only --warm-leg 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'curve': 'p256', 'sort': $sort, 'ms_per_step': round(d['ms_per_step'],4), 'value': round(d['value']), 'ok': d['bitmap_correct'], 'q_us': round(d['kernel_us']['k_verify_keyed_q'],1), 'warm_ms': round(d.get('warm_key_cache',{}).get('ms_per_step',0),4)}))" | tee -a "$OUT/ab.jsonl"
done
for sort in 1 0 1 0; do
  SBV_BENCH_PRIMARY_ONLY=1 SBV_GROUP_SORT=$sort timeout 200 python tools/bench_ed25519.py 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'curve': 'ed25519', 'sort': $sort, 'ms_per_step': round(d['ms_per_step'],4), 'value': round(d['value']), 'ok': d['bitmap_correct'], 'grouping': d['key_grouping']}))" | tee -a "$OUT/ab.jsonl"
done
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/stats.log" 2>&1; echo "rc=$?" >> "$OUT/stats.log" ); tail -1 "$OUT/stats.log"
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, os, sys
out = sys.argv[1]
f = os.path.join(out, "stats", "p_kernel_trace.csv")
if os.path.exists(f):
    rows = [r for r in csv.DictReader(open(f)) if "sbv::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    packs = [i for i, r in enumerate(rows) if "k_pack_bitmap" in r["Kernel_Name"]]
    if len(packs) >= 2:
        seg = rows[packs[-2] + 1:packs[-1] + 1]
        t0 = int(seg[0]["Start_Timestamp"])
        with open(os.path.join(out, "timeline.txt"), "w") as fh:
            for r in seg:
                fh.write("%8.3f %8.3f  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r["Kernel_Name"].split("(")[0]))
        print(open(os.path.join(out, "timeline.txt")).read())
PY
cp "$OUT/stats/p_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/stats"
