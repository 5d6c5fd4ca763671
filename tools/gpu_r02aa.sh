#!/bin/bash
# round 2, step aa: key-sorted grouped list + XCD-aware Q phase: parity subset, A/B against the compaction order, kernel
# stats and HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the sorted form
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02aa
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "grouping or 2_20 or config" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log" ); tail -4 "$OUT/pytest.log"
grep -q "rc=0" "$OUT/pytest.log" || exit 1
for sort in 0 1 0 1; do
  SBV_GROUP_SORT=$sort timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --primary-only --warm-leg 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'sort': $sort, 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'ok': d['bitmap_correct'], 'kernel_us': d['kernel_us'], 'warm_ms': d.get('warm_key_cache',{}).get('ms_per_step')}))" | tee -a "$OUT/ab.jsonl"
done
cd /tmp
run() { local name=$1; shift; ( timeout 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/$name.log" 2>&1; echo "rc=$?" >> "$OUT/$name.log" ); tail -1 "$OUT/$name.log"; }
run stats --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, collections, json, os, sys
out = sys.argv[1]
summary = {}
st = os.path.join(out, "stats", "p_kernel_stats.csv")
if os.path.exists(st):
    summary["kernel_stats"] = [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in csv.DictReader(open(st)) if "sbv::" in r["Name"]]
for d in ("pmc_fetch", "pmc_write"):
    f = os.path.join(out, d, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sbv::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("sbv::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        summary.setdefault("pmc", {}).setdefault(k, {})[c] = {"mean": sum(v) / len(v), "dispatches": len(v)}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary.get("pmc", {}))[:3000])
PY
# kernel timeline of one step (start offsets relative to the step's first kernel), from the kernel trace
python - "$OUT" <<'PY'
import csv, os, sys
out = sys.argv[1]
f = os.path.join(out, "stats", "p_kernel_trace.csv")
if os.path.exists(f):
    rows = [r for r in csv.DictReader(open(f)) if "sbv::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last step: walk back from the last k_pack_bitmap to the previous one
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
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
