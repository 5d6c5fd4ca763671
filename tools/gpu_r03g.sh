#!/bin/bash
# round 3, session g: latency of the small path after the PCIe fix; rocprofv3 kernel stats, timeline and HBM traffic
# (FETCH_SIZE / WRITE_SIZE in their own passes) of the committed library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03g
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for small in 1 0; do SBV_SMALL=$small timeout 100 python tools/latency_small.py 2>/dev/null | tee -a "$OUT/latency_small.jsonl"; done
( timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --primary-only --warm-leg > "$OUT/bench_primary.json" 2> "$OUT/bench_primary.err" ); python3 -c "
import json
d=json.load(open('$OUT/bench_primary.json')); print({k: d[k] for k in ('value','ms_per_step','bitmap_correct')}, d['kernel_us'], d.get('warm_key_cache',{}).get('ms_per_step'))"
cd /tmp
run() { local name=$1; shift; ( timeout 200 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/$name.log" 2>&1; echo "rc=$?" >> "$OUT/$name.log" ); tail -1 "$OUT/$name.log"; }
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
for k in summary.get("kernel_stats", [])[:8]: print(k)
PY
cp "$OUT/stats/p_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
