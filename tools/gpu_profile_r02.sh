#!/bin/bash
# rocprofv3 collection for the round's profiles/: kernel-trace stats, then PMC in separate passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; never combined with sys/hip traces).
set -u
TAG=${1:-r02p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/bench_unprofiled.json" 2> /dev/null   # also warms the /tmp batch cache
cd /tmp
run() { local name=$1; shift; ( timeout 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/$name.log" 2>&1; echo "rc=$?" >> "$OUT/$name.log" ); tail -1 "$OUT/$name.log"; }
run stats --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, collections, json, os, sys
out = sys.argv[1]
summary = {}
st = os.path.join(out, "stats", "p_kernel_stats.csv")
if os.path.exists(st):
    summary["kernel_stats"] = [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in csv.DictReader(open(st)) if "sbv::" in r["Name"]]
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
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
print(json.dumps(summary)[:1500])
PY
# keep only the small artefacts
cp "$OUT/stats/p_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq"
ls -la "$OUT"
