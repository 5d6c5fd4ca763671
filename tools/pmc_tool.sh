#!/bin/bash
# Per-kernel means of a PMC counter set over one run of a tools/ script: pmc_tool.sh <tag> <counters...> -- <script.py> [args]   (counters only: no tracing in the same run)
TAG=$1; shift; CNT=(); while [ "$1" != "--" ]; do CNT+=("$1"); shift; done; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp; timeout 600 rocprofv3 --pmc "${CNT[@]}" --output-format csv -d "$OUT/pmc" -o p -- python "$ROOT/tools/$1" "${@:2}" > "$OUT/pmc.log" 2>&1 )
python3 - "$OUT" <<'PY'
import csv, collections, json, os, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(out, "pmc", "p_counter_collection.csv"))):
    if "sbv::" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0].replace("sbv::", "").replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
summary = {}
for (k, c), v in sorted(agg.items()):
    summary.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "max": max(v), "dispatches": len(v)}
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for k, v in summary.items():
    print(k, {c: (round(x["mean"], 1), round(x["max"], 1), x["dispatches"]) for c, x in v.items()})
PY
rm -rf "$OUT/pmc"
