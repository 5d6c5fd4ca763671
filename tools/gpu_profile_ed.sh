#!/bin/bash
# PMC counters of the Ed25519 grouped step (configs[4]); separate passes, no trace domains combined with --pmc
set -u
TAG=${1:-r02w}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"
cat > /tmp/ed_leg.py <<'PY'
import sys, torch, json
sys.path.insert(0, sys.argv[1])
import bench, consensus_amd as sbv
sbv.init(0)
print(json.dumps(bench.leg_ed25519(sbv, torch, 1 << 20, 3, torch.cuda.Stream())))
PY
python /tmp/ed_leg.py "$ROOT" > "$OUT/unprofiled.json" 2>/dev/null
cd /tmp
run() { local name=$1; shift; ( timeout 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python /tmp/ed_leg.py "$ROOT" > "$OUT/$name.log" 2>&1; echo "rc=$?" >> "$OUT/$name.log" ); tail -1 "$OUT/$name.log"; }
run pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
run pmc_mem --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT
run pmc_fetch --pmc FETCH_SIZE
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, collections, json, os, sys
out = sys.argv[1]
summary = {}
for d in ("pmc_sq", "pmc_mem", "pmc_fetch"):
    f = os.path.join(out, d, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sbv::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("sbv::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        summary.setdefault(k, {})[c] = round(sum(v) / len(v))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in summary.items():
    print(k, v)
PY
rm -rf "$OUT/pmc_sq" "$OUT/pmc_mem" "$OUT/pmc_fetch"
