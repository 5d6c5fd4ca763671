#!/bin/bash
# round 3, session p: the second table stream = the context's idle stream (no fifth stream): fresh processes, both settings
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03p
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for rep in 1 2; do for t in 2 1; do
  echo "# SBV_GROUP_TSTREAMS=$t rep $rep" >> "$OUT/fresh.jsonl"
  SBV_GROUP_TSTREAMS=$t timeout 120 python tools/sweep_sizes.py 17 18 19 20 >> "$OUT/fresh.jsonl" 2>> "$OUT/fresh.err"
done; done
python3 - "$OUT/fresh.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"): print(l.strip()); continue
    d = json.loads(l); print(d["log2_tuples"], d["cold"]["ms"], d["warm"]["ms"], d["cold"]["generic"])
PY
for t in 2 1; do
( SBV_GROUP_TSTREAMS=$t timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only --warm-leg > "$OUT/bench_primary_t$t.json" 2> "$OUT/bench_primary_t$t.err" ); python3 -c "
import json
d=json.load(open('$OUT/bench_primary_t$t.json')); print('tstreams $t', {k: d[k] for k in ('value','ms_per_step','bitmap_correct')}, d['kernel_us'], d.get('warm_key_cache',{}).get('ms_per_step'))"
done
