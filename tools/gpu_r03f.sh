#!/bin/bash
# round 3, session f: full GPU suite after the Ed25519 / key-cache fix and with the key-affine tests; latency of the small path A/B;
# default bench (projection leg)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03f
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 800 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" ); grep -n "passed\|failed\|FAILED\|rc=" "$OUT/pytest_gpu.log" | tail -8
for small in 1 0; do SBV_SMALL=$small timeout 100 python tools/latency_small.py 2>/dev/null | tee -a "$OUT/latency_small.jsonl"; done
( timeout 420 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?" >> "$OUT/bench_default.err" ); tail -2 "$OUT/bench_default.err"
python3 - "$OUT/bench_default.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench line", e); sys.exit(0)
print({k: d[k] for k in ("value", "ms_per_step", "bitmap_correct")}, d["kernel_us"])
for k in ("projected_strong_scaling", "m2_commit_quorum_us"):
    print(k, json.dumps(d.get(k))[:1800])
PY
