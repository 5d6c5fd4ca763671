#!/bin/bash
# round 2, step h: multi-device entry (world of one + forced RCCL), full gpu suite, default bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02h
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q --durations=6 > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" ); tail -14 "$OUT/pytest_gpu.log"
( timeout 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?" >> "$OUT/bench_default.err" ); tail -2 "$OUT/bench_default.err"; cut -c1-400 "$OUT/bench_default.json"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("int_mul_issue_fraction", "all_valid", "end_to_end", "sharded_entry", "ed25519", "m2_commit_quorum_us", "without_key_grouping", "registered_key_path", "cpu_baseline"):
    print(k, json.dumps(d.get(k))[:600])
PY
