#!/bin/bash
# round 2, step y: full GPU suite, devcheck, default-flags bench with the final library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02y
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 120 tools/devcheck tests/golden/devcheck_tuples.bin > "$OUT/devcheck.log" 2>&1; echo "rc=$?" >> "$OUT/devcheck.log" ); tail -2 "$OUT/devcheck.log"
( timeout 900 python -m pytest tests -m gpu -x -q -s > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" )
grep -E "passed|failed|rc=|\[sign\]" "$OUT/pytest_gpu.log" | tail -4
( timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?" >> "$OUT/bench_default.err" )
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f M/s ms %.3f ok %s roofline %s" % (d["value"] / 1e6, d["ms_per_step"], d.get("bitmap_correct"), d.get("roofline")))
for k in ("warm_key_cache", "all_valid", "end_to_end", "sharded_entry", "ed25519", "m2_commit_quorum_us", "registered_keys", "grouping_off"):
    v = d.get(k)
    if isinstance(v, dict):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, list, str))})
print("cpu", d.get("cpu_baseline"))
PY
