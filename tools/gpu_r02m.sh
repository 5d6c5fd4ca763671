#!/bin/bash
# round 2, step m: persistent key-table cache: parity, warm vs cold
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02m
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log" ); tail -6 "$OUT/pytest.log"
( timeout 400 python bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$?" >> "$OUT/bench.err" ); tail -2 "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1fM ms %.3f ok %s" % (d["value"] / 1e6, d["ms_per_step"], d["bitmap_correct"]), d["kernel_us"])
for k in ("warm_key_cache", "int_mul_issue_fraction", "all_valid", "end_to_end", "sharded_entry", "ed25519", "m2_commit_quorum_us", "without_key_grouping", "registered_key_path"):
    print(k, json.dumps(d.get(k))[:700])
PY
