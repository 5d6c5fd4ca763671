#!/bin/bash
# round 2, step d: new full-size config tests + the extended bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02d
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_host.py -m gpu -x -q --durations=8 > "$OUT/pytest_configs.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_configs.log" ); tail -16 "$OUT/pytest_configs.log"
( timeout 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?" >> "$OUT/bench_default.err" ); tail -2 "$OUT/bench_default.err"; cut -c1-3000 "$OUT/bench_default.json"
