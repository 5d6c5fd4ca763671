#!/bin/bash
# Second-generation GPU round: every step has its own timeout and log; nothing can eat the budget.
set -u
TAG=${1:-r01x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
step() { local name=$1 limit=$2; shift 2; echo "== $name"; ( timeout "$limit" "$@" > "$OUT/$name.log" 2>&1; echo "rc=$?" >> "$OUT/$name.log" ); tail -${TAILN:-4} "$OUT/$name.log"; }
step devcheck 60 tools/devcheck tests/golden/devcheck_tuples.bin
TAILN=6 step pytest_gpu 600 python -m pytest tests -m gpu -x -q -s
step bench 240 python bench.py --steps 5 --warmup 2
cp "$OUT/bench.log" "$OUT/bench.json" 2>/dev/null
step bench_ed25519 240 python tools/bench_ed25519.py
step replay 300 python tools/replay_bench.py
echo "== done"
