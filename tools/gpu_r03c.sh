#!/bin/bash
# round 3, session c: GPU suite after the limb-wrap fix; A/B of table pieces per Q chunk, fill rows per lane, Q chunks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03c
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" ); tail -4 "$OUT/pytest_gpu.log"
( timeout 400 python tools/ab_env.py 18,20 SBV_GROUP_TSUB=1 SBV_GROUP_TSUB=2 SBV_GROUP_TSUB=3 SBV_GROUP_TSUB=2,SBV_GROUP_PARTS=2 SBV_GROUP_TSUB=2,SBV_GROUP_CHUNKS=3 SBV_GROUP_TSUB=4,SBV_GROUP_CHUNKS=1 SBV_GROUP_TSUB=2,SBV_SIDE_PRIO=1 > "$OUT/ab_env.jsonl" 2> "$OUT/ab_env.err"; echo "rc=$?" >> "$OUT/ab_env.err" ); cat "$OUT/ab_env.jsonl"; tail -2 "$OUT/ab_env.err"
