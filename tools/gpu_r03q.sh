#!/bin/bash
# round 3, session q: rows of 16 entries per fill lane (SBV_GROUP_PARTS = 1, 2, 4, 7) with the knobs reset between variants
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03q
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 300 python tools/ab_env.py 17,18,20 default SBV_GROUP_PARTS=2 SBV_GROUP_PARTS=4 SBV_GROUP_PARTS=7 > "$OUT/ab_parts.jsonl" 2> "$OUT/ab_parts.err"; echo "rc=$?" >> "$OUT/ab_parts.err" ); cat "$OUT/ab_parts.jsonl"; tail -1 "$OUT/ab_parts.err"
