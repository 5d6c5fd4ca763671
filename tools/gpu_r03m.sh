#!/bin/bash
# round 3, session m: table-building schedule A/B with the knobs reset between variants (sessions c and k left tsub / wide
# stuck at the previous variant's value): a stream per chunk for rows + fill, 3 chunks, CU-masked side streams
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03m
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 400 python tools/ab_env.py 17,18,20 default SBV_GROUP_TSTREAMS=2 SBV_GROUP_CHUNKS=3 SBV_GROUP_CHUNKS=3,SBV_GROUP_TSTREAMS=3 SBV_GROUP_CHUNKS=4,SBV_GROUP_TSTREAMS=4 SBV_TABLE_CUS=64 SBV_TABLE_CUS=128 SBV_GROUP_TSTREAMS=2,SBV_TABLE_CUS=96 SBV_GROUP_TSUB=2 SBV_GROUP_TSUB=2,SBV_GROUP_TSTREAMS=2 > "$OUT/ab_sched.jsonl" 2> "$OUT/ab_sched.err"; echo "rc=$?" >> "$OUT/ab_sched.err" ); cat "$OUT/ab_sched.jsonl"; tail -2 "$OUT/ab_sched.err"
