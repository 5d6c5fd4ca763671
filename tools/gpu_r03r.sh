#!/bin/bash
# round 3, session r: the two wide forms of the table builder on their own (rows: one lane per entry; fill: rows split over
# 2 / 3 lanes), knobs reset between variants
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03r
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 300 python tools/ab_env.py 17,18,20 default SBV_GROUP_WIDE=1 SBV_GROUP_WIDE=2,SBV_GROUP_FSPLIT=3 SBV_GROUP_WIDE=2,SBV_GROUP_FSPLIT=2 SBV_GROUP_WIDE=3,SBV_GROUP_FSPLIT=3 > "$OUT/ab_wide_bits.jsonl" 2> "$OUT/ab_wide_bits.err"; echo "rc=$?" >> "$OUT/ab_wide_bits.err" ); cat "$OUT/ab_wide_bits.jsonl"; tail -1 "$OUT/ab_wide_bits.err"
