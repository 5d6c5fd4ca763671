#!/bin/bash
# round 3, session n: tuples per lane of stage A (Montgomery's trick along the lane's chunk, SBV_PREP_T): kernel duration and
# step time at 2^20, one process per value (the value is read once per process)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03n
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for T in 8 4 2 16 6; do
  ( cd /tmp && SBV_PREP_T=$T timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t$T" -o p -- python "$ROOT/tools/sweep_sizes.py" 20 > "$OUT/t$T.jsonl" 2> "$OUT/t$T.err" )
  f=$(find "$OUT/t$T" -name "*kernel_stats.csv" | head -1)
  echo "== SBV_PREP_T=$T" | tee -a "$OUT/prep_T.txt"
  python3 -c "
import json,sys
d=json.loads(open('$OUT/t$T.jsonl').read().strip().split('\n')[-1]); print('cold', d['cold']['ms'], 'warm', d['warm']['ms'], d['cold']['ok'], d['warm']['ok'])" | tee -a "$OUT/prep_T.txt"
  [ -n "$f" ] && grep -E "k_p256_prep|k_gphase_generic|k_verify_keyed_q|k_group_insert" "$f" | cut -d, -f1-6 | tee -a "$OUT/prep_T.txt"
  rm -rf "$OUT/t$T"
done
