#!/bin/bash
# round 3, session v: the rows kernel at 168 VGPRs (3 waves per SIMD) beside the 3-wave G phase, compile-time variant
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03v
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for rep in 1 2; do for v in "" _r3; do
  echo "# libsbv$v rep $rep" >> "$OUT/variants.jsonl"
  timeout 60 python tools/with_lib.py consensus_amd/libsbv$v.so tools/sweep_sizes.py 18 20 >> "$OUT/variants.jsonl" 2>> "$OUT/variants.err"
done; done
python3 - "$OUT/variants.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"): print(l.strip()); continue
    d = json.loads(l); print(d["log2_tuples"], "cold", d["cold"]["ms"], "warm", d["warm"]["ms"], d["cold"]["ok"], d["warm"]["ok"])
PY
