#!/bin/bash
# round 3, session t: the G phase as its own kernel at 2 / 3 / 4 waves per SIMD (and the Q phase at 3), compile-time variants
# of the library built beside the product (consensus_amd/libsbv_v*.so), fresh process per run
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03t
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for rep in 1 2; do for v in "" _v1 _v2 _v3 _v4; do
  echo "# libsbv$v rep $rep" >> "$OUT/variants.jsonl"
  timeout 100 python tools/with_lib.py consensus_amd/libsbv$v.so tools/sweep_sizes.py 18 20 >> "$OUT/variants.jsonl" 2>> "$OUT/variants.err"
done; done
python3 - "$OUT/variants.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"): print(l.strip()); continue
    d = json.loads(l); print(d["log2_tuples"], "cold", d["cold"]["ms"], "warm", d["warm"]["ms"], d["cold"]["ok"], d["warm"]["ok"])
PY
