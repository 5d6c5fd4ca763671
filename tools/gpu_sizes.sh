#!/bin/bash
# bench.py at several batch sizes (primary leg only), one JSON summary line each; every step bounded.
set -u
TAG=${1:-sizes}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for lg in 16 18 19 20 21; do
  n=$((1 << lg))
  ( timeout 150 python bench.py --steps 5 --warmup 2 --tuples $n --no-cpu-baseline --primary-only > "$OUT/bench_2p$lg.log" 2>&1; echo "rc=$?" >> "$OUT/bench_2p$lg.log" )
  python3 - "$OUT/bench_2p$lg.log" $lg <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(json.dumps({"log2_tuples": int(sys.argv[2]), "value_M_per_s": round(d["value"] / 1e6, 1), "ms_per_step": round(d["ms_per_step"], 3),
                          "bitmap_correct": d["bitmap_correct"], "grouped": d["key_grouping"]["enabled"], "groups": d["key_grouping"]["groups"]}))
        break
else:
    print("2^%s: no result" % sys.argv[2], open(sys.argv[1]).read()[-300:])
PY
done
