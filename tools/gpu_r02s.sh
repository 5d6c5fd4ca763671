#!/bin/bash
# round 2, step s: chunks of the key-comb pipeline (1..4) by batch size, cold and warm cache
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02s
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for n in 262144 524288 1048576; do
for ch in 1 2 3 4; do
  ( SBV_GROUP_CHUNKS=$ch timeout 150 python bench.py --steps 8 --warmup 2 --tuples $n --no-cpu-baseline --primary-only --warm-leg >> "$OUT/bench_${n}_c$ch.log" 2>&1; echo "rc=$?" >> "$OUT/bench_${n}_c$ch.log" )
done
done
python - "$OUT" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "value %.1fM" % (d["value"] / 1e6), "ms %.3f" % d["ms_per_step"], "ok", d["bitmap_correct"], "warm", (d.get("warm_key_cache") or {}).get("value"))
PY
