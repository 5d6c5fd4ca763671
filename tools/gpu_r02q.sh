#!/bin/bash
# round 2, step q: comb kernels at 2 vs 3 waves/SIMD after the fused reductions
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02q
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for v in w2 w3 w2 w3; do
  lib=$ROOT/consensus_amd/libsbv.so; [ $v = w3 ] && lib=$ROOT/consensus_amd/libsbv_w3.so
  ( SBV_LIB=$lib timeout 150 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --primary-only >> "$OUT/bench_$v.log" 2>&1; echo "rc=$?" >> "$OUT/bench_$v.log" )
done
python - "$OUT" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "value %.1fM" % (d["value"] / 1e6), "ms %.3f" % d["ms_per_step"], "ok", d["bitmap_correct"], {k: round(v) for k, v in d["kernel_us"].items()})
PY
