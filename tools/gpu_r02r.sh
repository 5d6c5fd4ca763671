#!/bin/bash
# round 2, step r: wave priority of the table kernels (0 / 1 / 3), at 2^20 and 2^18; devcheck
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02r
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 120 tools/devcheck tests/golden/devcheck_tuples.bin > "$OUT/devcheck.log" 2>&1; echo "rc=$?" >> "$OUT/devcheck.log" )
for rep in 1 2; do
for v in p0 p1 p3; do
  lib=$ROOT/consensus_amd/libsbv.so; [ $v = p1 ] && lib=$ROOT/consensus_amd/libsbv_prio1.so; [ $v = p3 ] && lib=$ROOT/consensus_amd/libsbv_prio3.so
  for n in 1048576 262144; do
  ( SBV_LIB=$lib timeout 150 python bench.py --steps 8 --warmup 2 --tuples $n --no-cpu-baseline --primary-only >> "$OUT/bench_${v}_$n.log" 2>&1; echo "rc=$?" >> "$OUT/bench_${v}_$n.log" )
  done
done
done
python - "$OUT" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], "value %.1fM" % (d["value"] / 1e6), "ms %.3f" % d["ms_per_step"], "ok", d["bitmap_correct"], {k: round(v) for k, v in d["kernel_us"].items()})
PY
tail -3 "$OUT/devcheck.log"
