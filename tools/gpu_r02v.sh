#!/bin/bash
# round 2, step v: Ed25519 on the carry-free field: GPU parity tests, then the configs[4] leg at 3 vs 2 waves/SIMD
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02v
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ed25519.py -m gpu -x -q > "$OUT/pytest_ed.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_ed.log" )
grep -E "passed|failed|rc=" "$OUT/pytest_ed.log" | tail -3
cat > /tmp/ed_leg.py <<'PY'
import sys, torch, json
sys.path.insert(0, ".")
import bench, consensus_amd as sbv
sbv.init(0)
r = bench.leg_ed25519(sbv, torch, 1 << 20, 6, torch.cuda.Stream())
print(json.dumps(r))
PY
for rep in 1 2; do
for v in w3 w2; do
  lib=$ROOT/consensus_amd/libsbv.so; [ $v = w2 ] && lib=$ROOT/consensus_amd/libsbv_edw2.so
  ( SBV_LIB=$lib timeout 200 python /tmp/ed_leg.py 2>&1 | grep '^{' | sed "s/^/$v /" >> "$OUT/ed_ab.log" )
done
done
cat "$OUT/ed_ab.log" | cut -c1-200
