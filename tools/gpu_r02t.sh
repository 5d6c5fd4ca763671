#!/bin/bash
# round 2, step t: full GPU suite (chain emulation, batch signing included), Ed25519 kernel profile, devcheck
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02t
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -s > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" )
tail -5 "$OUT/pytest_gpu.log"; grep "\[sign\]" "$OUT/pytest_gpu.log"
cat > /tmp/ed_prof.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench, consensus_amd as sbv
sbv.init(0)
print(bench.leg_ed25519(sbv, torch, 1 << 20, 4, torch.cuda.Stream()))
PY
( cd "$ROOT" && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/edprof" -o ed -- python /tmp/ed_prof.py > "$OUT/edprof.log" 2>&1; echo "rc=$?" >> "$OUT/edprof.log" )
tail -3 "$OUT/edprof.log"
f=$(find "$OUT/edprof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
find "$OUT/edprof" -name "*.db" -delete 2>/dev/null; find "$OUT/edprof" -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
