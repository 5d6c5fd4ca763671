#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03b
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for m in ac c aoc; do echo "== mode $m"; timeout 60 python tools/diag_cache.py $m 2>&1 | grep -v amdgpu.ids | tail -6; done | tee "$OUT/diag.log"
echo "== chunks=1 mode ac"; SBV_GROUP_CHUNKS=1 timeout 60 python tools/diag_cache.py ac 2>&1 | tail -3 | tee -a "$OUT/diag.log"
