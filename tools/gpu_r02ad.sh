#!/bin/bash
# round 2, step ad: secp256k1 variant on the GPU: parity tests through the C-ABI + kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ad
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_gpu_k256.py -m gpu -x -q -s > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log" ); tail -6 "$OUT/pytest.log"
