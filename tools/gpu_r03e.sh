#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03e
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
for k in "headline_batch or full_batch_2_20_bitmap" "chunked_ragged or full_batch_2_20_bitmap" "ed25519_edge or secp256k1_edge or full_batch_2_20_bitmap"; do
echo "== $k"
timeout 280 python -m pytest tests/test_gpu_edge_scale.py tests/test_gpu_parity.py -q -x -k "$k" 2>&1 | grep -v "^$" | tail -12 | tee -a "$OUT/p2.log"
done
