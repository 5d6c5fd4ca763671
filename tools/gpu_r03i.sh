#!/bin/bash
# round 3, session i: secp256k1 grouped step on the GPU (tests, rate, timeline); the small path re-check
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03i
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_gpu_k256.py tests/test_gpu_edge_scale.py -m gpu -q -k "k256 or secp256k1 or grouped_step or golden or ragged or 2_20" > "$OUT/pytest_k256.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_k256.log" ); grep -n "passed\|failed\|FAILED\|rc=\|secp256k1:" "$OUT/pytest_k256.log" | tail -8
cat > /tmp/k256_rate.py <<'PY'
import json, os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import numpy as np, torch
import consensus_amd as sbv
import hostlib
h = hostlib.load()
n = 1 << 20
tuples = np.zeros(n * 160, dtype=np.uint8); expect = np.zeros(n // 8, dtype=np.uint8)
h.sbvh_k256_gen_batch(0x5B7F2026, n, 1024, 8, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)
sbv.init(0)
d_t = torch.from_numpy(tuples).cuda(); d_b = torch.zeros(n // 8, dtype=torch.uint8, device="cuda"); st = torch.cuda.current_stream()
for label, grouping in (("grouped", True), ("one_lane", False)):
    sbv.set_grouping(grouping)
    sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
    steps = 6 if grouping else 2
    t0 = time.perf_counter()
    for _ in range(steps): sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"path": label, "ms": round(1e3 * dt, 3), "M_per_s": round(n / dt / 1e6, 1), "ok": bool((d_b.cpu().numpy() == expect).all()), "groups": sbv.last_group_stats()}), flush=True)
sbv.set_grouping(True)
PY
timeout 200 python /tmp/k256_rate.py 2>/dev/null | tee "$OUT/k256_rate.jsonl"
cat > /tmp/k256_tl.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import numpy as np, torch
import consensus_amd as sbv
import hostlib
h = hostlib.load()
n = 1 << 20
tuples = np.zeros(n * 160, dtype=np.uint8); expect = np.zeros(n // 8, dtype=np.uint8)
h.sbvh_k256_gen_batch(0x5B7F2026, n, 1024, 8, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)
sbv.init(0)
d_t = torch.from_numpy(tuples).cuda(); d_b = torch.zeros(n // 8, dtype=torch.uint8, device="cuda"); st = torch.cuda.current_stream()
for _ in range(3): sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
PY
bash tools/gpu_timeline.sh r03i/tl k_k256_prep_rec python /tmp/k256_tl.py
