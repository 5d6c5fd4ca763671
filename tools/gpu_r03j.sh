#!/bin/bash
# round 3, session j: the whole GPU suite and the bench line exactly as the driver runs it (steps 20, warmup 5), then kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03j
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" ); grep -n "passed\|failed\|FAILED\|rc=" "$OUT/pytest_gpu.log" | tail -8
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"; echo "rc=$?" >> "$OUT/bench_driver_flags.err" ); tail -2 "$OUT/bench_driver_flags.err"
python3 - "$OUT/bench_driver_flags.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench line", e); sys.exit(0)
print({k: d[k] for k in ("value", "ms_per_step", "bitmap_correct")}, d["kernel_us"], d["roofline"]["frac"], d.get("int_mul_issue_fraction", {}).get("value"))
for k in ("secp256k1", "m2_commit_quorum_us", "verify_proposal_k10000_us", "replay_550k", "front_end_msgs_per_s", "warm_key_cache", "ed25519", "registered_key_path", "without_key_grouping"):
    print(k, json.dumps(d.get(k))[:420])
PY
python3 -c "import secrets" ; python - <<'PY'
import subprocess, os
PY
timeout 200 python /dev/stdin <<'PY' 2>/dev/null | tee "$OUT/k256_rate.jsonl"
import json, os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import consensus_amd as sbv
import hostlib
h = hostlib.load()
n = 1 << 20
tuples = np.zeros(n * 160, dtype=np.uint8); expect = np.zeros(n // 8, dtype=np.uint8)
h.sbvh_k256_gen_batch(0x5B7F2026, n, 1024, 8, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)
sbv.init(0)
d_t = torch.from_numpy(tuples).cuda(); d_b = torch.zeros(n // 8, dtype=torch.uint8, device="cuda"); st = torch.cuda.current_stream()
t0 = time.perf_counter()
sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
first = time.perf_counter() - t0
steps = 8
t0 = time.perf_counter()
for _ in range(steps): sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), st.cuda_stream)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(json.dumps({"path": "grouped, 20-bit comb, prefetch", "ms": round(1e3 * dt, 3), "M_per_s": round(n / dt / 1e6, 1), "ok": bool((d_b.cpu().numpy() == expect).all()), "first_call_s": round(first, 2)}), flush=True)
PY
