#!/bin/bash
# round 3, session d: GPU suite with the edge-scale tests and the one-launch latency form; full default bench (new legs);
# A/B of the profiling events' cost; kernel timeline of a cold 2^18 batch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03d
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 700 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" ); tail -6 "$OUT/pytest_gpu.log"
( timeout 420 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?" >> "$OUT/bench_default.err" ); tail -2 "$OUT/bench_default.err"
python3 - "$OUT/bench_default.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench line", e); sys.exit(0)
print({k: d[k] for k in ("value", "ms_per_step", "bitmap_correct")}, d["kernel_us"], d["roofline"]["frac"], d.get("int_mul_issue_fraction", {}).get("value"))
for k in ("m2_commit_quorum_us", "verify_proposal_k10000_us", "replay_550k", "front_end_msgs_per_s", "warm_key_cache", "sharded_entry"):
    print(k, json.dumps(d.get(k))[:700])
PY
for prof in 2 0; do
  ( SBV_BENCH_PROFILE=$prof timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --primary-only > "$OUT/bench_prof$prof.json" 2> "$OUT/bench_prof$prof.err" )
  python3 -c "
import json
d=json.load(open('$OUT/bench_prof$prof.json')); print('profile level $prof:', round(d['ms_per_step'],4), 'ms/step', d['kernel_us'])"
done
bash tools/gpu_timeline.sh r03d/tl18 k_p256_prep python "$ROOT/tools/sweep_sizes.py" 18
