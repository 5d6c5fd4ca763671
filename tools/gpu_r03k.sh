#!/bin/bash
# round 3, session k: the wide table builder (+ variable-time inversion) against the chain-of-additions form on one box:
# A/B of cold / warm steps at 2^17, 2^18, 2^20; parity subset; bench line with driver flags; per-kernel durations
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03k
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 300 python tools/ab_env.py 17,18,20 default SBV_GROUP_WIDE=0 SBV_GROUP_FSPLIT=1 SBV_GROUP_FSPLIT=2 SBV_GROUP_FSPLIT=4 SBV_GROUP_CHUNKS=2 SBV_GROUP_CHUNKS=4 > "$OUT/ab_wide.jsonl" 2> "$OUT/ab_wide.err"; echo "rc=$?" >> "$OUT/ab_wide.err" ); cat "$OUT/ab_wide.jsonl"; tail -2 "$OUT/ab_wide.err"
( timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_scale.py -m gpu -q -x > "$OUT/pytest_subset.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_subset.log" ); tail -4 "$OUT/pytest_subset.log"
( timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"; echo "rc=$?" >> "$OUT/bench_driver_flags.err" ); tail -2 "$OUT/bench_driver_flags.err"
python3 - "$OUT/bench_driver_flags.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench line", e); sys.exit(0)
print({k: d[k] for k in ("value", "ms_per_step", "bitmap_correct")}, d["kernel_us"], d["roofline"]["frac"])
for k in ("m2_commit_quorum_us", "verify_proposal_k10000_us", "warm_key_cache", "projected_strong_scaling"):
    print(k, json.dumps(d.get(k))[:500])
PY
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o r03k -- python "$ROOT/tools/sweep_sizes.py" 18 20 > "$OUT/sweep_prof.jsonl" 2> "$OUT/sweep_prof.err" ); cat "$OUT/sweep_prof.jsonl"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200 > "$OUT/kernel_stats_sweep.csv" && cat "$OUT/kernel_stats_sweep.csv"
find "$OUT/prof" -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null; find "$OUT/prof" -name "*.db" -delete 2>/dev/null
