#!/bin/bash
# round 3, session a: GPU suite on the quad-lane table chain / rows on the isomorphic curve / record-only stage A / warm-key
# grouping; primary bench + warm leg; kernel timeline; size sweep cold / warm / one-lane
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03a
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 500 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" ); tail -5 "$OUT/pytest_gpu.log"
( timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --primary-only --warm-leg > "$OUT/bench_primary.json" 2> "$OUT/bench_primary.err"; echo "rc=$?" >> "$OUT/bench_primary.err" ); tail -2 "$OUT/bench_primary.err"
python3 -c "
import json
d=json.load(open('$OUT/bench_primary.json'))
print({k: d[k] for k in ('value','ms_per_step','bitmap_correct')}, d['kernel_us'], d.get('warm_key_cache',{}).get('ms_per_step'))"
( timeout 300 python tools/sweep_sizes.py 10 12 14 16 17 18 19 20 > "$OUT/sweep_sizes.jsonl" 2> "$OUT/sweep_sizes.err"; echo "rc=$?" >> "$OUT/sweep_sizes.err" ); cat "$OUT/sweep_sizes.jsonl"; tail -2 "$OUT/sweep_sizes.err"
bash tools/gpu_timeline.sh r03a/tl k_p256_prep python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --primary-only
