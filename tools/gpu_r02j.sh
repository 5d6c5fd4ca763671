#!/bin/bash
# round 2, step j: stage A on the carry-free scalar field: parity + bench (T sweep) + timeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02j
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log" ); tail -4 "$OUT/pytest.log"
for T in 8 4 16 2 8; do
  ( SBV_PREP_T=$T timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/bench_T$T.log" 2>&1; echo "rc=$?" >> "$OUT/bench_T$T.log" )
  python - "$OUT/bench_T$T.log" "$T" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("T", sys.argv[2], "value %.1fM" % (d["value"] / 1e6), "ms %.3f" % d["ms_per_step"], "ok", d["bitmap_correct"], {k: round(v) for k, v in d["kernel_us"].items()})
        break
else:
    print("T", sys.argv[2], "NO RESULT", open(sys.argv[1]).read()[-300:])
PY
done
bash tools/gpu_timeline.sh r02j/timeline k_p256_prep python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --primary-only
