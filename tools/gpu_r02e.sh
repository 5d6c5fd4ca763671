#!/bin/bash
# round 2, step e: table building on the carry-free field (bases / rows / fill): parity, A/B of chunks x rows-per-lane, timeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02e
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log" ); tail -4 "$OUT/pytest.log"
for cfg in "2 1" "3 1" "4 1" "4 2" "4 4" "3 2" "6 1" "8 1"; do
  set -- $cfg
  ( SBV_GROUP_CHUNKS=$1 SBV_GROUP_PARTS=$2 timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/bench_c$1_r$2.log" 2>&1; echo "rc=$?" >> "$OUT/bench_c$1_r$2.log" )
  python - "$OUT/bench_c$1_r$2.log" "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("chunks", sys.argv[2], "rows/lane", sys.argv[3], "value %.1fM" % (d["value"] / 1e6), "ms %.3f" % d["ms_per_step"], "ok", d["bitmap_correct"], {k: round(v) for k, v in d["kernel_us"].items()})
        break
else:
    print("chunks", sys.argv[2], "rows/lane", sys.argv[3], "NO RESULT", open(sys.argv[1]).read()[-300:])
PY
done
SBV_GROUP_CHUNKS=4 bash tools/gpu_timeline.sh r02e/timeline k_p256_prep python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --primary-only
