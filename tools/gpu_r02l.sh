#!/bin/bash
# round 2, step l: G comb width A/B (16 / 18 / 20 / 22 bits), keyed + coop kernels on the carry-free field, full gpu suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02l
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" ); tail -10 "$OUT/pytest_gpu.log"
for B in 16 20 18 22 20 16; do
  ( SBV_G_BITS=$B timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/bench_B$B.log" 2>&1; echo "rc=$?" >> "$OUT/bench_B$B.log" )
  python - "$OUT/bench_B$B.log" "$B" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("G bits", sys.argv[2], "value %.1fM" % (d["value"] / 1e6), "ms %.3f" % d["ms_per_step"], "ok", d["bitmap_correct"], {k: round(v) for k, v in d["kernel_us"].items()})
        break
else:
    print("G bits", sys.argv[2], "NO RESULT", open(sys.argv[1]).read()[-300:])
PY
done
( timeout 300 python tools/replay_bench.py quick > "$OUT/replay.jsonl" 2> "$OUT/replay.err"; echo "rc=$?" >> "$OUT/replay.err" ); python - "$OUT/replay.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["config"], "| verify_proposal_us %.0f" % d["verify_proposal_us"], "| prev_commits %.0f" % d["prev_commits_serial_us"], "| quorum_us %.0f" % d["commit_quorum_latency_us"], "| batch_us %.0f" % d["batch_total_us"])
PY
