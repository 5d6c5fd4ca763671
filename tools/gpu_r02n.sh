#!/bin/bash
# round 2, step n: all-distinct-keys kernel on the carry-free field: parity (full suite) + bench incl. without_key_grouping + size sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02n
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_gpu.log" ); tail -5 "$OUT/pytest_gpu.log"
( timeout 400 python bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$?" >> "$OUT/bench.err" ); tail -2 "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1fM ms %.3f ok %s" % (d["value"] / 1e6, d["ms_per_step"], d["bitmap_correct"]), d["kernel_us"])
for k in ("warm_key_cache", "without_key_grouping", "registered_key_path"):
    print(k, json.dumps(d.get(k))[:500])
PY
for lg in 12 14 16 17 18 19 21; do
  ( timeout 120 python bench.py --tuples $((1 << lg)) --steps 6 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/size_$lg.log" 2>&1 )
  python - "$OUT/size_$lg.log" "$lg" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("2^%s" % sys.argv[2], "value %.1fM" % (d["value"] / 1e6), "ms %.3f" % d["ms_per_step"], "ok", d["bitmap_correct"], "grouped", d["key_grouping"]["enabled"])
        break
else:
    print("2^%s" % sys.argv[2], "NO RESULT", open(sys.argv[1]).read()[-200:])
PY
done
