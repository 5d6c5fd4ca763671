#!/bin/bash
# One gpurun call: smoke -> microbench -> parity tests -> bench -> rocprofv3 (stats, then PMC passes).
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [quick]
set -u
TAG=${1:-r01}
MODE=${2:-full}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
echo "== host: $(nproc) cores; $(lscpu | grep 'Model name' | sed 's/.*: *//')" | tee "$OUT/host.txt"
rocm-smi --showproductname 2>/dev/null | head -12 >> "$OUT/host.txt"

echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee "$OUT/smoke.log" | tail -3
echo "== microbench"; timeout 300 tools/microbench > "$OUT/microbench.jsonl" 2> "$OUT/microbench.err"; tail -3 "$OUT/microbench.jsonl"
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tee "$OUT/pytest_gpu.log" | tail -15
echo "== bench"; timeout 900 python bench.py --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"; cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
if [ "$MODE" = "full" ]; then
  cd /tmp
  echo "== rocprofv3 kernel-trace/stats"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o stats -- \
      python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/prof_stats.log" 2>&1
  tail -2 "$OUT/prof_stats.log"
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== rocprofv3 --pmc $C"
    timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/prof_pmc_$C" -o pmc -- \
        python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/prof_pmc_$C.log" 2>&1
    tail -1 "$OUT/prof_pmc_$C.log"
  done
  echo "== rocprofv3 --pmc SQ counters"
  timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv \
      -d "$OUT/prof_pmc_SQ" -o pmc -- python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/prof_pmc_SQ.log" 2>&1
  tail -1 "$OUT/prof_pmc_SQ.log"
  cd "$ROOT"
  find "$OUT" -name "*.csv" -size +20M -delete     # keep gpurun_out under the 64 MiB merge limit
fi
du -sh "$OUT"
echo "== done"
