#!/bin/bash
# ONE gpurun call = one session: `bash tools/gpu_session.sh <tag> <step> [<step> ...]`, every step bounded by its own timeout,
# every artefact under gpurun_out/<tag>/ (copy what is to be judged into profiles/rNN/).  Steps (arguments after ':' are
# comma-separated; spaces inside a step need quoting by the caller):
#   pytest[:<-k expression>]      python -m pytest tests -m gpu -q  (whole tier, or the selection)
#   bench[:<name>[:<extra flags>]] bench.py with the driver's flags (--gpus 1 --steps 20 --warmup 5) -> bench_<name>.json
#   stats                         rocprofv3 --kernel-trace --stats of bench.py (primary leg) -> kernel_stats.csv, timeline.txt
#   pmc                           separate --pmc passes: FETCH_SIZE, WRITE_SIZE, SQ counters -> pmc_summary.json
#   sweep:<log2,log2,...>         tools/sweep_sizes.py (cold / warm / one-lane per batch size)
#   ab:<log2,..>:<variant>:<variant>...   tools/ab_env.py in ONE process (variant = NAME=VAL+NAME=VAL or "default")
#   abfresh:<log2,..>:<variant>:...       the same, a fresh process per variant, two passes (per-process knobs)
#   py:<script>[:args...]         python tools/<script> args   (anything else a session needs)
#   trace:<script>[:args...]      the same under rocprofv3 --kernel-trace --stats -> kernel_stats_<script>.csv
#   isa                           tools/isa_stats.sh (static ISA statistics; needs no GPU, here for the record of the build)
set -u
TAG=${1:?tag}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
echo "== host: $(nproc) cores; $(lscpu | grep 'Model name' | sed 's/.*: *//')" | tee "$OUT/host.txt"
run() { local name=$1 limit=$2; shift 2; ( timeout "$limit" "$@" > "$OUT/$name.log" 2>&1; echo "rc=$?" >> "$OUT/$name.log" ); }
for step in "$@"; do
  IFS=':' read -r kind a1 a2 rest <<< "$step"
  echo "== $step"
  case "$kind" in
    pytest)
      if [ -n "${a1:-}" ]; then run "pytest_$(echo "$a1" | tr -c 'A-Za-z0-9_\n' '_')" 1500 python -m pytest tests -m gpu -q -k "$a1"; tail -4 "$OUT"/pytest_*.log | tail -6
      else run pytest_gpu 2400 python -m pytest tests -m gpu -q -rs --durations=25; tail -6 "$OUT/pytest_gpu.log"; fi ;;
    bench)
      name=${a1:-driver_flags}
      ( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ${a2:-} > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "rc=$?" >> "$OUT/bench_$name.err" )
      python3 - "$OUT/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value %.1f M/s  %.3f ms/step  roofline %s  m2 %s" % (d["value"] / 1e6, d["ms_per_step"], d.get("roofline"), d.get("m2_commit_quorum_us")))
except Exception as e:
    print("no bench line:", e)
PY
      tail -2 "$OUT/bench_$name.err" ;;
    stats)
      timeout 200 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --primary-only > /dev/null 2>&1      # warms the /tmp batch cache
      ( cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st" -o p -- python "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --primary-only > "$OUT/stats.log" 2>&1; echo "rc=$?" >> "$OUT/stats.log" )
      cp "$OUT/st/p_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
      python3 - "$OUT/st/p_kernel_trace.csv" "$OUT/timeline.txt" <<'PY'
import csv, sys
try:
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sbv::" in r["Kernel_Name"]]
except Exception as e:
    print("no trace", e); sys.exit(0)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_p256_prep" in r["Kernel_Name"]]
if idx:
    # the last step: its first kernel is the group insert / memset a little before stage A
    last = idx[-1]
    t0 = int(rows[last]["Start_Timestamp"])
    with open(sys.argv[2], "w") as f:
        for r in rows[max(0, last - 2):]:
            line = "%-28s start %8.3f ms  end %8.3f ms" % (r["Kernel_Name"].split("(")[0].replace("sbv::", ""), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6)
            print(line); f.write(line + "\n")
PY
      rm -rf "$OUT/st"; cut -c1-150 "$OUT/kernel_stats.csv" | head -14 ;;
    pmc)
      timeout 200 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --primary-only > /dev/null 2>&1
      for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
        set -- $pass; pname=$1; shift
        # the primary leg AND the Ed25519 / secp256k1 legs (round 5: their roofline.traffic used to be null: no PMC pass covered them)
        ( cd /tmp; SBV_BENCH_ED_HOT=0 timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$pname" -o p -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --legs ed25519,secp256k1 > "$OUT/pmc_$pname.log" 2>&1; echo "rc=$?" >> "$OUT/pmc_$pname.log" )
      done
      python3 - "$OUT" <<'PY'
import csv, collections, json, os, sys
out = sys.argv[1]
summary = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = os.path.join(out, d, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sbv::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("sbv::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        summary.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "dispatches": len(v)}
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for k, v in summary.items():
    if "FETCH_SIZE" in v: print(k, {c: round(x["mean"], 1) for c, x in v.items() if c in ("FETCH_SIZE", "WRITE_SIZE")})
PY
      rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" ;;
    sweep)
      ( timeout 600 python tools/sweep_sizes.py $(echo "${a1:-12,14,16,17,18,19,20}" | tr ',' ' ') > "$OUT/sweep_sizes.jsonl" 2> "$OUT/sweep_sizes.err" ); cat "$OUT/sweep_sizes.jsonl" ;;
    ab)
      ( timeout 900 python tools/ab_env.py "$a1" $(echo "${a2:-default}:${rest:-}" | tr ':' ' ' | tr '+' ',') > "$OUT/ab_$a1.jsonl" 2> "$OUT/ab_$a1.err" ); cat "$OUT/ab_$a1.jsonl" ;;
    abfresh)
      for rep in 1 2; do for v in $(echo "${a2:-default}:${rest:-}" | tr ':' ' '); do
        echo "# $v rep $rep" >> "$OUT/abfresh_$a1.jsonl"
        if [ "$v" = default ]; then timeout 300 python tools/ab_env.py "$a1" default >> "$OUT/abfresh_$a1.jsonl" 2>> "$OUT/abfresh_$a1.err"
        else env $(echo "$v" | tr '+' ' ') timeout 300 python tools/ab_env.py "$a1" default >> "$OUT/abfresh_$a1.jsonl" 2>> "$OUT/abfresh_$a1.err"; fi
      done; done; cat "$OUT/abfresh_$a1.jsonl" ;;
    py)
      name=$(basename "${a1%.py}")
      ( timeout 900 python "tools/$a1" $(echo "${a2:-}:${rest:-}" | tr ':' ' ') > "$OUT/$name.out" 2> "$OUT/$name.err"; echo "rc=$?" >> "$OUT/$name.err" ); tail -30 "$OUT/$name.out"; tail -3 "$OUT/$name.err" ;;
    trace)          # trace:<script>[:args...]  rocprofv3 --kernel-trace --stats of python tools/<script>: per-kernel durations
      name=$(basename "${a1%.py}")
      ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tr_$name" -o p -- python "$ROOT/tools/$a1" $(echo "${a2:-}:${rest:-}" | tr ':' ' ') > "$OUT/trace_$name.log" 2>&1; echo "rc=$?" >> "$OUT/trace_$name.log" )
      cp "$OUT/tr_$name/p_kernel_stats.csv" "$OUT/kernel_stats_$name.csv" 2>/dev/null
      python3 - "$OUT/tr_$name/p_kernel_trace.csv" "$OUT/timeline_$name.txt" <<'PY'
import csv, sys
try:
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sbv::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = max(i for i, r in enumerate(rows) if "group_insert" in r["Kernel_Name"] or "k_p256_prep" in r["Kernel_Name"] or "k_k256_prep" in r["Kernel_Name"])
    firsts = [i for i, r in enumerate(rows) if "group_insert" in r["Kernel_Name"]]
    start = firsts[-1] if firsts else last
    t0 = int(rows[start]["Start_Timestamp"])
    with open(sys.argv[2], "w") as f:
        for r in rows[start:]:
            f.write("%-28s start %8.3f ms  end %8.3f ms\n" % (r["Kernel_Name"].split("(")[0].replace("sbv::", "").replace("void ", ""), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
except Exception as e:
    print("no timeline", e)
PY
      rm -rf "$OUT/tr_$name"
      python3 - "$OUT/kernel_stats_$name.csv" <<'PY'
import csv, sys
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "sbv::" in r["Name"]:
            print("%-34s calls %6s  avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"].split("(")[0].replace("sbv::", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
except Exception as e:
    print("no stats", e)
PY
      ;;
    isa) bash tools/isa_stats.sh > "$OUT/isa_stats.txt" 2>&1; grep "^==" "$OUT/isa_stats.txt" | cut -c1-200 ;;
    *) echo "unknown step $step" ;;
  esac
done
find "$OUT" -name "*.csv" -size +20M -delete
du -sh "$OUT"; echo "== done"
