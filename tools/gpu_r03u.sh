#!/bin/bash
# round 3, session u: the whole GPU suite, the bench line with the driver's flags, size sweep, and rocprofv3 kernel stats +
# timeline + HBM traffic of the committed library (rows + fill of each chunk on their own stream, pieced uploads in the sharded
# entry, copy-free VerifyProposal)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03u
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
for k in ("secp256k1", "m2_commit_quorum_us", "verify_proposal_k10000_us", "replay_550k", "front_end_msgs_per_s", "warm_key_cache", "ed25519", "registered_key_path", "without_key_grouping", "end_to_end", "sharded_entry"):
    print(k, json.dumps(d.get(k))[:520])
ps = d.get("projected_strong_scaling", {}).get("partitions", {})
for g, v in ps.items(): print(g, {k: round(x["projected_speedup"], 2) for k, x in v.items()})
PY
timeout 200 python tools/sweep_sizes.py 10 12 14 16 17 18 19 20 > "$OUT/sweep_sizes.jsonl" 2> "$OUT/sweep_sizes.err"; python3 - "$OUT/sweep_sizes.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["log2_tuples"], {k: (d[k]["ms"], d[k]["M_per_s"], d[k]["generic"]) for k in ("cold", "warm", "one_lane") if k in d})
PY
SBVH_TRACE=1 timeout 120 python - <<'PY' 2>&1 | grep "VerifyProposal" | tail -4 | tee "$OUT/verify_proposal_trace.txt"
import ctypes, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import hostlib
lib = hostlib.load()
cb = hostlib.BACKEND_FN(lambda *a: -1)
for on_device in (1, 0):
    v = lib.sbvh_verifier_new(0, 0, cb, None, 1 << 20, 50, 0)
    lib.sbvh_set_device_client_keys(v, on_device)
    res = hostlib.ReplayResult()
    lib.sbvh_replay(v, 4, 10000, 2, 0, min(64, os.cpu_count() or 8), ctypes.byref(res))
    lib.sbvh_verifier_free(v)
PY
cd /tmp
run() { local name=$1; shift; ( timeout 200 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o p -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --primary-only > "$OUT/$name.log" 2>&1; echo "rc=$?" >> "$OUT/$name.log" ); tail -1 "$OUT/$name.log"; }
run stats --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, collections, json, os, sys
out = sys.argv[1]
summary = {}
st = os.path.join(out, "stats", "p_kernel_stats.csv")
if os.path.exists(st):
    summary["kernel_stats"] = [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in csv.DictReader(open(st)) if "sbv::" in r["Name"]]
for d in ("pmc_fetch", "pmc_write"):
    f = os.path.join(out, d, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sbv::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("sbv::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        summary.setdefault("pmc", {}).setdefault(k, {})[c] = {"mean": sum(v) / len(v), "dispatches": len(v)}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
f = os.path.join(out, "stats", "p_kernel_trace.csv")
if os.path.exists(f):
    rows = [r for r in csv.DictReader(open(f)) if "sbv::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    packs = [i for i, r in enumerate(rows) if "k_pack_bitmap" in r["Kernel_Name"]]
    if len(packs) >= 2:
        seg = rows[packs[-2] + 1:packs[-1] + 1]
        t0 = int(seg[0]["Start_Timestamp"])
        with open(os.path.join(out, "timeline.txt"), "w") as fh:
            for r in seg:
                fh.write("%8.3f %8.3f  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r["Kernel_Name"].split("(")[0]))
        print(open(os.path.join(out, "timeline.txt")).read())
for k in summary.get("kernel_stats", [])[:8]: print(k)
PY
cp "$OUT/stats/p_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
