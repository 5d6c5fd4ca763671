cd ${GRAFT_REPO_ROOT:-/root/repo}; OUT=gpurun_out/r05r; mkdir -p $OUT
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --legs replay_550k > /dev/null 2>&1
for rep in 1 2; do
SBV_HOT_KEYS=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --legs replay_550k,end_to_end > $OUT/bench_hot_off_$rep.json 2> $OUT/bench_hot_off_$rep.err
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --legs replay_550k,end_to_end > $OUT/bench_hot_on_$rep.json 2> $OUT/bench_hot_on_$rep.err
done
timeout 300 python tools/key_sweep.py 20 1024,4096,16384 > $OUT/key_sweep.json 2> $OUT/key_sweep.err; cat $OUT/key_sweep.err | grep -v amdgpu
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05r/bench_hot_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line", e); continue
    r=d.get("replay_550k",{}); e=d.get("end_to_end",{})
    print(f.split("/")[-1], "value %.1f"%(d["value"]/1e6), "replay %.2f ms"%r.get("ms_per_call",0), r.get("last_call_us"), "| e2e cache on:", {k:(round(v["value"]/1e6,1) if isinstance(v,dict) and "value" in v else v) for k,v in e.get("pinned_key_cache_on",{}).items() if k!="last_call_us"}, "| cache off:", {k:round(v["value"]/1e6,1) for k,v in e.get("pinned",{}).items() if k!="last_call_us"})
PY
