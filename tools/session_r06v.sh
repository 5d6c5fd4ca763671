cd $GRAFT_REPO_ROOT; O=gpurun_out/r06v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ed25519.py -x -q -s > $O/pytest_ed.log 2>&1; echo "pytest_ed rc=$?" >> $O/pytest_ed.log
tail -25 $O/pytest_ed.log
timeout 900 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --legs ed25519 > $O/bench_ed.json 2> $O/bench_ed.err; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r06v/bench_ed.json") if x.startswith("{")]
d=json.loads(l[-1])
def find(o,k):
    if isinstance(o,dict):
        if k in o: return o[k]
        for v in o.values():
            r=find(v,k)
            if r is not None: return r
e=find(d,"ed25519")
print(json.dumps({k:e[k] for k in e if k in("value","ms_per_step","bitmap_correct","warm_key_cache","hot_keys")})[:1500])
PY
tail -5 $O/bench_ed.err
