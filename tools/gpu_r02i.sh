#!/bin/bash
# round 2, step i: pipelined host-pointer entry: concurrency test + end-to-end legs (1 and 2 submitting threads)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02i
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "concurrent or ragged or 2_20 or chunked" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log" ); tail -4 "$OUT/pytest.log"
timeout 300 python - > "$OUT/e2e.json" 2> "$OUT/e2e.err" <<'PY'
import ctypes, json, os, sys, threading, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
import consensus_amd as sbv, synth
sbv.init(0)
n = 1 << 20
tuples, valid = synth.gen_batch(0x5B7F2026, n)
res = {}
for kind in ("pinned", "pageable"):
    bufs = []
    for k in range(2):
        if kind == "pinned":
            ptr = sbv.host_alloc(n * 160); ctypes.memmove(ptr, tuples.ctypes.data, n * 160)
        else:
            ptr = tuples.ctypes.data
        bufs.append(ptr)
    got = [np.zeros(n // 8, dtype=np.uint8) for _ in range(2)]
    sbv.verify_batch_ptr(bufs[0], n, got[0].ctypes.data)
    for threads in (1, 2):
        reps = 6
        def work(k):
            for _ in range(reps):
                sbv.verify_batch_ptr(bufs[k], n, got[k].ctypes.data)
        th = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        res[f"{kind}_{threads}_threads"] = {"value": n * reps * threads / dt, "ms_per_call_per_thread": 1e3 * dt / reps,
                                            "ok": bool((got[0] == valid).all() and (threads == 1 or (got[1] == valid).all()))}
print(json.dumps(res))
PY
cat "$OUT/e2e.json"; tail -3 "$OUT/e2e.err"
