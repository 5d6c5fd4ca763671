#!/bin/bash
# round 2, step x: is the Ed25519 comb phase bound by its table gathers?  Same batch shape with 64 keys (25 MB of tables) vs 1024 (402 MB)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02x
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
cat > /tmp/ed_keys.py <<'PY'
import sys, os, time, json, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import consensus_amd as sbv, hostlib
nkeys = int(sys.argv[2]); n = 1 << 20
h = hostlib.load()
tuples = np.zeros(n * 128, dtype=np.uint8); expect = np.zeros((n + 7) // 8, dtype=np.uint8)
h.sbvh_ed25519_gen_batch(20260921, n, nkeys, 0, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)      # all valid: no one-lane list
sbv.init(0)
d_t = torch.from_numpy(tuples).cuda(); d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 4
print(json.dumps({"nkeys": nkeys, "ms": dt * 1e3, "ok": bool((d_b.cpu().numpy() == expect).all())}))
PY
for nk in 1024 64; do
  ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/k$nk" -o p -- python /tmp/ed_keys.py "$ROOT" $nk > "$OUT/k$nk.log" 2>&1 )
  grep '^{' "$OUT/k$nk.log"
  grep -E "k_ed_qphase|k_ed_gphase|k_ed_keytab_window" "$OUT/k$nk/p_kernel_stats.csv" | awk -F'",' '{print $1}' | cut -c1-40 | paste - <(grep -E "k_ed_qphase|k_ed_gphase|k_ed_keytab_window" "$OUT/k$nk/p_kernel_stats.csv" | awk -F, '{print $(NF-4), $(NF-3)}') 
  rm -rf "$OUT/k$nk"
done
