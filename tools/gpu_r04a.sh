#!/bin/bash
# round 4, first session (prepared at the end of round 3, when the GPU budget was spent): measure what round 3 left switched off.
#   1. opt-in correctness: k_group_coop (SBV_TEST_COOP=1) and the key-affine part without the host round trip (SBV_PART_NOSYNC=1)
#   2. warm small batches with and without the coop launch (fresh processes: the knob is read when the context is created)
#   3. the split of the 33 windows over the Q launches: a short first chunk (SBV_GROUP_CHUNK0: its tables are ready when the
#      3-wave G phase ends — timeline_r03u.txt shows the first Q launch waiting 0.3 ms for an even one), 1 and 3 chunks
#      and the symmetric fill (SBV_GROUP_WIDE=4: babies 1..8 only, both sides of every giant from one inverse)
#   4. the projection leg with SBV_PART_NOSYNC=1
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04a
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
( SBV_TEST_COOP=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k coop_form > "$OUT/pytest_coop.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_coop.log" ); tail -3 "$OUT/pytest_coop.log"
( SBV_PART_NOSYNC=1 timeout 400 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "key_affine or by_key" > "$OUT/pytest_part_nosync.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_part_nosync.log" ); tail -3 "$OUT/pytest_part_nosync.log"
( SBV_TEST_STRESS=1 timeout 400 python -m pytest tests/test_gpu_configs.py -m gpu -q -k two_threads > "$OUT/pytest_stress.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_stress.log" ); tail -3 "$OUT/pytest_stress.log"
for rep in 1 2; do for v in 0 32768; do
  echo "# SBV_GROUP_COOP_MAX=$v rep $rep" >> "$OUT/coop.jsonl"
  SBV_GROUP_COOP_MAX=$v timeout 100 python tools/sweep_sizes.py 10 12 13 14 15 >> "$OUT/coop.jsonl" 2>> "$OUT/coop.err"
done; done
python3 - "$OUT/coop.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("#"): print(l.strip()); continue
    d = json.loads(l); print(d["log2_tuples"], "cold", d["cold"]["ms"], "warm", d["warm"]["ms"], d["cold"]["ok"], d["warm"]["ok"])
PY
( timeout 300 python tools/ab_env.py 18,19,20 default SBV_GROUP_CHUNK0=6 SBV_GROUP_CHUNK0=8 SBV_GROUP_CHUNK0=10 SBV_GROUP_CHUNK0=12 SBV_GROUP_WIDE=4 SBV_GROUP_WIDE=4,SBV_GROUP_CHUNK0=10 SBV_GROUP_CHUNKS=1 SBV_GROUP_CHUNKS=3 SBV_GPHASE_SPLIT_MIN=0 SBV_GPHASE_SPLIT_MIN=0,SBV_GROUP_CHUNK0=10 > "$OUT/ab_chunks.jsonl" 2> "$OUT/ab_chunks.err" ); cat "$OUT/ab_chunks.jsonl"
for ns in 0 1; do
( SBV_PART_NOSYNC=$ns timeout 300 python - > "$OUT/projection_nosync$ns.json" 2> "$OUT/projection_nosync$ns.err" <<'PY'
import json, os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
import torch
import bench, synth
import consensus_amd as sbv
n = 1 << 20
tuples, valid = synth.gen_batch(bench.SEED, n)
sbv.init(0); sbv.key_cache(False)
d_tuples = torch.from_numpy(tuples).cuda()
stream = torch.cuda.current_stream()
d_b = torch.zeros(n // 8, dtype=torch.uint8, device="cuda")
for _ in range(3): sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize(); base = 1e3 * (time.perf_counter() - t0) / 8
print(json.dumps(bench.leg_projected_strong_scaling(sbv, torch, d_tuples, valid, n, stream, base)))
PY
); python3 -c "
import json
d=json.load(open('$OUT/projection_nosync$ns.json'))
print('nosync=$ns', {g: {k: round(x['projected_speedup'], 2) for k, x in v.items()} for g, v in d['partitions'].items()})"
done
# 5. Ed25519 grouped step with the second table stream
for rep in 1 2; do for v in 0 1; do
  SBV_ED_TSTREAMS=$v timeout 200 python - >> "$OUT/ed_tstreams.jsonl" 2>> "$OUT/ed_tstreams.err" <<'PY'
import json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
import torch
import bench
import consensus_amd as sbv
sbv.init(0)
r = bench.leg_ed25519(sbv, torch, 1 << 20, 8, torch.cuda.current_stream())
print(json.dumps({"SBV_ED_TSTREAMS": os.environ.get("SBV_ED_TSTREAMS"), "ms": r.get("ms_per_step"), "ok": r.get("bitmap_correct")}))
PY
done; done; cat "$OUT/ed_tstreams.jsonl"
# 6. secp256k1 grouped step: tuples per inversion in stage A
for v in 1 4 8 1 4 8; do
  SBV_K256_PREP_T=$v timeout 200 python - >> "$OUT/k256_prep_t.jsonl" 2>> "$OUT/k256_prep_t.err" <<'PY'
import json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
import torch
import bench
import consensus_amd as sbv
sbv.init(0)
r = bench.leg_secp256k1(sbv, torch, 1 << 20, 8, torch.cuda.current_stream())
print(json.dumps({"SBV_K256_PREP_T": os.environ.get("SBV_K256_PREP_T"), "ms": r.get("ms_per_step"), "ok": r.get("bitmap_correct")}))
PY
done; cat "$OUT/k256_prep_t.jsonl"
# 7. the one comb width of G that fits the 256 MiB Infinity Cache and was never run: 19 bits (14 additions, 218 MB) against 20
for rep in 1 2; do for v in 20 19; do
  echo "# SBV_G_BITS=$v rep $rep" >> "$OUT/gbits.jsonl"
  SBV_G_BITS=$v timeout 200 python tools/ab_env.py 18,20 default >> "$OUT/gbits.jsonl" 2>> "$OUT/gbits.err"
done; done; cat "$OUT/gbits.jsonl"
