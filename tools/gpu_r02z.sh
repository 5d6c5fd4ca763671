#!/bin/bash
# round 2, step z: where the 0.3 ms of a quorum-sized batch (15 signatures, registered keys) go
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02z
mkdir -p "$OUT"; cd "$ROOT"; export TMPDIR=/tmp
cat > /tmp/m2.py <<'PY'
import sys, os, time, ctypes, hashlib, json, statistics
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import consensus_amd as sbv, hostlib
N = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551
sbv.init(0)
h = hostlib.load()
keys = b"".join(hashlib.sha256(b"k%d" % i).digest() for i in range(16))
pubs = []
for i in range(16):
    q = ctypes.create_string_buffer(64); h.sbvh_pubkey(keys[32 * i:32 * i + 32], q); pubs.append(q.raw)
slots = sbv.register_keys(pubs)
for n in (15, 64, 1024):
    dig = b"".join(hashlib.sha256(b"m%d" % i).digest() for i in range(n))
    sigs, ok = sbv.sign_batch(keys, dig, [i % 16 for i in range(n)])
    rsh = b"".join(sigs[64 * i:64 * i + 64] + dig[32 * i:32 * i + 32] for i in range(n))
    sl = [slots[i % 16] for i in range(n)]
    rows = []
    for it in range(60):
        t0 = time.perf_counter()
        bm = sbv.verify_batch_keyed(rsh, sl, n)
        wall = 1e6 * (time.perf_counter() - t0)
        t = sbv.last_timing()
        rows.append((wall, t.h2d_us, t.prep_us, t.verify_us, t.d2h_us, t.total_us))
    assert bm == bytes([0xFF] * (n // 8)) + (bytes([(1 << (n % 8)) - 1]) if n % 8 else b"")
    med = [round(statistics.median(r[k] for r in rows[10:]), 1) for k in range(6)]
    print(json.dumps({"n": n, "wall_us": med[0], "h2d_us": med[1], "prep_us": med[2], "verify_us": med[3], "d2h_us": med[4], "api_total_us": med[5]}))
PY
python /tmp/m2.py "$ROOT" 2>&1 | grep '^{' | tee "$OUT/m2_breakdown.jsonl"
