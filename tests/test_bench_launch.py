"""CPU tier: the launch path of `bench.py --gpus N` (VERDICT r5 #1a).  The driver's scaling run starts the ranks itself, but the form
`python bench.py --gpus N` — what a user types, and what the driver uses for N = 1 — used to exit before touching a GPU when N > 1.
Now it re-executes itself through torch.distributed.run on 127.0.0.1; `--dry-run` rehearses exactly that path over gloo without a GPU
(nothing is verified in a dry run: libsbv.so has no CPU path, and the oracle is not the product)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _last_json(out: str):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_gpus_2_launches_its_own_ranks_dry_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--tuples", "4096", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["self_launched"] is True
    assert line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert len(line["per_rank_ms_per_step"]) == 2 and line["ms_per_step"] == max(line["per_rank_ms_per_step"])
    assert line["bitmap_correct"] is True and line["value"] == 0.0                 # a dry run measures nothing
    assert line["config"]["global_batch"] == 2 * 4096
    assert sum(1 for ln in p.stdout.splitlines() if ln.startswith("{")) == 1       # ONE line, from rank 0


def test_bench_under_an_external_launcher_dry_run():
    """The driver's form: python -m torch.distributed.run ... bench.py --gpus 2 (bench.py must NOT launch again)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--tuples", "2048", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 2 and line["self_launched"] is False and len(line["per_rank_ms_per_step"]) == 2


def test_rank_count_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "disagree" in (p.stderr + p.stdout)


def test_rank_shards_are_rotations_with_matching_verdicts(oracle):
    """shard_of_rank: rank r's tuples are the base batch rotated by r * 4099 positions and its expected bitmap is rotated with them —
    checked against the oracle's verdicts on the rotated tuples (a mis-rotated bitmap would make every rank > 0 report bitmap_correct
    false on the 8-GPU run nobody can rehearse here)."""
    import bench
    import synth
    n = 6000
    base_t, base_v = bench.shard_of_rank(synth, np, None, 0, n)
    for rank in (1, 3):
        t, v = bench.shard_of_rank(synth, np, None, rank, n)
        k = (rank * 4099) % n
        assert bytes(t[:160]) == bytes(base_t[160 * k:160 * (k + 1)])
        got = ctypes.create_string_buffer((n + 7) // 8)
        oracle.sbvo_p256_verify_batch(t.ctypes.data, n, got, os.cpu_count() or 1)
        assert got.raw == v.tobytes()
    assert 0 < int(np.unpackbits(base_v).sum()) < n
