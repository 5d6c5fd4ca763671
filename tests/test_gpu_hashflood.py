"""GPU tier: hash-flooding defence of the grouped steps (consensus_amd/csrc/p256_group.h).  2^17 distinct keys
crafted to collide in the grouping table under the UNKEYED hash of rounds 1-4, each carrying a real signature of another key, are
interleaved with 2^17 honest tuples.  Every verdict must equal the expected one (the honest tuples' the generator's, the crafted
ones rejected) and the step must take no more than twice the time of the same batch with random keys with the library's random
per-context seed (the crafted keys no longer collide), three times with SBV_HASH_SEED=0 (they do: the probe bound of 64 keeps insert
work per tuple constant; unbounded probing made this batch quadratic).  All three instantiations of the table: P-256, secp256k1 (64-byte
keys), Ed25519 (32-byte keys)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scheme", ["p256", "k256", "ed25519"])
@pytest.mark.parametrize("seed_env", [None, "0"])
def test_crafted_colliding_keys_cost_no_more_than_random_ones(scheme, seed_env):
    env = dict(os.environ)
    env.pop("SBV_HASH_SEED", None)
    if seed_env is not None:
        env["SBV_HASH_SEED"] = seed_env
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hashflood_run.py"), scheme, "17"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    for leg in ("random", "colliding", "random_again"):
        assert r[leg]["verdicts_ok"], (leg, r)
    assert r["random"]["accepted"] == r["colliding"]["accepted"] > 0
    ref = min(r["random"]["ms"], r["random_again"]["ms"])
    # 0.5 ms of slack for launch jitter on a busy box.  Random seed (the product): twice the random batch at most.  SBV_HASH_SEED=0 (the
    # attacker knows the seed — a configuration only this test creates): every crafted tuple walks its 64 probes, a constant per tuple;
    # since the honest half got faster (hot keys: 1.5 ms for the random batch, PCIe included) that constant is a larger multiple of it
    # (3.7 ms measured, profiles/r05), so the bound here is three times
    assert r["colliding"]["ms"] <= (2.0 if seed_env is None else 3.0) * ref + 0.5, r
