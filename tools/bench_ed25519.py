#!/usr/bin/env python3
"""Ed25519 variant (BASELINE.json configs[4]): 2^20 signatures, 1024 keys, one MI355X.  One JSON line.
Tuples (R|S|A|k, 128 B) are resident in HBM when the timed region starts; k = SHA-512(R||A||M) mod L is
computed on the host by sbv_ed25519_make_tuples (device SHA-512 is a 'next' item)."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import consensus_amd as sbv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps, warmup = 5, 1
sbv.init(0)
cache = f"/tmp/sbv_ed_batch_{n}.npz"
if os.path.exists(cache):
    z = np.load(cache); tuples, expect = z["tuples"], z["expect"]
else:
    # synthetic signatures come from the host library's RFC 8032 signer (consensus_amd/host/ed25519_host.cc: the
    # api.Signer half of the product; cross-checked byte for byte against the oracle's generator in tests/test_datagen.py)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostlib  # noqa: E402
    h = hostlib.load()
    tuples = np.zeros(n * 128, dtype=np.uint8); expect = np.zeros((n + 7) // 8, dtype=np.uint8)
    h.sbvh_ed25519_gen_batch(0x5B7F2026, n, 1024, 8, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)
    np.savez(cache, tuples=tuples, expect=expect)
if os.environ.get("SBV_ED_COLD"):        # every step builds every comb (the headline convention); default: the scheme's key-table cache stays on
    sbv.key_cache(False, 0, sbv.SCHEME_ED25519)
d_t = torch.from_numpy(tuples).cuda()
d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream()
for _ in range(warmup):
    sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize()
sbv.profile_enable(True)
t0 = time.perf_counter()
for _ in range(steps):
    sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_, verify_us, launches = sbv.profile_read()
sbv.profile_enable(False)
ok = bool((d_b.cpu().numpy() == expect).all())
groups, n_grouped, n_ungrouped, n_key_rejected = sbv.last_group_stats()
was_grouped = (n_grouped + n_ungrouped + n_key_rejected) == n
# secondary: the same batch with in-step key grouping off (one lane per signature, 256 doublings each)
if os.environ.get("SBV_BENCH_PRIMARY_ONLY"):
    steps_plain = 0
else:
    steps_plain = steps
sbv.set_grouping(False)
t1 = time.perf_counter()
for _ in range(steps_plain):
    sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize()
dt_plain = max(time.perf_counter() - t1, 1e-9)
ok_plain = bool((d_b.cpu().numpy() == expect).all())
sbv.set_grouping(True)
kern = verify_us / max(1, launches) * 1e-6
print(json.dumps({"metric": "Ed25519 verifies/sec at batch=1M (configs[4])", "value": n * steps / dt, "unit": "verifies/s", "n_gpus": 1,
                  "steps": steps, "ms_per_step": 1e3 * dt / steps, "bitmap_correct": ok, "dtype": "u32", "data": "synthetic",
                  "kernel_us": {"all_kernels_of_the_step": verify_us / max(1, launches)},
                  "key_grouping": {"enabled": was_grouped, "groups": groups, "tuples_key_tables": n_grouped,
                                   "tuples_one_lane_kernel": n_ungrouped, "tuples_rejected_for_their_key": n_key_rejected},
                  "without_key_grouping": {"value": n * steps_plain / dt_plain, "unit": "verifies/s", "bitmap_correct": ok_plain},
                  "roofline": {"bound": "hbm", "achieved": 128.125 * n / kern / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": 128.125 * n / kern / 1e9 / 8000.0, "traffic": None,
                               "kernel": "all kernels of the step (k_ed_gphase_generic + k_ed_qphase x chunks when grouped)"}}))
