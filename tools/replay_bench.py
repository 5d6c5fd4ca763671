#!/usr/bin/env python3
"""Configs 3 and 4 of BASELINE.json and the commit-quorum latency (M2), through the C++ api.Verifier
mirror over the real backend.  Prints one JSON object per configuration.

  config 3: 4 nodes (f=1, Q=3), 10k-request proposal: VerifyProposal + prev-commit + commit-vote pattern
  config 4: 16 nodes (f=5, Q=11): 50k decisions x 11 consenter signatures as one batch (decision replay)
  M2      : N = 16, 15 concurrent VerifyConsenterSig calls -> 10 accepted (coalesced micro-batch)
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostlib  # noqa: E402

lib = hostlib.load()


class Timing(ctypes.Structure):          # include/sbv.h: sbv_timing
    _fields_ = [("h2d_us", ctypes.c_double), ("prep_us", ctypes.c_double), ("verify_us", ctypes.c_double),
                ("d2h_us", ctypes.c_double), ("total_us", ctypes.c_double), ("n", ctypes.c_uint64)]


# libsbv.so is already in the process (libsbv_host.so links it): same handle, same context
libsbv = ctypes.CDLL(os.path.join(ROOT, "consensus_amd", "libsbv.so"))
cb = hostlib.BACKEND_FN(lambda *a: -1)
threads = min(128, os.cpu_count() or 8)


def run(name, n_nodes, K, sequences, decisions, wait_us):
    v = lib.sbvh_verifier_new(0, 0, cb, None, 1 << 20, wait_us, 0)
    res = hostlib.ReplayResult()
    rc = lib.sbvh_replay(v, n_nodes, K, sequences, decisions, threads, ctypes.byref(res))
    last = None
    try:                        # device-side split of the LAST backend call of this configuration
        tm = Timing()
        if libsbv.sbv_last_timing(ctypes.byref(tm)) == 0:
            last = {"n": tm.n, "h2d_us": tm.h2d_us, "prep_us": tm.prep_us, "verify_us": tm.verify_us, "d2h_us": tm.d2h_us,
                    "total_us": tm.total_us}
    except Exception as e:      # noqa: BLE001 - diagnostics only
        last = {"error": repr(e)}
    lib.sbvh_verifier_free(v)
    q, f = ctypes.c_int(), ctypes.c_int()
    lib.sbvh_compute_quorum(n_nodes, ctypes.byref(q), ctypes.byref(f))
    out = {"config": name, "rc": rc, "n_nodes": n_nodes, "quorum": q.value, "K": K, "sequences": sequences,
           "coalesce_wait_us": wait_us,
           "verify_proposal_us": res.verify_proposal_us, "verify_proposal_sigs_per_s": K / (res.verify_proposal_us * 1e-6) if res.verify_proposal_us else None,
           "prev_commits_serial_us": res.prev_commits_us, "commit_quorum_latency_us": res.commit_quorum_us,
           "decisions": decisions, "batch_tuples": res.batch_tuples, "batch_total_us": res.batch_total_us,
           "batch_first_call_us": res.batch_first_us,
           "batch_sigs_per_s": res.batch_tuples / (res.batch_total_us * 1e-6) if res.batch_total_us else None,
           "amortised_us_per_decision": res.batch_total_us / decisions if decisions else None,
           "proposals_with_quorum": res.proposals_with_quorum, "backend_batches": res.backend_batches,
           "max_backend_batch": res.max_backend_batch, "setup_s": res.setup_s, "last_backend_call": last}
    print(json.dumps(out), flush=True)


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
run("config1-shape: 4 nodes, K=100", 4, 100, 5, 0, 50)
run("config3: 4 nodes, K=10000", 4, 10000 if not quick else 2000, 3, 0, 50)
run("M2: 16 nodes commit-quorum latency", 16, 10, 9, 0, 50)
run("M2: 16 nodes commit-quorum latency (coalesce 200us)", 16, 10, 9, 0, 200)
run("config4: 16 nodes, 50k decisions x 11 sigs", 16, 10, 1, 50000 if not quick else 5000, 50)
