#!/usr/bin/env python3
"""M2 through the C++ api.Verifier mirror with the coalescer's per-batch trace on (SBVH_TRACE=1): which batches a burst of 15
commit votes becomes, how long the leader collects, how long the backend call takes.  Prints the replay's summary as JSON and
the trace lines of the last sequences."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostlib
    lib = hostlib.load()
    cb = hostlib.BACKEND_FN(lambda *a: -1)
    v = lib.sbvh_verifier_new(0, 0, cb, None, 1 << 20, 50, 0)
    res = hostlib.ReplayResult()
    rc = lib.sbvh_replay(v, 16, 10, 15, 0, min(64, os.cpu_count() or 8), ctypes.byref(res))
    lib.sbvh_verifier_free(v)
    print(json.dumps({"rc": rc, "status": res.status, "commit_quorum_us": res.commit_quorum_us, "prev_commits_serial_us": res.prev_commits_us,
                      "verify_proposal_us": res.verify_proposal_us, "backend_batches": res.backend_batches, "max_backend_batch": res.max_backend_batch}))
    sys.exit(0)
for rep in (1, 2):
    for trace in ("0", "1"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SBVH_TRACE=trace), capture_output=True, text=True, timeout=300)
        print("trace", trace, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
        if trace == "1" and rep == 2:
            lines = [l for l in r.stderr.splitlines() if ("coalescer batch" in l and "n=1:" not in l) or "arrivals" in l]
            print("\n".join(lines[-16:]))
