#!/usr/bin/env python3
"""bench.py — ECDSA P-256 verifies/sec at batch = 2^20 per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    N > 1 without a launcher (WORLD_SIZE unset): bench.py starts its own N ranks through torch.distributed.run on 127.0.0.1 and
    rank 0 prints the line; under the driver's own `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
    it runs as one of the ranks.  --dry-run: the N-rank plumbing over gloo on CPUs (no GPU, no verification, value 0): the CPU
    tier's rehearsal of the launch path.

A step = one pass of the hot path (stage A + stage B kernels, through the C-ABI's
device-pointer entry) over one batch of 2^20 synthetic tuples already resident in HBM.  With
N > 1 ranks the global batch of N * 2^20 tuples is sharded by tuple (weak scaling), and — as
the batch then outgrows one GPU — each step ends with an RCCL all-gather of the per-rank
accept bitmaps (128 KiB per rank).  Prints ONE JSON line on rank 0.

PyTorch is plumbing here (device memory, streams, torch.distributed); the product is libsbv.so.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

ALGO_BYTES_PER_VERIFY = 160.125          # 5 x 32 B in + 1 bit out
HBM_PEAK_GBPS = 8000.0                   # /opt/skills/guides/MI355X_MICROARCH.md (spec)
SEED = 0x5B7F2026
# Integer-multiply roofline (SURVEY.md §8d "int_mul_issue_fraction"): measured peak of v_mad_u64_u32 / v_mad_i64_i32 on
# MI355X (profiles/r01/microbench.jsonl, 8 waves/SIMD) and the multiply-accumulates one mixed addition of the comb
# phases executes on its hot path (p256_pt29.h pt29_madd after the fused reductions: 8 products x 81 + 2 squares x 45 = 738
# product multiply-accumulates, 9 reductions x 59 = 531, 9 for the zero filter — DESIGN.md §4.8; the static ISA mix of the
# kernel is in profiles/r02/isa_stats_r02.txt).  Round 2's earlier runs used the pre-fusion count 1418 and reported 0.64-0.66.
PEAK_LANE_MADS_PER_S = 33.8e12
# comb additions of one grouped tuple's stage B: 13 for u1*G (20-bit comb of G, k_gphase_generic) + 32 key-comb windows and the
# carry window in the 22 % of wavefronts that need it (p256_comb29.h: qphase29_point) = 32.22 for u2*Q (k_verify_keyed_q)
G_ADDS_PER_TUPLE = 13.0
Q_ADDS_PER_TUPLE = 32.22
MADS_PER_MIXED_ADD = 738 + 531 + 9


def cpu_baseline(tuples, n, gpu_bitmap):
    """Reference-side CPU number, timed on this box's host cores on a bounded sample.

    kind = "port": the reference (Go crypto/ecdsa behind api.Verifier) cannot run here (no Go
    toolchain); the oracle is a scalar C restatement, OpenSSL's assembly P-256
    is reported beside it as the closer proxy for Go's nistec assembly.  This is the ONLY place
    bench.py touches oracle/ — as the thing timed for the baseline and as a parity check of the
    sample, never as the measured product."""
    visible = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = visible
    quota = None                        # cgroup v2 CPU quota of the container, in CPUs (None = unlimited / unknown)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    cores = max(1, min(visible, affinity, int(quota + 0.999) if quota else visible))     # threads actually used
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
    lib.sbvo_p256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    probe = min(n, 256)
    out = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    t0 = time.perf_counter()
    lib.sbvo_p256_verify_batch(tuples.ctypes.data, probe, out, 1)
    per_thread = probe / (time.perf_counter() - t0)
    sample = int(min(n, max(1024, per_thread * cores * 2.0))) & ~7       # ~2 s wall, ~2*cores core-seconds
    t0 = time.perf_counter()
    lib.sbvo_p256_verify_batch(tuples.ctypes.data, sample, out, cores)
    dt = time.perf_counter() - t0
    parity = out.raw[:sample // 8] == bytes(gpu_bitmap[:sample // 8])
    res = {"value": sample / dt, "unit": "verifies/s", "cores": cores, "kind": "port",
           "sample": f"first {sample} tuples of the same batch, oracle/p256_oracle.c on {cores} threads, {dt:.2f} s",
           "parity_with_gpu_on_sample": parity, "one_thread_value": per_thread,
           "host": {"cpus_visible": visible, "affinity": affinity, "cgroup_cpu_quota": quota}}
    ssl_path = os.path.join(ROOT, "oracle", "libsbv_openssl.so")
    if os.path.exists(ssl_path):
        ssl = ctypes.CDLL(ssl_path)
        ssl.sbvssl_p256_verify_batch.argtypes = lib.sbvo_p256_verify_batch.argtypes
        s2 = int(min(n, sample * 4)) & ~7
        t0 = time.perf_counter()
        ssl.sbvssl_p256_verify_batch(tuples.ctypes.data, s2, out, cores)
        dt2 = time.perf_counter() - t0
        res["openssl_value"] = s2 / dt2
        res["openssl_parity_with_gpu_on_sample"] = out.raw[:s2 // 8] == bytes(gpu_bitmap[:s2 // 8])
    return res

def _host_cores():
    visible = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = visible
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    return max(1, min(visible, affinity, int(quota + 0.999) if quota else visible))


def cpu_baseline_variant(scheme, tuples, n, gpu_bitmap):
    """CPU baseline of a variant scheme's leg: OpenSSL on this box's host cores over a bounded sample of the SAME
    batch (about 2 s of wall clock), its bitmap compared with the device's on the sample.  kind "port": the reference's stock
    verifier would be Go's crypto/ed25519 (absent: no Go toolchain); secp256k1 is not in Go's standard library at all.
    oracle/ is touched here as the thing timed and as the checker only."""
    ssl_path = os.path.join(ROOT, "oracle", "libsbv_openssl.so")
    if not os.path.exists(ssl_path):
        return {"error": "oracle/libsbv_openssl.so not built (libcrypto absent)"}
    ssl = ctypes.CDLL(ssl_path)
    cores = _host_cores()
    if scheme == "ed25519":
        ssl.sbvssl_ed25519_verify_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        run = lambda m, out, th: ssl.sbvssl_ed25519_verify_gen_batch(SEED, tuples.ctypes.data, 0, m, out, th)   # noqa: E731
        what = "EVP_DigestVerify(Ed25519) on (A, message, R|S) of"
    else:
        ssl.sbvssl_k256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        run = lambda m, out, th: ssl.sbvssl_k256_verify_batch(tuples.ctypes.data, m, out, th)                      # noqa: E731
        what = "ECDSA_do_verify(NID_secp256k1) on"
    out = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    probe = min(n, 64)
    t0 = time.perf_counter()
    run(probe, out, 1)
    per_thread = probe / (time.perf_counter() - t0)
    sample = int(min(n, max(512, per_thread * cores * 2.0))) & ~7
    t0 = time.perf_counter()
    run(sample, out, cores)
    dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "verifies/s", "cores": cores, "kind": "port",
            "sample": f"OpenSSL {what} the first {sample} tuples of the same batch, {cores} threads, {dt:.2f} s",
            "parity_with_gpu_on_sample": out.raw[:sample // 8] == bytes(gpu_bitmap[:sample // 8]), "one_thread_value": per_thread}


def variant_roofline(kernel, bytes_per_verify, lanes, share, dominant_us, dominant_launches):
    """roofline object of a variant leg: the dominant kernel (its Q phase) is charged `share` of a tuple's algorithmic bytes per
    launch — the comb additions one launch executes / all comb additions of a tuple's stage B — over its live lanes, divided by
    its average launch duration (HIP-event pairs on the launch stream inside the leg's timed loop: sbv_profile_read_dominant)."""
    if not dominant_launches:
        return None
    kern_s = dominant_us / dominant_launches * 1e-6
    achieved = bytes_per_verify * lanes * share / kern_s / 1e9
    traffic = None                  # per-launch HBM bytes of this kernel from a rocprofv3 --pmc run over THIS leg (tools/gpu_session.sh pmc, tools/traffic_from_pmc.py)
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(kernel + "_hbm_bytes_per_launch")
    except Exception:
        traffic = None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "kernel": kernel, "avg_launch_us": dominant_us / dominant_launches, "launches": int(dominant_launches), "units_per_launch": int(lanes),
            "share_of_a_tuples_stage_b_per_launch": share,
            "note": "integer-ALU bound like the P-256 step (DESIGN.md 4.8); traffic: PMC counters of a builder session (profiles/traffic.json), null when no pass covered this kernel"}


def leg_warm_cache(sbv, torch, d_tuples, d_bitmap, valid, n, steps, stream):
    """The same batch with the persistent key-table cache ON (the library's default): after the first call the 1024 keys'
    tables are resident, later calls skip the doubling chains and the table kernels.  Reported beside the headline, never
    as the headline (which is measured with the cache off: every step cold)."""
    sbv.key_cache(True)
    try:
        sbv.hot_keys(1024, 0xFFFFFFFF)      # `value`: the 8-bit tables only (no key ever gets hot enough) ...
        sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_bitmap.data_ptr(), stream.cuda_stream)      # fills the cache
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_bitmap.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        entries, hits, misses, cap = sbv.key_cache_stats()
        out = {"value": n * steps / dt, "unit": "verifies/s", "ms_per_step": 1e3 * dt / steps,
               "bitmap_correct": bool((d_bitmap.cpu().numpy() == valid).all()),
               "cache": {"keys_cached": entries, "groups_hit_last_step": hits, "groups_missed_last_step": misses, "capacity": cap}}
        # ... `hot_keys`: the library's default (promotion from 4096 hits on, 16 keys per batch): the batches it takes until every signer
        # owns a 16-bit comb, timed as they go, then the same measurement with the wide pass serving the batch
        try:
            sbv.hot_keys(1024, 4096)
            # (the ramp also ends when three batches in a row promoted nobody: a pool smaller than the signer set)
            ramp, last, still = [], -1, 0
            for _ in range(160):
                t0 = time.perf_counter()
                sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_bitmap.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                ramp.append(time.perf_counter() - t0)
                promoted, pool, wide_lanes, min_hits = sbv.hot_key_stats()
                still = still + 1 if promoted == last and promoted > 0 else 0
                last = promoted
                if promoted >= min(entries, pool) or still >= 3:
                    break
            ramp = ramp[:len(ramp) - still] if still else ramp
            sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_bitmap.data_ptr(), stream.cuda_stream)  # the last promotions are published behind the batch that made them
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_bitmap.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            dth = time.perf_counter() - t0
            promoted, pool, wide_lanes, min_hits = sbv.hot_key_stats()
            out["hot_keys"] = {"value": n * steps / dth, "unit": "verifies/s", "ms_per_step": 1e3 * dth / steps,
                               "bitmap_correct": bool((d_bitmap.cpu().numpy() == valid).all()),
                               "promoted_keys": promoted, "pool_keys": pool, "min_hits": min_hits, "tuples_through_the_wide_pass_last_step": wide_lanes,
                               "batches_until_all_promoted": len(ramp), "keys_cached": entries, "ms_per_batch_while_promoting_median": 1e3 * sorted(ramp)[len(ramp) // 2],
                               "combs_equal_host_builder": [bool(sbv.hot_selfcheck(i)) for i in (0, max(0, promoted - 1))] if promoted else [],
                               "note": "generic tuples, keys inside the tuples: slots with >= min_hits verified tuples own a 16-bit comb (35.7 MB each), built on the device behind the verdicts"}
        except Exception as e:      # noqa: BLE001
            out["hot_keys"] = {"error": repr(e)}
        return out
    finally:
        sbv.hot_keys(1024, 4096)
        sbv.key_cache(False)


def leg_key_count_sweep(sbv, synth, torch, n, stream, key_counts=None, reps=3):
    """Throughput as a function of DISTINCT KEYS (missing #7 / next #3): the headline rests on 1024 signers; a
    VerifyProposal of K mostly-distinct clients (internal/bft/view.go:553-559; config.go:94-98 lets K be anything) lands elsewhere
    on this curve.  n tuples (7/8 valid) over 256 ... n distinct keys, device-resident, cold (key-table cache off: every step builds
    what it uses) and warm (cache on, second pass on), with what the grouping step decided (groups, tuples through tables,
    tuples through the one-lane kernel) and the whole bitmap checked against the generator's at every point."""
    import numpy as np
    out = []
    for K in (key_counts or (256, 1024, 4096, 16384, 65536, 262144, n)):
        K = min(K, n)
        tuples, valid = synth.gen_batch(SEED + 0x900 + K, n, K, 8)
        d_t = torch.from_numpy(tuples).cuda()
        d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
        point = {"keys": K, "signatures_per_key": n / K}
        for mode in ("cold", "warm", "hot"):
            # warm: the cached 8-bit tables alone (promotion threshold out of reach); hot: the default policy once it has settled — up to
            # 1024 of the keys that passed 4096 hits own a 16-bit comb (points with fewer than 1024 tuples per key take a few batches)
            if mode == "hot" and K > 16384:
                continue
            sbv.key_cache(mode != "cold")
            try:
                sbv.hot_keys(1024, 4096 if mode == "hot" else 0xFFFFFFFF)
                sbv.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                if mode == "hot":
                    last = -1
                    for i in range(400):
                        sbv.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
                        torch.cuda.synchronize()
                        promoted = sbv.hot_key_stats()[0]
                        if promoted >= min(K, 1024) or (i > 70 and promoted == last and promoted == 0):
                            break
                        last = promoted
                    sbv.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
                    torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    sbv.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                dt = sorted(ts)[len(ts) // 2]
                g = sbv.last_group_stats()
                point[mode] = {"verifies_per_s": n / dt, "ms": 1e3 * dt, "groups": g[0], "tuples_through_tables": g[1],
                               "tuples_one_lane_kernel": g[2], "bitmap_correct": bool((d_b.cpu().numpy() == valid).all())}
                if mode == "hot":
                    hs = sbv.hot_key_stats()
                    point[mode]["promoted_keys"], point[mode]["tuples_through_the_wide_pass"] = hs[0], hs[2]
            finally:
                sbv.hot_keys(1024, 4096)
                sbv.key_cache(False)
        out.append(point)
        del d_t, d_b
    worst = 0.0
    for a, b in zip(out, out[1:]):
        if b["signatures_per_key"] >= 16:
            worst = max(worst, a["cold"]["verifies_per_s"] / b["cold"]["verifies_per_s"])
    return {"tuples": n, "points": out, "largest_cold_step_between_adjacent_points_down_to_16_sigs_per_key": worst,
            "all_bitmaps_correct": all(p[m]["bitmap_correct"] for p in out for m in ("cold", "warm", "hot") if m in p)}


def leg_all_valid(sbv, synth, torch, n, steps, stream):
    """The same batch shape with every signature valid ("also report all-valid")."""
    import numpy as np
    tuples, valid = synth.gen_batch(SEED + 0x100, n, 1024, 0)
    d_t = torch.from_numpy(tuples).cuda()
    d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
    sbv.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sbv.verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "verifies/s", "ms_per_step": 1e3 * dt / steps,
            "bitmap_correct": bool((d_b.cpu().numpy() == valid).all() and (valid == 0xFF).all())}


def leg_end_to_end(sbv, tuples, valid, n, steps):
    """PCIe-inclusive rate of the host-pointer entry sbv_p256_verify_batch (what a cgo caller uses): tuples in host memory
    when the clock starts, bitmap in host memory when it stops.  Never `value` (inputs there are resident in HBM).
    One submitting thread = latency of a call (upload, then kernels); two threads = the pipelined entry at work (one
    call's upload overlaps the other's kernels: two staging slots, context lock released while waiting)."""
    import threading
    import numpy as np
    out = {}
    for kind in ("pinned", "pinned_key_cache_on", "pageable"):
        ptrs = []
        # "pinned_key_cache_on": the library's DEFAULT configuration (the rest of this run switches the key-table cache off so that the
        # headline is cold).  With it on, one caller's batch goes up in 2^18-tuple pieces beside its own kernels (round 5: sbv_api.hip,
        # sbv_p256_verify_batch through the two staging slots): what a single VerifyProposal caller gets (internal/bft/view.go:555).
        cache_on = kind == "pinned_key_cache_on"
        sbv.key_cache(cache_on)
        try:
            srcs = []
            for _ in range(2):
                if kind.startswith("pinned"):
                    ptr = sbv.host_alloc(n * 160)
                    if not ptr:
                        raise RuntimeError("sbv_host_alloc failed")
                    ptrs.append(ptr)
                    ctypes.memmove(ptr, tuples.ctypes.data, n * 160)
                    srcs.append(ptr)
                else:
                    srcs.append(tuples.ctypes.data)
            got = [np.zeros((n + 7) // 8, dtype=np.uint8) for _ in range(2)]
            sbv.verify_batch_ptr(srcs[0], n, got[0].ctypes.data)          # warm-up (staging buffers)
            res = {}
            if cache_on:
                # ... and the steady state of a node that has seen these signers before: the calls it takes until the hot keys own their wide
                # combs (promotion from 4096 verified tuples on, 64 keys per launch; ~11 ms per promoting launch, once per key) are not timed
                last, calls, still = -1, 0, 0
                for calls in range(1, 81):
                    sbv.verify_batch_ptr(srcs[0], n, got[0].ctypes.data)
                    promoted, pool = sbv.hot_key_stats()[:2]
                    still = still + 1 if promoted == last else 0
                    last = promoted
                    if pool == 0 or promoted >= min(sbv.key_cache_stats()[0], pool) or (calls >= 8 and still >= 8):
                        break
                res["calls_until_hot_keys_settled"] = calls
                res["promoted_keys"] = last
            for threads in (1, 2):
                def work(k):
                    for _ in range(steps):
                        sbv.verify_batch_ptr(srcs[k], n, got[k].ctypes.data)
                th = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                dt = time.perf_counter() - t0
                res[f"{threads}_submitting_thread" + ("s" if threads > 1 else "")] = {
                    "value": n * steps * threads / dt, "unit": "verifies/s", "ms_per_call": 1e3 * dt / steps,
                    "bitmap_correct": bool(all((got[k] == valid).all() for k in range(threads)))}
            tm = sbv.last_timing()
            res["last_call_us"] = {"h2d": tm.h2d_us, "prep": tm.prep_us, "stage_b": tm.verify_us, "d2h": tm.d2h_us, "total": tm.total_us}
            out[kind] = res
        except Exception as e:      # noqa: BLE001 - a secondary leg must not take the headline down
            out[kind] = {"error": repr(e)}
        finally:
            sbv.key_cache(False)
            for ptr in ptrs:
                sbv.host_free(ptr)
    return out


def leg_sharded(sbv, tuples, valid, n, steps):
    """The native multi-device entry sbv_p256_verify_batch_sharded (one process drives every GPU of the node; RCCL all-gather
    of the bitmap shards when the batch is split).  On a one-GPU box it degenerates to one shard on one device; the N-GPU
    scaling curve comes from the driver's `--gpus N` runs (one rank per GPU)."""
    import numpy as np
    ndev = sbv.init_all()
    got = np.zeros((n + 7) // 8, dtype=np.uint8)
    info = sbv.verify_batch_sharded(tuples.ctypes.data, n, got.ctypes.data)
    t0 = time.perf_counter()
    for _ in range(steps):
        info = sbv.verify_batch_sharded(tuples.ctypes.data, n, got.ctypes.data)
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "verifies/s", "devices": ndev, "shards": info.shards,
            "mode": {0: "one device", 1: "RCCL all-gather", 2: "per-device D2H"}.get(info.mode, info.mode),
            "bitmap_correct": bool((got == valid).all()),
            "last_call_us": {"h2d": info.h2d_us, "kernels": info.kernels_us, "gather": info.gather_us, "total": info.total_us}}


def variant_warm_leg(sbv, torch, scheme, call, d_b, expect, n, steps):
    """The same batch of a variant scheme with that scheme's persistent key-table cache ON (sbv_key_cache(scheme), the library's
    default): the first call builds and keeps the 1024 keys' combs, the timed calls find them."""
    sbv.key_cache(True, 0, scheme)
    try:
        call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            call()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        entries, hits, misses, cap = sbv.key_cache_stats(scheme)
        return {"value": n * steps / dt, "unit": "verifies/s", "ms_per_step": 1e3 * dt / steps,
                "bitmap_correct": bool((d_b.cpu().numpy() == expect).all()),
                "cache": {"keys_cached": entries, "groups_hit_last_step": hits, "groups_missed_last_step": misses, "capacity": cap}}
    finally:
        sbv.key_cache(False, 0, scheme)


def leg_ed25519(sbv, torch, n, steps, stream, cpu=False, hot=True):
    """BASELINE.json configs[4]: 2^20 Ed25519 signatures, 1024 keys, 7/8 valid, R|S|A|k tuples resident in HBM."""
    import numpy as np
    cache = f"/tmp/sbv_ed_batch_{n}.npz"
    if os.path.exists(cache):
        z = np.load(cache)
        tuples, expect = z["tuples"], z["expect"]
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hostlib
        h = hostlib.load()
        tuples = np.zeros(n * 128, dtype=np.uint8)
        expect = np.zeros((n + 7) // 8, dtype=np.uint8)
        h.sbvh_ed25519_gen_batch(SEED, n, 1024, 8, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)
        try:
            np.savez(cache, tuples=tuples, expect=expect)
        except Exception:
            pass
    d_t = torch.from_numpy(tuples).cuda()
    d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
    sbv.key_cache(False, 0, sbv.SCHEME_ED25519)      # `value` of this leg is the cold number (every step builds every comb), as the headline's
    sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    sbv.profile_read_dominant(); sbv.profile_read()
    sbv.profile_enable(2)              # event pairs around the dominant kernel's launches only, on the launch stream
    t0 = time.perf_counter()
    for _ in range(steps):
        sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dom_us, dom_launches = sbv.profile_read_dominant()
    sbv.profile_read()
    sbv.profile_enable(False)
    got = d_b.cpu().numpy()
    out = {"metric": "Ed25519 verifies/sec at batch=1M (configs[4])", "value": n * steps / dt, "unit": "verifies/s",
           "ms_per_step": 1e3 * dt / steps, "bitmap_correct": bool((got == expect).all()),
           "algorithmic_GBps": 128.125 * n * steps / dt / 1e9}
    # k_ed_qphase: one launch per chunk of the 32 key-comb windows; a tuple's stage B is 13 additions from the comb of B (G phase)
    # + 32 from the key's comb, so a launch executes (32 / launches per step) / 45 of it.  Lanes = every tuple of the batch but
    # the few whose key repeats too rarely to be grouped (1024 keys x 1024 uses: none).
    per_step = max(1, int(round(dom_launches / max(1, steps))))
    _, lanes, n_ung, n_rej = sbv.last_group_stats()
    if lanes + n_ung + n_rej != n:
        lanes = n
    b_windows = -(-254 // int(os.environ.get("SBV_ED_B_BITS", "20")))      # additions of the G phase: 13 from the 20-bit comb of B (the default)
    out["roofline"] = variant_roofline("k_ed_qphase", 128.125, lanes, (32.0 / per_step) / (32.0 + b_windows), dom_us, dom_launches)
    if cpu:
        out["cpu_baseline"] = cpu_baseline_variant("ed25519", tuples, n, got)
    try:
        try:
            sbv.ed_hot_keys(0, 0)       # the warm figure is the cached 8-bit combs alone; the hot keys have a leg of their own below
        except AttributeError:          # (an older build of the library under tools/ab_lib.sh)
            pass
        out["warm_key_cache"] = variant_warm_leg(sbv, torch, sbv.SCHEME_ED25519,
                                                 lambda: sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream), d_b, expect, n, steps)
    except Exception as e:      # noqa: BLE001
        out["warm_key_cache"] = {"error": repr(e)}
    # the scheme's hot keys (sbv_ed25519_hot_keys, opt-in): the calls it takes until every signer owns a 16-bit comb of -A, then the
    # settled rate — [k](-A) in 16 additions from that comb instead of 32
    if not hot:
        sbv.ed_hot_keys(1024, 4096)
        sbv.key_cache(True, 0, sbv.SCHEME_ED25519)
        return out
    try:
        call = lambda: sbv.ed25519_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)      # noqa: E731
        sbv.key_cache(True, 0, sbv.SCHEME_ED25519)
        sbv.ed_hot_keys(1024, 512)
        calls, promoted, pool = 0, 0, 1
        t0 = time.perf_counter()
        while calls < 48 and promoted < min(pool, 1024):
            call()
            torch.cuda.synchronize()
            calls += 1
            promoted, pool = sbv.ed_hot_key_stats()[:2]
            if pool == 0:
                break
        ramp_s = time.perf_counter() - t0
        if pool:
            call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                call()
            torch.cuda.synchronize()
            dth = time.perf_counter() - t0
            promoted, pool, wide_lanes, min_hits = sbv.ed_hot_key_stats()
            out["hot_keys"] = {"value": n * steps / dth, "unit": "verifies/s", "ms_per_step": 1e3 * dth / steps,
                               "bitmap_correct": bool((d_b.cpu().numpy() == expect).all()), "promoted": promoted, "pool": pool,
                               "tuples_served_wide_last_step": wide_lanes, "min_hits": min_hits, "calls_until_settled": calls,
                               "ramp_s": ramp_s, "comb_bytes": 64 << 20,
                               "combs_equal_host_builder": [bool(sbv.ed_hot_selfcheck(i)) for i in (0, max(0, promoted - 1))] if promoted else []}
        else:
            out["hot_keys"] = {"error": "no room for the pool"}
    except Exception as e:      # noqa: BLE001
        out["hot_keys"] = {"error": repr(e)}
    finally:
        try:
            sbv.ed_hot_keys(1024, 4096)     # the library's default
        except Exception:      # noqa: BLE001  (a library build without the entry: tools/ab_lib.sh against an older one)
            pass
        sbv.key_cache(True, 0, sbv.SCHEME_ED25519)    # the library's default
    return out


def leg_secp256k1(sbv, torch, n, steps, stream, cpu=False):
    """The "other curves" variant: n secp256k1 signatures, 1024 keys, 7/8 valid, 160-byte tuples resident in
    HBM, through the grouped step of this curve (k256_group.h: per-batch key combs; `value` with the curve's key-table cache OFF,
    every step cold, `warm_key_cache` with it on); `one_lane` is the same batch with grouping off (256 doublings per signature).
    Signatures come from the host library's RFC 6979 signer."""
    import numpy as np
    cache = f"/tmp/sbv_k256_batch_{n}.npz"
    if os.path.exists(cache):
        z = np.load(cache)
        tuples, expect = z["tuples"], z["expect"]
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hostlib
        h = hostlib.load()
        tuples = np.zeros(n * 160, dtype=np.uint8)
        expect = np.zeros((n + 7) // 8, dtype=np.uint8)
        h.sbvh_k256_gen_batch(SEED, n, 1024, 8, tuples.ctypes.data, expect.ctypes.data, os.cpu_count() or 1)
        try:
            np.savez(cache, tuples=tuples, expect=expect)
        except Exception:
            pass
    d_t = torch.from_numpy(tuples).cuda()
    d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
    sbv.key_cache(False, 0, sbv.SCHEME_SECP256K1)
    sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    sbv.profile_read_dominant(); sbv.profile_read()
    sbv.profile_enable(2)
    t0 = time.perf_counter()
    for _ in range(steps):
        sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dom_us, dom_launches = sbv.profile_read_dominant()
    sbv.profile_read()
    sbv.profile_enable(False)
    got = d_b.cpu().numpy()
    out = {"metric": "ECDSA secp256k1 verifies/sec at batch=1M (grouped step)", "value": n * steps / dt, "unit": "verifies/s", "tuples": n,
           "ms_per_step": 1e3 * dt / steps, "bitmap_correct": bool((got == expect).all()),
           "algorithmic_GBps": 160.125 * n * steps / dt / 1e9}
    # k_k256_qphase: two launches per step over the key-sorted list; the same accounting as the P-256 headline (13 additions from
    # the 20-bit comb of G + 32.22 from the key's comb per tuple; the ~5 % of tuples with corrupted keys are not on the list)
    per_step = max(1, int(round(dom_launches / max(1, steps))))
    _, lanes, n_ung, n_rej = sbv.last_group_stats()       # the grouping counters are shared by the three schemes' grouped steps
    if lanes + n_ung + n_rej != n:
        lanes = n
    out["roofline"] = variant_roofline("k_k256_qphase", 160.125, lanes, (Q_ADDS_PER_TUPLE / per_step) / (G_ADDS_PER_TUPLE + Q_ADDS_PER_TUPLE),
                                       dom_us, dom_launches)
    if cpu:
        out["cpu_baseline"] = cpu_baseline_variant("secp256k1", tuples, n, got)
    try:
        out["warm_key_cache"] = variant_warm_leg(sbv, torch, sbv.SCHEME_SECP256K1,
                                                 lambda: sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream), d_b, expect, n, steps)
    except Exception as e:      # noqa: BLE001
        out["warm_key_cache"] = {"error": repr(e)}
    finally:
        sbv.key_cache(True, 0, sbv.SCHEME_SECP256K1)
    sbv.set_grouping(False)
    try:
        sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sbv.secp256k1_verify_batch_dev(d_t.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t0
        out["one_lane"] = {"value": n / d1, "ms_per_step": 1e3 * d1, "bitmap_correct": bool((d_b.cpu().numpy() == expect).all())}
    finally:
        sbv.set_grouping(True)
    return out


def leg_proposals():
    """BASELINE.json configs[2] at its stated size: VerifyProposal of a K = 10 000-request proposal (4 nodes, f = 1) through
    the C++ api.Verifier mirror (internal/bft/view.go:553-559 is the call site), host pointers in, verdict out — (a) client
    keys registered on the device (comb slots, raw-messages front end), (b) client keys NOT registered: generic tuples with the
    key inline, grouped by key inside the batch, tables kept by the key cache between proposals.  Median of 3 proposals."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostlib
    lib = hostlib.load()
    cb = hostlib.BACKEND_FN(lambda *a: -1)
    import consensus_amd as sbv
    out = {"K": 10000, "n_nodes": 4, "unit": "us", "note": "median of 3 proposals; the key-table cache is on, as in a running node (the library's default; "
           "the headline leg switches it off to stay cold): the first proposal builds the 64 clients' combs, the later ones find them"}
    sbv.key_cache(True)
    try:
        for label, on_device in (("registered_client_keys", 1), ("unregistered_client_keys", 0)):
            v = lib.sbvh_verifier_new(0, 0, cb, None, 1 << 20, 50, 0)
            lib.sbvh_set_device_client_keys(v, on_device)
            res = hostlib.ReplayResult()
            rc = lib.sbvh_replay(v, 4, 10000, 3, 0, min(64, os.cpu_count() or 8), ctypes.byref(res))
            lib.sbvh_verifier_free(v)
            out[label] = {"verify_proposal_us": res.verify_proposal_us, "request_sigs_per_s": 10000 / (res.verify_proposal_us * 1e-6) if res.verify_proposal_us else None,
                          "prev_commits_serial_us": res.prev_commits_us, "commit_quorum_us": res.commit_quorum_us} if rc == 0 and res.status == 0 else {"error": f"rc {rc} status {res.status}"}
    finally:
        sbv.key_cache(False)
    return out


def leg_replay_550k(sbv, synth):
    """BASELINE.json configs[3] at its stated size: 50 000 proposals x 11 consenter signatures (N = 16: f = 5, Q = 11,
    internal/bft/util.go:183-187) as ONE call of the native multi-device entry with group = 11, quorum = 10 (the Q - 1 votes a
    replica needs besides its own: view.go:531) — host pointers in, accept bitmap + per-proposal quorum bits out (>= 10
    accepted signatures by distinct keys, viewchanger.go:681-727).  16 signer keys, 1/8 of the signatures corrupted.  The
    quorum bits are checked against bits recomputed from the oracle's verdicts on the first 10 000 proposals, the accept
    bitmap against the generator's expectation on all 550 000."""
    import numpy as np
    group, quorum, props = 11, 10, 50000
    n = group * props
    tuples, valid = synth.gen_batch(SEED + 0x300, n, 16, 8)
    got = np.zeros((n + 7) // 8, dtype=np.uint8)
    qgot = np.zeros((props + 7) // 8, dtype=np.uint8)
    sbv.init_all()
    sbv.key_cache(True)          # a replaying replica sees the same 16 consenters proposal after proposal
    try:
        info = sbv.verify_batch_sharded(tuples.ctypes.data, n, got.ctypes.data, group, quorum, qgot.ctypes.data)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            info = sbv.verify_batch_sharded(tuples.ctypes.data, n, got.ctypes.data, group, quorum, qgot.ctypes.data)
            ts.append(time.perf_counter() - t0)
    finally:
        sbv.key_cache(False)
    dt = sorted(ts)[1]
    bits = np.unpackbits(got, bitorder="little")[:n].reshape(props, group)
    qbits = np.unpackbits(qgot, bitorder="little")[:props]
    # independent statement of the rule on the generator's expectation (keys are distinct inside a proposal: tuple j is signed
    # by key j % 16 and 11 consecutive tuples never repeat one; a corrupted key cannot verify)
    want_bits = np.unpackbits(valid, bitorder="little")[:n].reshape(props, group)
    want_q = (want_bits.sum(axis=1) >= quorum).astype(np.uint8)
    sample_props = 10000
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
    lib.sbvo_p256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    ob = np.zeros((sample_props * group + 7) // 8, dtype=np.uint8)
    lib.sbvo_p256_verify_batch(tuples.ctypes.data, sample_props * group, ob.ctypes.data, os.cpu_count() or 1)
    oq = (np.unpackbits(ob, bitorder="little")[:sample_props * group].reshape(sample_props, group).sum(axis=1) >= quorum).astype(np.uint8)
    return {"proposals": props, "signatures": n, "group": group, "quorum": quorum, "sigs_per_s": n / dt, "ms_per_call": 1e3 * dt,
            "proposals_with_quorum": int(qbits.sum()), "accept_bitmap_correct": bool((bits == want_bits).all()),
            "quorum_bits_equal_generator_rule": bool((qbits == want_q).all()),
            "quorum_bits_equal_oracle_on_first_10000_proposals": bool((qbits[:sample_props] == oq).all()),
            "devices": info.devices, "shards": info.shards,
            "last_call_us": {"h2d": info.h2d_us, "kernels": info.kernels_us, "gather": info.gather_us, "total": info.total_us},
            "note": "PCIe-inclusive (88 MB of tuples from host memory per call); key-table cache warm as for a replaying replica"}


def leg_replay_550k_keyed(sbv, synth):
    """configs[3] through the REGISTERED-key sharded entry (round 5: sbv_p256_verify_batch_keyed_sharded): the same 50 000 x 11
    signatures as replay_550k, as a replica that registered its 16 consenters ships them — 96-byte records r | s | hash + a 4-byte
    key slot from page-locked host memory (100 B per signature over PCIe instead of 160), the consenters' wide combs resident on
    every device, uploads in pieces beside the kernels, quorum bits by distinct slot.  PCIe-inclusive, host pointers in and out."""
    import numpy as np
    group, quorum, props = 11, 10, 50000
    n = group * props
    tuples, valid = synth.gen_batch(SEED + 0x300, n, 16, 8)
    t2 = tuples.reshape(n, 160)
    keys, counts = np.unique(t2[:, 96:160], axis=0, return_counts=True)
    keys = keys[counts >= 64]
    sbv.init_all()
    sbv.clear_keys()
    reg = sbv.register_keys([bytes(k) for k in keys])
    sbv.widen_keys(reg)                                        # what RegisterConsenter does: 16 keys -> 20-bit combs
    slots_of = dict(zip((bytes(k) for k in keys), reg))
    slots = np.fromiter((slots_of.get(bytes(k), 0xFFFFFFFF) for k in t2[:, 96:160]), dtype=np.uint32, count=n)
    p_rsh, p_slots = sbv.host_alloc(n * 96), sbv.host_alloc(n * 4)
    if not p_rsh or not p_slots:
        raise RuntimeError("sbv_host_alloc failed")
    try:
        h_rsh = np.ctypeslib.as_array((ctypes.c_uint8 * (n * 96)).from_address(p_rsh))
        h_slots = np.ctypeslib.as_array((ctypes.c_uint32 * n).from_address(p_slots))
        h_rsh[:] = np.ascontiguousarray(t2[:, :96]).reshape(-1)
        h_slots[:] = slots
        got = np.zeros((n + 7) // 8, dtype=np.uint8)
        qgot = np.zeros((props + 7) // 8, dtype=np.uint8)
        info = sbv.verify_batch_keyed_sharded(p_rsh, p_slots, n, got.ctypes.data, group, quorum, qgot.ctypes.data)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            info = sbv.verify_batch_keyed_sharded(p_rsh, p_slots, n, got.ctypes.data, group, quorum, qgot.ctypes.data)
            ts.append(time.perf_counter() - t0)
        wide = sbv.wide_key_stats()
    finally:
        sbv.host_free(p_rsh); sbv.host_free(p_slots)
        sbv.wide_keys()
        sbv.clear_keys()
    dt = sorted(ts)[len(ts) // 2]
    bits = np.unpackbits(got, bitorder="little")[:n].reshape(props, group)
    qbits = np.unpackbits(qgot, bitorder="little")[:props]
    want_bits = np.unpackbits(valid, bitorder="little")[:n].reshape(props, group)
    want_q = (want_bits.sum(axis=1) >= quorum).astype(np.uint8)       # the 11 signers of a proposal are distinct by construction
    return {"proposals": props, "signatures": n, "group": group, "quorum": quorum, "sigs_per_s": n / dt, "ms_per_call": 1e3 * dt,
            "best_ms_per_call": 1e3 * min(ts), "proposals_with_quorum": int(qbits.sum()),
            "accept_bitmap_correct": bool((bits == want_bits).all()), "quorum_bits_equal_generator_rule": bool((qbits == want_q).all()),
            "devices": info.devices, "shards": info.shards, "wide_keys": wide[0], "wide_bits": wide[1],
            "last_call_us": {"h2d": info.h2d_us, "kernels": info.kernels_us, "gather": info.gather_us, "total": info.total_us},
            "note": "PCIe-inclusive (55 MB of records + slots from page-locked host memory per call), registered consenter keys with their wide combs"}


def leg_consenter_keys(sbv, synth, torch, stream, steps):
    """configs[3]'s signatures as a replica that has REGISTERED its consenters verifies them (VerifyConsenterSig /
    VerifyConsenterSigBatch: internal/bft/view.go:631, 834; decision replay controller.go:587-633): 550 000 records r|s|hash +
    key slot, 16 keys, resident in HBM, through sbv_p256_verify_batch_keyed_dev — with the 8-bit combs every registered key has
    (13 + 32.2 additions) and with the consenters' wide combs (sbv_p256_widen_keys, round 4: 13 + 13 additions with the 20-bit combs a
    16-key set gets by default, built on the device)."""
    import numpy as np
    group, props = 11, 50000
    n = group * props
    tuples, valid = synth.gen_batch(SEED + 0x300, n, 16, 8)
    t2 = tuples.reshape(n, 160)
    keys, counts = np.unique(t2[:, 96:160], axis=0, return_counts=True)
    keys = keys[counts >= 64]
    sbv.clear_keys()
    reg = sbv.register_keys([bytes(k) for k in keys])
    slots_of = dict(zip((bytes(k) for k in keys), reg))
    slots = np.fromiter((slots_of.get(bytes(k), 0xFFFFFFFF) for k in t2[:, 96:160]), dtype=np.uint32, count=n)
    d_rsh = torch.from_numpy(np.ascontiguousarray(t2[:, :96]).reshape(-1)).cuda()
    d_slots = torch.from_numpy(slots).cuda()
    d_b = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")

    def timed():
        sbv.verify_batch_keyed_dev(d_rsh.data_ptr(), d_slots.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        sbv.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            sbv.verify_batch_keyed_dev(d_rsh.data_ptr(), d_slots.data_ptr(), n, d_b.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kp, kv, kl = sbv.profile_read()
        sbv.profile_enable(False)
        return {"sigs_per_s": n * steps / dt, "ms_per_call": 1e3 * dt / steps,
                "kernel_us": {"k_p256_prep_keyed": kp / max(1, kl), "k_p256_verify_keyed": kv / max(1, kl)},
                "bitmap_correct": bool((d_b.cpu().numpy() == valid).all())}

    out = {"signatures": n, "proposals": props, "group": group, "distinct_keys": int(len(keys))}
    try:
        out["combs_8bit"] = timed()
        bits = int(os.environ.get("SBV_BENCH_WIDE_BITS", "1"))          # 1 = the library's default policy: 20 bits up to 16 wide keys, 16 beyond
        sbv.wide_keys(bits, 64)
        t0 = time.perf_counter()
        sbv.widen_keys(reg)
        build_s = time.perf_counter() - t0
        wide, wbits, wmax, kib = sbv.wide_key_stats()
        out["combs_wide"] = dict(timed(), bits=wbits, wide_keys=wide, MiB_per_key=kib / 1024.0, build_s=build_s)
        out["speedup"] = out["combs_wide"]["sigs_per_s"] / out["combs_8bit"]["sigs_per_s"]
    finally:
        sbv.wide_keys()              # back to the default policy
        sbv.clear_keys()
    out["note"] = ("device-resident records (60.6 MB), 1/8 corrupted; the sharded generic entry's figure for the same signatures is "
                   "replay_550k (PCIe-inclusive, key-table cache)")
    return out


def leg_front_end(sbv, tuples, valid, n_all):
    """SURVEY §8f row 1 measured end to end: raw messages + DER signatures + key slots in HOST memory -> accept bitmap in host
    memory through sbv_p256_verify_msgs_keyed (SHA-256 and the strict DER parse run on the device).  2^18 signatures of the
    headline batch: message j is rebuilt from the generator's rule, (r, s) re-encoded as DER; a tuple whose corruption hit the
    hash gets a flipped message bit instead, one whose corruption hit the key keeps its signature but has no slot."""
    import hashlib
    import numpy as np
    n = min(n_all, 1 << 18)
    t2 = tuples.reshape(n_all, 160)[:n]
    keys, counts = np.unique(tuples.reshape(n_all, 160)[:, 96:160], axis=0, return_counts=True)
    keys = keys[counts >= 64]
    sbv.clear_keys()
    slots_of = dict(zip((bytes(k) for k in keys), sbv.register_keys([bytes(k) for k in keys])))
    slots = [slots_of.get(bytes(k), 0xFFFFFFFF) for k in t2[:, 96:160]]
    seed = SEED
    msgs, sigs = [], []

    def der_int(b):
        b = b.lstrip(b"\0") or b"\0"
        if b[0] & 0x80:
            b = b"\0" + b
        return b"\x02" + bytes([len(b)]) + b

    vbits = np.unpackbits(valid, bitorder="little")
    for j in range(n):
        m = bytearray(32)
        m[0:7] = b"sbv-msg"
        m[8:12] = seed.to_bytes(4, "big")
        m[24:32] = j.to_bytes(8, "big")
        row = t2[j].tobytes()
        if not vbits[j] and hashlib.sha256(bytes(m)).digest() != row[64:96]:
            m[16] ^= 1                      # the generator flipped a bit of the hash: any other message does
        body = der_int(row[0:32]) + der_int(row[32:64])
        msgs.append(bytes(m))
        sigs.append(b"\x30" + bytes([len(body)]) + body)
    got = sbv.verify_msgs_keyed(msgs, sigs, slots)           # warm-up (staging buffers) and the verdicts
    ok = bytes(got) == bytes(valid[:n // 8])
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        sbv.verify_msgs_keyed(msgs, sigs, slots)
        ts.append(time.perf_counter() - t0)
    tm = sbv.last_timing()
    sbv.clear_keys()
    dt = sorted(ts)[1]
    return {"messages": n, "msgs_per_s": n / dt, "ms_per_call": 1e3 * dt, "bitmap_correct": ok,
            "device_us": {"h2d": tm.h2d_us, "front_end_and_stage_a": tm.prep_us, "stage_b": tm.verify_us, "d2h": tm.d2h_us, "total": tm.total_us},
            "msgs_per_s_device_side": n / (tm.total_us * 1e-6) if tm.total_us else None,
            "note": "ms_per_call includes the Python binding's own packing of 2^18 byte strings; device_us is the library's split of the last call"}


def leg_projected_strong_scaling(sbv, torch, d_tuples, valid, n, stream, base_ms):
    """PROJECTION, not a measurement of several GPUs: the one visible GPU runs, one after another, what each of G devices
    would run for the SAME 2^20 batch, device-resident, key cache off (cold), and max over the parts stands for the step
    time of a G-GPU node.  Two partitions: `by_key` (sbv_p256_verify_batch_dev_part: device g verifies the tuples of the
    keys that hash to it — K / G tables, n / G tuples) and `contiguous` (device g verifies tuples [g n / G, (g+1) n / G) —
    all K tables on every device: the round-2 design).  Not included: PCIe (every device of the by-key form receives the
    whole batch; of the contiguous form 1 / G of it), the bitmap exchange (128 KiB: one all-reduce / all-gather), clock and
    HBM contention between devices (none: they are separate packages)."""
    import numpy as np
    want = np.unpackbits(valid, bitorder="little")[:n]
    out = {"base_ms_one_gpu_cold": base_ms, "label": "projection from one GPU running the parts sequentially", "partitions": {},
           "note": "the warm rows are the cached 8-bit tables alone: hot-key promotion is held off for this leg (a promotion inside five timed repetitions would be noise, and base and parts must run the same kernels)"}
    sbv.hot_keys(1024, 0xFFFFFFFF)
    words = (n + 31) // 32
    d_full = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")

    def one_gpu_warm_ms():
        sbv.key_cache(True)
        try:
            ts = []
            for rep in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_full.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return 1e3 * sorted(ts[1:])[1]
        finally:
            sbv.key_cache(False)
    warm_base = one_gpu_warm_ms()
    out["base_ms_one_gpu_warm"] = warm_base
    for G in (2, 4, 8):
        row = {}
        for label, warm, base in (("by_key", False, base_ms), ("by_key_warm_key_cache", True, warm_base)):
            acc = np.zeros(n, dtype=np.uint8)
            t_key = []
            sbv.key_cache(warm)
            try:
                for g in range(G):
                    d_w = torch.zeros(words, dtype=torch.int32, device="cuda")
                    ts = []
                    for rep in range(4):               # rep 0 warms allocations (and, with the cache on, this part's tables)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        sbv.verify_batch_dev_part(d_tuples.data_ptr(), n, g, G, d_w.data_ptr(), stream.cuda_stream)
                        torch.cuda.synchronize()
                        ts.append(time.perf_counter() - t0)
                    t_key.append(sorted(ts[1:])[1])
                    acc |= np.unpackbits(d_w.cpu().numpy().view(np.uint8), bitorder="little")[:n]
            finally:
                sbv.key_cache(False)
            row[label] = {"max_part_ms": 1e3 * max(t_key), "mean_part_ms": 1e3 * sum(t_key) / G, "bitmap_correct": bool((acc == want).all()),
                          "projected_speedup": base / (1e3 * max(t_key))}
        per = (n // G + 511) // 512 * 512
        t_con = []
        ok = True
        for g in range(G):
            lo, hi = g * per, min(n, (g + 1) * per)
            d_b = torch.zeros((hi - lo + 7) // 8, dtype=torch.uint8, device="cuda")
            ts = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sbv.verify_batch_dev(d_tuples.data_ptr() + 160 * lo, hi - lo, d_b.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t_con.append(sorted(ts[1:])[1])
            ok = ok and bool((np.unpackbits(d_b.cpu().numpy(), bitorder="little")[:hi - lo] == want[lo:hi]).all())
        row["contiguous"] = {"max_part_ms": 1e3 * max(t_con), "bitmap_correct": ok, "projected_speedup": base_ms / (1e3 * max(t_con))}
        out["partitions"][f"G={G}"] = row
    sbv.hot_keys(1024, 4096)
    return out


# upper bucket edges of the reference's LatencyBatchProcessing histogram (pkg/api/metrics.go:427-435), in microseconds: what a node's
# own metrics would show of these latencies — everything measured here lands in the first bucket (<= 5 ms), which is why the line also
# carries percentiles
LATENCY_BUCKETS_S = [0.005, 0.01, 0.015, 0.05, 0.1, 1, 10]
LATENCY_BUCKETS_US = [e * 1e6 for e in LATENCY_BUCKETS_S]


def dist_summary(samples_us, what):
    xs = sorted(samples_us)
    n = len(xs)
    pick = lambda q: xs[min(n - 1, int(q * n))]            # noqa: E731
    counts, prev = [], 0
    for edge in LATENCY_BUCKETS_US:
        c = sum(1 for x in xs if x <= edge)
        counts.append(c - prev)
        prev = c
    return {"n": n, "p50": pick(0.50), "p90": pick(0.90), "p99": pick(0.99), "max": xs[-1], "min": xs[0], "mean": sum(xs) / n, "unit": "us",
            "reference_histogram_buckets_le_s": LATENCY_BUCKETS_S, "reference_histogram_counts": counts, "what": what}


def leg_m2(tuples, n):
    """BASELINE.json's second metric — commit-quorum latency at N = 16 (Q = 11): wall time from "15 commit signatures in
    host memory" to ">= 10 accepted" (SURVEY.md §8d M2).  (a) gpu: the 15 concurrent VerifyConsenterSig calls of
    View.processCommits (view.go:537-541) coalesced into one micro-batch through the C++ api.Verifier mirror; (b) cpu: the
    same 15 verifications on 15 host threads (OpenSSL ECDSA_do_verify and the oracle port — proxies for stock crypto/ecdsa
    goroutines); (c) hybrid: what a Verifier with the Go adapter's gpuMin route picks for a quorum-sized batch."""
    out = {}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hostlib
        lib = hostlib.load()
        cb = hostlib.BACKEND_FN(lambda *a: -1)
        v = lib.sbvh_verifier_new(0, 0, cb, None, 1 << 20, 50, 0)
        res = hostlib.ReplayResult()
        rc = lib.sbvh_replay(v, 16, 10, 15, 0, min(64, os.cpu_count() or 8), ctypes.byref(res))
        lib.sbvh_verifier_free(v)
        out["gpu"] = res.commit_quorum_us if rc == 0 and res.status == 0 else None
        # the DISTRIBUTION: 1000 sequences of the same harness (K = 10 requests per proposal, so that a sequence is
        # VerifyProposal -> previous commits -> the burst of 15 votes, as a View runs it), every sequence's figure kept
        try:
            v = lib.sbvh_verifier_new(0, 0, cb, None, 1 << 20, 50, 0)
            nseq = 1000
            q_us = (ctypes.c_double * nseq)()
            p_us = (ctypes.c_double * nseq)()
            res2 = hostlib.ReplayResult()
            rc2 = lib.sbvh_replay_samples(v, 16, 10, nseq, min(64, os.cpu_count() or 8), ctypes.byref(res2), q_us, p_us)
            lib.sbvh_verifier_free(v)
            if rc2 == 0 and res2.status == 0:
                out["gpu_distribution"] = dist_summary(list(q_us), "15 concurrent VerifyConsenterSig -> 10 accepted, N = 16, through the Verifier's coalescer")
                out["verify_proposal_k10_distribution"] = dist_summary(list(p_us), "VerifyProposal of the same sequences (10 request signatures: the latency form of the registered-key path)")
        except Exception as e:      # noqa: BLE001
            out["gpu_distribution"] = {"error": repr(e)}
        out["gpu_note"] = ("median over 15 sequences; 15 warm voter threads call VerifyConsenterSig at once, the leader-combining coalescer hands the burst to "
                           "the one-launch latency form (stage A of the quorum on the host with one inversion, 16 lanes per signature, mapped host memory in and out); registered consenter keys")
    except Exception as e:      # noqa: BLE001
        out["gpu_error"] = repr(e)
    import numpy as np
    valid15 = np.ascontiguousarray(tuples[:15 * 160])            # 15 signatures of the headline batch (mostly valid)
    bm = ctypes.create_string_buffer(8)
    for name, so, fn in (("cpu_15_threads_openssl", "libsbv_openssl.so", "sbvssl_p256_verify_batch"),
                         ("cpu_15_threads_oracle_port", "libsbv_oracle.so", "sbvo_p256_verify_batch")):
        path = os.path.join(ROOT, "oracle", so)
        if not os.path.exists(path):
            continue
        f = getattr(ctypes.CDLL(path), fn)
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        ts = []
        for _ in range(25):
            t0 = time.perf_counter()
            f(valid15.ctypes.data, 15, bm, 15)
            ts.append(1e6 * (time.perf_counter() - t0))
        out[name] = sorted(ts)[len(ts) // 2]
        ts1 = []
        for _ in range(1000):
            t0 = time.perf_counter()
            f(valid15.ctypes.data, 1, bm, 1)
            ts1.append(1e6 * (time.perf_counter() - t0))
        out[name.replace("cpu_15_threads", "cpu_one_verify")] = sorted(ts1)[len(ts1) // 2]
        out[name.replace("cpu_15_threads", "cpu_one_verify") + "_distribution"] = dist_summary(ts1, "one verification on one warm core, 1000 in a row")
    one = out.get("cpu_one_verify_openssl") or out.get("cpu_one_verify_oracle_port")
    if one is not None and out.get("gpu") is not None:
        # What a Verifier that may use either pays for the quorum: 15 goroutines on 15 free cores finish in ONE verification's
        # time; the Go adapter's GPUMin decides which side a burst goes to (go/gpuverifier/verifier.go), and for a burst of 15 it
        # is whichever of the two numbers is smaller on the box at hand.
        out["hybrid"] = min(one, out["gpu"])
        out["hybrid_route"] = "cpu" if one <= out["gpu"] else "gpu"
        out["hybrid_note"] = ("min(one CPU verification = 15 running goroutines on 15 free cores, GPU micro-batch); the 15-pthread-spawn "
                              "figures are not used for it (they are dominated by thread start-up)")
    out["cpu_note"] = ("cpu_15_threads_* spawn 15 pthreads per measurement (the checker libraries have no thread pool), so they "
                       "carry ~0.3-0.5 ms of thread start-up; cpu_one_verify_* is one verification on one core = what 15 already "
                       "running goroutines on 15 free cores would need")
    out["unit"] = "us"
    return out


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher (VERDICT r5 #1a: this form used to die with
    SystemExit before touching a GPU).  One rank per GPU on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this driver
    env["SBV_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_run(args, world, rank):
    """The N-rank plumbing of the scaling run without a GPU: gloo, each rank's shard of the global batch, the all-gather of the bitmap
    shards, the max-over-ranks clock, rank 0's one JSON line.  NOTHING is verified (libsbv.so has no CPU path and the oracle is not the
    product): every rank contributes the generator's own expected bitmap; `value` is 0 and the line says dry_run."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    if world > 1:
        dist.init_process_group(backend="gloo")
    n = args.tuples
    tuples, valid = shard_of_rank(synth, np, dist if world > 1 else None, rank, n)
    nbytes = (n + 7) // 8
    mine = torch.from_numpy(np.ascontiguousarray(valid))
    gathered = torch.zeros(nbytes * world, dtype=torch.uint8)
    for _ in range(args.warmup):
        if world > 1:
            dist.all_gather_into_tensor(gathered, mine)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.all_gather_into_tensor(gathered, mine)
        else:
            gathered[:nbytes] = mine
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        allt = torch.zeros(world, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, t)
        per_rank = [float(x) for x in allt]
    ok = bool((gathered[rank * nbytes:(rank + 1) * nbytes].numpy() == valid).all())
    if rank == 0:
        print(json.dumps({
            "metric": "ECDSA P-256 verifies/sec at batch=1M", "value": 0.0, "unit": "verifies/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * max(per_rank) / max(1, args.steps), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic", "dry_run": True,
            "config": {"workload": "DRY RUN of the N-rank launch path over gloo: no GPU, no verification", "tuples_per_gpu": n,
                       "global_batch": n * world, "parallelism": f"shard-by-tuple x{world} + all-gather of bitmaps (gloo)"},
            "per_rank_ms_per_step": [1e3 * x / max(1, args.steps) for x in per_rank], "bitmap_correct": ok,
            "self_launched": os.environ.get("SBV_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def shard_of_rank(synth, np, dist, rank, n):
    """Rank r's shard of the global batch: the synthetic batch of SURVEY 8d (seed 0x5B7F2026: 1024 signers, 7/8 valid), its tuples
    rotated by r * 4099 positions — every GPU of the node sees the same signers, as the replicas of one cluster do, in its own order.
    Generating 2^20 signatures takes ~30 s of every host core, so rank 0 generates (or finds the /tmp cache) and the others load it."""
    if dist is not None and rank != 0:
        dist.barrier()
    tuples, valid = synth.gen_batch(SEED, n)
    if dist is not None and rank == 0:
        dist.barrier()
    if rank:
        k = (rank * 4099) % n
        t2 = np.roll(tuples.reshape(n, 160), -k, axis=0).reshape(-1)
        bits = np.unpackbits(valid, bitorder="little")[:n]
        valid = np.packbits(np.roll(bits, -k), bitorder="little")
        tuples = np.ascontiguousarray(t2)
    return tuples, valid


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tuples", type=int, default=1 << 20, help="tuples per GPU (default: the headline 2^20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--primary-only", action="store_true", help="skip the secondary legs (grouping off, registered keys): quick A/B runs")
    ap.add_argument("--warm-leg", action="store_true", help="with --primary-only: still run the warm-key-cache leg")
    ap.add_argument("--legs", default="", help="comma-separated names of the secondary legs to run (default: all of them); dev sessions")
    ap.add_argument("--dry-run", action="store_true", help="the N-rank launch path over gloo on CPUs: no GPU, nothing verified (CPU tier)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    import consensus_amd as sbv
    import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus disagree")
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: bench.py measures the HIP path only (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    t_init = time.perf_counter()
    sbv.init(local_rank)
    init_s = time.perf_counter() - t_init      # host tables (once per process) + this device's two comb uploads
    sbv.key_cache(False)       # the headline is the COLD number: every step rebuilds every key's tables (nothing cached between steps)

    n = args.tuples
    tuples, valid = shard_of_rank(synth, np, dist if world > 1 else None, rank, n)            # rank r holds shard r of the global batch
    d_tuples = torch.from_numpy(tuples).cuda()
    nbytes = (n + 7) // 8
    d_bitmap = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    d_all = torch.zeros(nbytes * world, dtype=torch.uint8, device="cuda") if world > 1 else None
    stream = torch.cuda.current_stream()

    gather_events = []

    def step(timed=False):
        sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_bitmap.data_ptr(), stream.cuda_stream)
        if world > 1:
            # the batch outgrew one GPU: RCCL all-gather of the per-rank accept bitmaps (128 KiB per rank) over xGMI
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                dist.all_gather_into_tensor(d_all, d_bitmap)
                e1.record(stream)
                gather_events.append((e0, e1))
            else:
                dist.all_gather_into_tensor(d_all, d_bitmap)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Inside the timed region only the dominant kernel is bracketed by HIP events (level 2: a pair per launch, on the launch
    # stream).  Every recorded event is a packet between two kernels of the step; the stage-A / stage-B split of `kernel_us`
    # comes from two extra, untimed steps below.  SBV_BENCH_PROFILE=0 drops the events altogether (A/B of their cost only:
    # the line then carries no roofline).
    prof_level = 0 if os.environ.get("SBV_BENCH_PROFILE") == "0" else 2
    sbv.profile_enable(prof_level)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    gather_ms = sum(a.elapsed_time(b) for a, b in gather_events) / len(gather_events) if gather_events else 0.0
    dominant_us, dominant_launches = sbv.profile_read_dominant()
    sbv.profile_read()
    sbv.profile_enable(True)
    for _ in range(2):
        step()
    fence()
    d2_us, d2_launches = sbv.profile_read_dominant()
    dom_steps = args.steps
    if dominant_launches == 0:              # SBV_BENCH_PROFILE=0: the dominant kernel's duration comes from the untimed pass
        dominant_us, dominant_launches, dom_steps = d2_us, d2_launches, 2
    prep_us, verify_us, split_launches = sbv.profile_read()
    prep_us, verify_us = prep_us / max(1, split_launches), verify_us / max(1, split_launches)
    launches = args.steps * ((n + (1 << 21) - 1) >> 21)          # calls x chunks of 2^21 tuples
    sbv.profile_enable(False)
    groups, n_grouped, n_ungrouped, n_key_rejected = sbv.last_group_stats()
    was_grouped = (n_grouped + n_ungrouped + n_key_rejected) == n

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    per_rank_ms = [1e3 * my_elapsed / args.steps]
    per_rank_dom_us = [dominant_us / max(1, dominant_launches)]
    if world > 1:
        allt = torch.zeros(2 * world, dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(allt, torch.tensor([per_rank_ms[0], per_rank_dom_us[0]], dtype=torch.float64, device="cuda"))
        per_rank_ms = [float(x) for x in allt[0::2]]
        per_rank_dom_us = [float(x) for x in allt[1::2]]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    got = d_bitmap.cpu().numpy()
    ok = bool((got == valid).all())

    # Secondary measurement (NOT part of `value`): the same batch with in-step key grouping switched off,
    # i.e. what a batch of 2^20 all-distinct keys costs (every tuple through the generic doubling kernel).
    ungrouped = None
    if world == 1 and not args.primary_only:
        try:
            sbv.set_grouping(False)
            sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_bitmap.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            sbv.profile_enable(True)
            tu = time.perf_counter()
            for _ in range(args.steps):
                sbv.verify_batch_dev(d_tuples.data_ptr(), n, d_bitmap.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            tu = time.perf_counter() - tu
            up, uv, ul = sbv.profile_read()
            sbv.profile_enable(False)
            ungrouped = {"value": n * args.steps / tu, "unit": "verifies/s",
                         "kernel_us": {"k_p256_prep": up / max(1, ul), "k_p256_verify": uv / max(1, ul)},
                         "bitmap_correct": bool((d_bitmap.cpu().numpy() == valid).all()),
                         "note": "sbv_p256_set_grouping(0): no key reuse exploited; the rate for all-distinct keys"}
        except Exception as e:
            ungrouped = {"error": repr(e)}
        finally:
            sbv.set_grouping(True)

    # Secondary measurement (NOT part of `value`): the same signatures through the registered-key entry
    # (key slots instead of inline public keys: what VerifyConsenterSig / decision replay use).
    keyed = None
    if world == 1 and not args.primary_only:
        try:
            t2 = tuples.reshape(n, 160)
            keys, counts = np.unique(t2[:, 96:160], axis=0, return_counts=True)
            keys = keys[counts >= 64]           # the signer pool; bit-flipped keys (rare repeats at most) get no slot = reject
            if len(keys) <= 8192:
                sbv.clear_keys()
                slots_of = dict(zip((bytes(k) for k in keys), sbv.register_keys([bytes(k) for k in keys])))
                slots = np.fromiter((slots_of.get(bytes(k), 0xFFFFFFFF) for k in t2[:, 96:160]), dtype=np.uint32, count=n)
                d_rsh = torch.from_numpy(np.ascontiguousarray(t2[:, :96]).reshape(-1)).cuda()
                d_slots = torch.from_numpy(slots).cuda()
                d_b2 = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
                sbv.verify_batch_keyed_dev(d_rsh.data_ptr(), d_slots.data_ptr(), n, d_b2.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                sbv.profile_enable(True)
                tk = time.perf_counter()
                for _ in range(args.steps):
                    sbv.verify_batch_keyed_dev(d_rsh.data_ptr(), d_slots.data_ptr(), n, d_b2.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                tk = time.perf_counter() - tk
                kp, kv, kl = sbv.profile_read()
                sbv.profile_enable(False)
                keyed = {"value": n * args.steps / tk, "unit": "verifies/s", "distinct_keys": int(len(keys)),
                         "kernel_us": {"k_p256_prep_keyed": kp / max(1, kl), "k_p256_verify_keyed": kv / max(1, kl)},
                         "bitmap_correct": bool((d_b2.cpu().numpy() == valid).all()),
                         "note": "sbv_p256_verify_batch_keyed_dev: r|s|hash + key slot per signature (100.125 B algorithmic), "
                                 "per-key comb tables resident in HBM; not the headline path"}
                sbv.clear_keys()
        except Exception as e:  # the headline number must not depend on the secondary leg
            keyed = {"error": repr(e)}
    extra = {}
    if world == 1 and not args.primary_only:
        for name, fn in (("warm_key_cache", lambda: leg_warm_cache(sbv, torch, d_tuples, d_bitmap, valid, n, max(2, args.steps // 2), stream)),
                         ("all_valid", lambda: leg_all_valid(sbv, synth, torch, n, max(2, args.steps // 2), stream)),
                         ("key_count_sweep", lambda: leg_key_count_sweep(sbv, synth, torch, n, stream)),
                         ("end_to_end", lambda: leg_end_to_end(sbv, tuples, valid, n, max(2, args.steps // 2))),
                         ("sharded_entry", lambda: leg_sharded(sbv, tuples, valid, n, max(2, args.steps // 2))),
                         ("ed25519", lambda: leg_ed25519(sbv, torch, n, max(2, args.steps // 2), stream, not args.no_cpu_baseline,
                                                         os.environ.get("SBV_BENCH_ED_HOT", "1") != "0")),      # 0: PMC passes (the hot leg's near-empty Q launches would dilute the per-launch counters)
                         ("secp256k1", lambda: leg_secp256k1(sbv, torch, n, max(2, args.steps // 2), stream, not args.no_cpu_baseline)),
                         ("projected_strong_scaling", lambda: leg_projected_strong_scaling(sbv, torch, d_tuples, valid, n, stream, 1e3 * elapsed / args.steps)),
                         ("m2_commit_quorum_us", lambda: leg_m2(tuples, n)),
                         ("verify_proposal_k10000_us", leg_proposals),
                         ("replay_550k", lambda: leg_replay_550k(sbv, synth)),
                         ("replay_550k_keyed", lambda: leg_replay_550k_keyed(sbv, synth)),
                         ("consenter_keys_550k", lambda: leg_consenter_keys(sbv, synth, torch, stream, max(2, args.steps // 2))),
                         ("front_end_msgs_per_s", lambda: leg_front_end(sbv, tuples, valid, n))):
            if args.legs and name not in args.legs.split(","):
                continue
            try:
                extra[name] = fn()
            except Exception as e:      # noqa: BLE001 - the headline number must not depend on a secondary leg
                extra[name] = {"error": repr(e)}
    if world == 1 and args.primary_only and args.warm_leg:
        extra["warm_key_cache"] = leg_warm_cache(sbv, torch, d_tuples, d_bitmap, valid, n, max(2, args.steps // 2), stream)
    if world > 1:
        flag = torch.tensor([1 if ok else 0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
        # the gathered bitmap must hold this rank's shard at its slot
        ok = ok and bool((d_all[rank * nbytes:(rank + 1) * nbytes].cpu().numpy() == valid).all())

    if rank == 0:
        total = n * world * args.steps
        value = total / elapsed
        # dominant kernel.  Grouped batch: k_verify_keyed_q, the key-comb additions over the grouped tuples; it is
        # launched once per chunk of windows, and one launch executes 32.22/chunks of the 45.22 comb additions (13 for
        # u1*G in k_gphase_generic + 32.22 for u2*Q) that make up a grouped tuple's stage B — it is charged that share
        # of the tuple's 160.125 algorithmic bytes.  Ungrouped batch: k_p256_verify, all of stage B, all n tuples.
        dom_launches_per_step = max(1, int(round(dominant_launches / max(1, dom_steps * ((n + (1 << 21) - 1) >> 21)))))
        kern_s = (dominant_us / max(1, dominant_launches)) * 1e-6
        if was_grouped:
            dom_name, dom_units = "k_verify_keyed_q", n_grouped
            share = (Q_ADDS_PER_TUPLE / dom_launches_per_step) / (G_ADDS_PER_TUPLE + Q_ADDS_PER_TUPLE)
        else:
            dom_name, dom_units, share = "k_p256_verify", n, 1.0
        achieved = ALGO_BYTES_PER_VERIFY * dom_units * share / kern_s / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")      # per-launch HBM bytes from a rocprofv3 --pmc run
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dom_name + "_hbm_bytes_per_launch")
                # NOT measured in this run: counters need their own rocprofv3 --pmc passes (tools/gpu_session.sh pmc); the figure is the
                # committed summary of the builder session named here (say so in the line)
                traffic_source = "profiles/traffic.json - " + str(tj.get("source", "rocprofv3 --pmc passes of a builder session"))
            except Exception:
                traffic = None
        line = {
            "metric": "ECDSA P-256 verifies/sec at batch=1M", "value": value, "unit": "verifies/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "configs[1]: standalone kernel, 2^20 synthetic P-256 (r,s,hash,pk) tuples per GPU, "
                                   "accept-bitmap out; 1024 keys, 7/8 valid + 1/8 single-bit-corrupted",
                       "tuples_per_gpu": n, "global_batch": n * world,
                       "parallelism": "shard-by-tuple" + (f" x{world} + RCCL all-gather of bitmaps" if world > 1 else "")},
            "bitmap_correct": ok, "init_s": init_s,
            "per_rank_ms_per_step": per_rank_ms, "per_rank_dominant_kernel_us": per_rank_dom_us,
            "all_gather_ms": gather_ms if world > 1 else None,
            "self_launched": os.environ.get("SBV_BENCH_SELF_LAUNCHED") == "1",
            "kernel_us": {"k_p256_prep": prep_us, "stage_b_all_kernels": verify_us,
                          dom_name: dominant_us / max(1, dominant_launches), dom_name + "_launches_per_step": dom_launches_per_step,
                          "launches": launches},
            "key_grouping": {"enabled": was_grouped, "groups": groups, "tuples_registered_key_kernel": n_grouped,
                             "tuples_generic_kernel": n_ungrouped, "tuples_rejected_for_their_key": n_key_rejected,
                             "key_sorted_list": os.environ.get("SBV_GROUP_SORT", "1") != "0",
                             "note": "in-step grouping by public key and counting sort of the grouped tuples by key (consensus_amd/csrc/p256_group.h); all of it is inside the timed region"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": dom_name, "units_per_launch": dom_units, "share_of_a_tuples_stage_b_per_launch": share,
                         "note": "algorithmic bytes = 160.125 B/verify x tuples that launch processed x the share of their "
                                 "stage B it executes / its avg duration (HIP events on the launch stream); the path is "
                                 "integer-ALU bound, see DESIGN.md"},
        }
        # share of the integer-multiply pipe's measured peak that the dominant kernel sustains
        if was_grouped:
            # 32 windows for every tuple + the carry window in the wavefronts that need it: 1 - (1 - 0.0039)^64 = 22 % for
            # uniform scalars (p256_comb29.h: qphase29_point)
            adds_per_launch = Q_ADDS_PER_TUPLE / dom_launches_per_step
            mads_per_s = dom_units * adds_per_launch * MADS_PER_MIXED_ADD / kern_s
            line["int_mul_issue_fraction"] = {"value": mads_per_s / PEAK_LANE_MADS_PER_S, "lane_mads_per_s": mads_per_s,
                                              "peak_lane_mads_per_s": PEAK_LANE_MADS_PER_S, "kernel": dom_name,
                                              "mads_per_mixed_addition": MADS_PER_MIXED_ADD,
                                              "note": "v_mad_i64_i32 / v_mad_u64_u32 issued per second by the dominant kernel / the microbenchmarked peak"}
        line.update(extra)
        if ungrouped is not None:
            line["without_key_grouping"] = ungrouped
        if keyed is not None:
            line["registered_key_path"] = keyed
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(tuples, n, got)
        print(json.dumps(line), flush=True)
        if not ok:
            sys.exit(3)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
