#!/usr/bin/env python3
"""Randomised campaign over the CPU emulation of the device algorithms added in round 4 (tests/emul, built with the field contract
checks on: an operand outside its bounds aborts the process).  No GPU needed.

  wide     p256_widetab29.h: the device-side builder of a wide comb against the host builder, byte for byte, random keys, widths 10-13
  keyed    the registered-key lanes with every slot widened (one-lane form, 8-lane form, prepared one-launch form) against the oracle
           on seeded batches with corrupted tuples
  ed       the Ed25519 grouped step (key check over the ungrouped candidates, batched finish) against the oracle on seeded batches
  p256g    the P-256 grouped step (sorted / compaction order, 1-4 chunks, the one-launch latency form, key-table cache on / off; round 5: table classes — rows only / full / upgrade)
  k256g    the secp256k1 grouped step (1-3 chunks, stage-A chunking, its key-table cache on / off)
  one      the one-lane kernels of the three schemes (all-distinct keys / small batches: 256 doublings per signature)

usage: fuzz_emul.py <minutes> [workers]      one JSON line per worker at the end; exit status 1 on any mismatch."""
import ctypes, json, multiprocessing as mp, os, random, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def worker(args):
    wid, minutes = args
    import p256_py as ec
    emul = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "libsbv_emul.so"))
    oracle = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsbv_oracle.so"))
    emul.sbve_widetab_build_mismatches.restype = ctypes.c_size_t
    emul.sbve_widetab_build_mismatches.argtypes = [ctypes.c_char_p, ctypes.c_int]
    emul.sbve_p256_verify_batch_keyed.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    emul.sbve_set_keyed_wide.argtypes = [ctypes.c_int, ctypes.c_uint]
    emul.sbve_ed25519_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    gen_args = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    oracle.sbvo_gen_batch.argtypes = gen_args
    oracle.sbvo_ed25519_gen_batch.argtypes = gen_args
    emul.sbve_p256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    emul.sbve_k256_verify_batch_grouped.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    emul.sbve_p256_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    emul.sbve_k256_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    emul.sbve_ed25519_verify_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    emul.sbve_key_cache.argtypes = [ctypes.c_int, ctypes.c_uint32]
    emul.sbve_set_full_table_min.argtypes = [ctypes.c_uint32]
    emul.sbve_hot_keys.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    emul.sbve_hot_stats.argtypes = [ctypes.c_void_p]
    emul.sbve_hot_comb_mismatches.argtypes = [ctypes.c_uint32]
    emul.sbve_hot_comb_mismatches.restype = ctypes.c_size_t
    emul.sbve_ed_chain_mismatches.restype = ctypes.c_ulong
    emul.sbve_ed_quad_mismatches.restype = ctypes.c_ulong
    emul.sbve_scheme_key_cache.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
    emul.sbve_ed_hot_keys.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    emul.sbve_ed_hot_stats.argtypes = [ctypes.c_void_p]
    emul.sbve_ed_hot_comb_mismatches.argtypes = [ctypes.c_uint32]
    emul.sbve_ed_hot_comb_mismatches.restype = ctypes.c_size_t
    oracle.sbvo_k256_gen_batch.argtypes = gen_args
    rng = random.Random(0xF022 + wid)
    out = {"worker": wid, "wide_keys": 0, "keyed_tuples": 0, "ed_tuples": 0, "p256g_tuples": 0, "k256g_tuples": 0, "one_tuples": 0, "mismatches": 0}
    t_end = time.time() + 60 * minutes
    it = 0
    while time.time() < t_end or it < 6:          # at least one iteration of every kind, however short the run
        it += 1
        kind = it % 6
        if kind == 0:
            q = ec.pt_mul(rng.randrange(1, ec.N), ec.G)
            kb = q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big")
            bits = rng.choice((10, 10, 11, 12, 13))
            bad = emul.sbve_widetab_build_mismatches(kb, bits)
            out["wide_keys"] += 1
            if bad != 0:
                out["mismatches"] += 1
                out.setdefault("first", ["wide", kb.hex(), bits, int(bad)])
        elif kind == 1:
            n, nkeys = rng.choice((200, 333, 517)), rng.choice((1, 2, 5))
            seed = rng.randrange(1 << 32)
            tup = ctypes.create_string_buffer(160 * n)
            exp = ctypes.create_string_buffer((n + 7) // 8)
            oracle.sbvo_gen_batch(seed, n, nkeys, rng.choice((2, 3, 5)), tup, exp, 1)
            keys, index, rsh, slots = [], {}, bytearray(), []
            raw = tup.raw
            for i in range(n):
                k = raw[160 * i + 96:160 * i + 160]
                if k not in index:
                    index[k] = len(keys); keys.append(k)
                rsh += raw[160 * i:160 * i + 96]; slots.append(index[k])
            arr = (ctypes.c_uint32 * n)(*slots)
            emul.sbve_set_keyed_wide(rng.choice((10, 11, 12)), len(keys))
            for form in (0, 1):
                emul.sbve_set_keyed_coop(form)
                bm = ctypes.create_string_buffer((n + 7) // 8)
                emul.sbve_p256_verify_batch_keyed(bytes(rsh), arr, n, b"".join(keys), len(keys), bm, 64, 4)
                out["keyed_tuples"] += n
                if bm.raw != exp.raw:
                    out["mismatches"] += 1
                    out.setdefault("first", ["keyed", seed, n, nkeys, form])
            # the prepared one-launch form (stage A of a call on the host half, 16 lanes per signature): calls of 1..32 records
            emul.sbve_set_keyed_coop(3)
            off, got = 0, []
            while off < n:
                m = min(rng.choice((1, 2, 15, 31, 32)), n - off)
                sub = (ctypes.c_uint32 * m)(*slots[off:off + m])
                bm = ctypes.create_string_buffer((m + 7) // 8)
                emul.sbve_p256_verify_batch_keyed(bytes(rsh[96 * off:96 * (off + m)]), sub, m, b"".join(keys), len(keys), bm, 64, 1)
                got += [(bm.raw[i >> 3] >> (i & 7)) & 1 for i in range(m)]
                off += m
            out["keyed_tuples"] += n
            if got != [(exp.raw[i >> 3] >> (i & 7)) & 1 for i in range(n)]:
                out["mismatches"] += 1
                out.setdefault("first", ["prepared", seed, n, nkeys])
            emul.sbve_set_keyed_coop(0)
            emul.sbve_set_keyed_wide(16, 0)
        elif kind == 5:
            which = (it // 6) % 3
            n = rng.choice((60, 130))
            seed = rng.randrange(1 << 32)
            width = 128 if which == 2 else 160
            tup = ctypes.create_string_buffer(width * n)
            exp = ctypes.create_string_buffer((n + 7) // 8)
            (oracle.sbvo_gen_batch, oracle.sbvo_k256_gen_batch, oracle.sbvo_ed25519_gen_batch)[which](seed, n, rng.choice((3, n)), rng.choice((2, 3, 5)), tup, exp, 1)
            bm = ctypes.create_string_buffer((n + 7) // 8)
            if which == 0:
                emul.sbve_p256_verify_batch(tup.raw, n, bm, 64, rng.choice((1, 4, 8)))
            elif which == 1:
                emul.sbve_k256_verify_batch(tup.raw, n, bm)
            else:
                emul.sbve_ed25519_verify_batch(tup.raw, n, bm)
            out["one_tuples"] += n
            if bm.raw != exp.raw:
                out["mismatches"] += 1
                out.setdefault("first", ["one", which, seed, n])
        elif kind == 3 or kind == 4:
            p256 = kind == 3
            n, nkeys = rng.choice((300, 640, 1000)), rng.choice((2, 5, 12))
            seed = rng.randrange(1 << 32)
            tup = ctypes.create_string_buffer(160 * n)
            exp = ctypes.create_string_buffer((n + 7) // 8)
            (oracle.sbvo_gen_batch if p256 else oracle.sbvo_k256_gen_batch)(seed, n, nkeys, rng.choice((2, 3, 7)), tup, exp, 1)
            cache = rng.random() < 0.5
            if p256:
                emul.sbve_set_group_sort(rng.choice((0, 1, 1)))
                emul.sbve_set_group_chunks(rng.choice((1, 2, 3, 4)))
                emul.sbve_set_group_coop(rng.choice((0, 0, 1)))
                emul.sbve_key_cache(1 if cache else 0, 64)
                # round 5, hot keys: a small pool of 16-bit combs (each costs the emulator about a second), promotion after the first or
                # the second pass; the third pass verifies the promoted keys' tuples in the wide pass
                hot = cache and rng.random() < 0.35
                emul.sbve_hot_keys(rng.choice((1, 2, 3)) if hot else 0, rng.choice((40, 150, 400)))
            else:
                hot = False
                emul.sbve_set_k256_prep_t(rng.choice((1, 4, 8)))
                emul.sbve_scheme_key_cache(1, 1 if cache else 0, 64)
            for rep in range((3 if hot else 2) if cache else 1):      # with the cache: a cold pass, then a warm one over the same keys
                bm = ctypes.create_string_buffer((n + 7) // 8)
                if p256:
                    # round 5, table classes: rows only / full tables / a mix, and through the cache an upgrade (rows-only slot, then hot)
                    emul.sbve_set_full_table_min(rng.choice((0, 16, 64, 200, 10**6)))
                    emul.sbve_p256_verify_batch_grouped(tup.raw, n, bm, rng.choice((1, 8, 30)), rng.choice((4, 64)), 12, None)
                else:
                    emul.sbve_k256_verify_batch_grouped(tup.raw, n, bm, rng.choice((1, 8, 30)), rng.choice((4, 64)), 12, rng.choice((1, 2, 3)), None)
                out["p256g_tuples" if p256 else "k256g_tuples"] += n
                if bm.raw != exp.raw:
                    out["mismatches"] += 1
                    out.setdefault("first", ["p256g" if p256 else "k256g", seed, n, nkeys, cache, rep])
            if hot:
                hs = (ctypes.c_uint32 * 4)()
                emul.sbve_hot_stats(hs)
                out["hot_promotions"] = out.get("hot_promotions", 0) + hs[0]
                out["hot_wide_tuples"] = out.get("hot_wide_tuples", 0) + hs[2]
                if hs[0] and emul.sbve_hot_comb_mismatches(hs[0] - 1) != 0:
                    out["mismatches"] += 1
                    out.setdefault("first", ["hot comb", seed, n, nkeys])
                emul.sbve_hot_keys(0, 4096)
            if p256:
                emul.sbve_key_cache(0, 64); emul.sbve_set_group_sort(1); emul.sbve_set_group_chunks(3); emul.sbve_set_group_coop(0)
            else:
                emul.sbve_scheme_key_cache(1, 0, 64)
        else:
            n, nkeys = rng.choice((300, 700, 1100)), rng.choice((3, 7, 20))
            seed = rng.randrange(1 << 32)
            tup = ctypes.create_string_buffer(128 * n)
            exp = ctypes.create_string_buffer((n + 7) // 8)
            oracle.sbvo_ed25519_gen_batch(seed, n, nkeys, rng.choice((2, 3, 5)), tup, exp, 1)
            bm = ctypes.create_string_buffer((n + 7) // 8)
            # round 6, this scheme's cache and hot keys: a small pool of 16-bit combs of -A (each costs the emulator about a second), promoted by
            # the first or the second pass; later passes serve the all-hot wavefronts from them
            ed_cache = rng.random() < 0.5
            ed_hot = ed_cache and rng.random() < 0.5
            if ed_cache:
                emul.sbve_scheme_key_cache(2, 1, rng.choice((4, 16, 64)))
                emul.sbve_ed_hot_keys(rng.choice((1, 2, 3)) if ed_hot else 0, rng.choice((40, 150, 400)))
            for rep in range((3 if ed_hot else 2) if ed_cache else 1):
                bm = ctypes.create_string_buffer((n + 7) // 8)
                emul.sbve_ed25519_verify_batch_grouped(tup.raw, n, bm, rng.choice((1, 8, 40)), 64, 12, rng.choice((1, 2, 3)), rng.choice((2, 4, 8)), None)
                out["ed_tuples"] += n
                if bm.raw != exp.raw:
                    out["mismatches"] += 1
                    out.setdefault("first", ["ed cached" if ed_cache else "ed", seed, n, nkeys, ed_hot, rep])
            if ed_hot:
                hs = (ctypes.c_uint32 * 6)()
                emul.sbve_ed_hot_stats(hs)
                out["ed_hot_promotions"] = out.get("ed_hot_promotions", 0) + hs[0]
                out["ed_hot_wide_tuples"] = out.get("ed_hot_wide_tuples", 0) + hs[2]
                if hs[0] and emul.sbve_ed_hot_comb_mismatches(hs[0] - 1) != 0:
                    out["mismatches"] += 1
                    out.setdefault("first", ["ed hot comb", seed, n, nkeys])
            if ed_cache:
                emul.sbve_ed_hot_keys(0, 4096)
                emul.sbve_scheme_key_cache(2, 0, 0)
            if emul.sbve_ed_quad_mismatches() != 0:                   # round 6: the quad form of the one-lane kernel against the one-lane kernel, every ungrouped tuple
                out["mismatches"] += 1
                out.setdefault("first", ["ed quad one-lane", seed, n, nkeys])
            if emul.sbve_ed_chain_mismatches() != 0:                  # round 5: the quad-lane base chain against the one-lane chain, every cold key
                out["mismatches"] += 1
                out.setdefault("first", ["ed quad chain", seed, n, nkeys])
            if bm.raw != exp.raw:
                out["mismatches"] += 1
                out.setdefault("first", ["ed", seed, n, nkeys])
    # the emulator's own cross-checks: every lane of a multi-lane group ended with the same point; the host half of the prepared
    # form computed the scalars the stage-A kernel computes
    emul.sbve_coop_disagreements.restype = ctypes.c_ulong
    emul.sbve_small_disagreements.restype = ctypes.c_ulong
    emul.sbve_group_sort_violations.restype = ctypes.c_ulong
    out["lane_disagreements"] = int(emul.sbve_coop_disagreements()) + int(emul.sbve_small_disagreements()) + int(emul.sbve_group_sort_violations())
    out["mismatches"] += out["lane_disagreements"]
    return out


if __name__ == "__main__":
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    with mp.Pool(workers) as pool:
        res = pool.map(worker, [(w, minutes) for w in range(workers)])
    for r in res:
        print(json.dumps(r))
    total = {k: sum(r[k] for r in res) for k in ("wide_keys", "keyed_tuples", "ed_tuples", "p256g_tuples", "k256g_tuples", "one_tuples", "mismatches")}
    print(json.dumps({"total": total, "minutes": minutes, "workers": workers}))
    sys.exit(1 if total["mismatches"] else 0)
