#!/usr/bin/env python3
"""profiles/traffic.json from the pmc step of tools/gpu_session.sh (pmc_summary.json: {kernel: {counter: {mean, dispatches}}}):
per-launch HBM bytes of every kernel = 2 x FETCH_SIZE KiB (gfx950 tallies 128-byte requests as 64, MI355X_MICROARCH.md) +
WRITE_SIZE KiB.  bench.py reads <kernel>_hbm_bytes_per_launch for roofline.traffic of the P-256 leg and (round 5) of the Ed25519 and
secp256k1 legs.  usage: traffic_from_pmc.py <pmc_summary.json> <tag> [note]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summary = json.load(open(sys.argv[1]))
tag = sys.argv[2]
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, counters only: no tracing in the same run) over "
              f"`python bench.py --steps 4 --warmup 2 --no-cpu-baseline --legs ed25519,secp256k1`, session {tag} (tools/gpu_session.sh pmc); "
              f"summary: profiles/{tag[:3]}/pmc_summary_{tag}.json",
    "method": "FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 tallies 128-B "
              "requests as 64 B); the guide calibrates that factor on wide coalesced reads only - the comb phases read 64-byte (96-byte: "
              "Ed25519) table entries as 16-byte loads per lane, so for them the doubled figure is an upper bound; WRITE_SIZE as is.  "
              "Per launch = mean over the dispatches of the profiled run.",
    "launch": "2^20 tuples",
}
# round 6: the chunks' launches of the cold step are two instantiations (<0, false> parks the accumulator, <0, true> compares);
# bench.py's `roofline` is about their average launch, as its HIP-event average is
for base, insts in (("k_verify_keyed_q", ("void k_verify_keyed_q<0, false>", "void k_verify_keyed_q<0, true>")),
                    ("k_ed_qphase", ("void k_ed_qphase<false>", "void k_ed_qphase<true>")),
                    ("k_k256_qphase", ("void k_k256_qphase<false>", "void k_k256_qphase<true>"))):
    full = [summary[k] for k in insts if k in summary]
    if len(full) == 2:
        summary[base] = {c: {"mean": sum(x[c]["mean"] for x in full) / 2, "dispatches": sum(x[c]["dispatches"] for x in full)}
                         for c in ("FETCH_SIZE", "WRITE_SIZE") if all(c in x for x in full)}
for k, c in sorted(summary.items()):
    f = c.get("FETCH_SIZE", {}).get("mean")
    w = c.get("WRITE_SIZE", {}).get("mean")
    if f is None or w is None:
        continue
    name = k.replace("void ", "")
    if name == "k_verify_keyed_q<0>":          # rounds 2-5: the one instantiation of the chunks' launches
        name = "k_verify_keyed_q"
    out[name + "_fetch_kib_raw"] = f
    out[name + "_write_kib_raw"] = w
    out[name + "_hbm_bytes_per_launch"] = int(round((2 * f + w) * 1024))
    out[name + "_dispatches"] = c.get("FETCH_SIZE", {}).get("dispatches")
if len(sys.argv) > 3:
    out["note"] = sys.argv[3]
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.endswith("_hbm_bytes_per_launch") and ("qphase" in k or "keyed_q" in k)}, indent=1))
