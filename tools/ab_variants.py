#!/usr/bin/env python3
"""One JSON line: the Ed25519 (configs[4]) and secp256k1 grouped steps at 2^20, cold and warm, through bench.py's own legs (SBV_LIB selects
the library build; tools/ab_lib.sh alternates two builds in fresh processes).  usage: ab_variants.py [ed25519,secp256k1]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import bench
    import consensus_amd as sbv
    which = (sys.argv[1] if len(sys.argv) > 1 else "ed25519,secp256k1").split(",")
    sbv.init(0)
    stream = torch.cuda.current_stream()
    out = {"lib": os.path.basename(sbv.LIB_PATH)}
    n = 1 << 20
    for name, fn in (("ed25519", bench.leg_ed25519), ("secp256k1", bench.leg_secp256k1)):
        if name not in which:
            continue
        r = fn(sbv, torch, n, 10, stream, False, os.environ.get("AB_NO_HOT") is None) if name == "ed25519" else fn(sbv, torch, n, 10, stream, False)
        keep = {k: r[k] for k in ("ms_per_step", "bitmap_correct") if k in r}
        for k, v in r.items():
            if isinstance(v, dict) and "ms_per_step" in v:
                keep[k] = {"ms_per_step": v["ms_per_step"], "bitmap_correct": v.get("bitmap_correct")}
        if "roofline" in r:
            keep["dominant_kernel_us"] = r["roofline"].get("avg_launch_us")
        out[name] = keep
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
