#!/bin/bash
# Static gfx950 ISA statistics of the hot kernels of the committed sources (no GPU needed: hipcc cross-compiles).
# usage: tools/isa_stats.sh > profiles/rNN/isa_stats_<tag>.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/consensus_amd/csrc"
TMP=$(mktemp -d)
for f in p256_group_kernels ed25519_group_kernels k256_kernels k256_group_kernels p256_kernels; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -S --cuda-device-only $f.hip -o $TMP/$f.s 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c $f.hip -o $TMP/$f.o -Rpass-analysis=kernel-resource-usage 2> $TMP/$f.res
done
python3 - "$TMP" <<'PY'
import collections, re, sys, os
tmp = sys.argv[1]
want = {"p256_group_kernels": ["k_verify_keyed_q", "k_gphase_generic", "k_gphase_sorted", "k_keytab29_chain", "k_keytab29_rows", "k_keytab29_fill", "k_keytab29_entries", "k_keytab29_fill_parts", "k_keytab29_fill_sym", "k_group_coop", "k_group_sort_count", "k_group_sort_scan",
                               "k_group_sort_scatter", "k_group_classify", "k_group_keycheck"],
        "ed25519_group_kernels": ["k_ed_qphase", "k_ed_gphase", "k_ed_finish"], "k256_kernels": ["k_k256_verify", "k_k256_prep"], "k256_group_kernels": ["k_k256_qphase", "k_k256_gphase_generic"],
        "p256_kernels": ["k_p256_prep", "k_p256_verify", "k_p256_verify_keyed", "k_p256_verify_keyed_coop", "k_p256_verify_prepared_small"]}

def report(k, m, txt, res):
    name = m.group(1)
    inst = re.search(r"I(L[ib]\d+E(?:L[ib]\d+E)*)E", name)
    label = k + ("<" + ", ".join(re.findall(r"L[ib](\d+)E", inst.group(1))) + ">" if inst else "")

    start = m.end()
    end = txt.index(".Lfunc_end", start)
    ins = []
    for l in txt[start:end].split("\n"):
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        ins.append(l.split()[0])
    c = collections.Counter(ins)
    mad = sum(v for a, v in c.items() if a.startswith("v_mad_i64") or a.startswith("v_mad_u64"))
    scr = sum(v for a, v in c.items() if a.startswith("scratch_"))
    r = re.search(r"Function Name: " + re.escape(name) + r".*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)", res, re.S)
    meta = "VGPRs %s, scratch %s B/lane, occupancy %s waves/SIMD, VGPR spills %s" % r.groups() if r else ""
    print(f"== {label}: {len(ins)} instructions, {mad} 64-bit multiply-accumulates ({100.0 * mad / max(1, len(ins)):.0f} %), {c.get('s_nop', 0)} s_nop, {scr} scratch accesses; {meta}")
    print("   top: " + ", ".join(f"{a} {v}" for a, v in c.most_common(8)))

print("# static ISA statistics of the hot kernels (hipcc --offload-arch=gfx950 -O3), sources as committed; tools/isa_stats.sh")
for f, kernels in want.items():
    txt = open(os.path.join(tmp, f + ".s")).read()
    res = open(os.path.join(tmp, f + ".res")).read()
    for k in kernels:
        ms = list(re.finditer(r"^(_ZN3sbv[L]?\d+" + k + r"(?:E|I)[^\n:]*):", txt, re.M))      # plain kernels and every template instantiation
        for m in ms:
            report(k, m, txt, res)
PY
rm -rf "$TMP"
