cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ae; mkdir -p $O
python tools/ab_variants.py ed25519 >/dev/null 2>&1
for r in 1 2 3 4; do
  SBV_ED_UNGROUPED_KEYS=0 python tools/ab_variants.py ed25519 2>/dev/null | tail -1 >> $O/ab_ed_ungrouped_keys_off.jsonl
  python tools/ab_variants.py ed25519 2>/dev/null | tail -1 >> $O/ab_ed_ungrouped_keys_on.jsonl
done
for f in $O/*.jsonl; do echo "== $f"; cut -c30-420 $f; done
