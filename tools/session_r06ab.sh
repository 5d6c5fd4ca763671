cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ab; mkdir -p $O
python tools/ab_variants.py ed25519 >/dev/null 2>&1
for r in 1 2 3; do
  for v in default one3 one4; do
    if [ $v = default ]; then python tools/ab_variants.py ed25519 2>/dev/null | tail -1 >> $O/ab_ed_one_$v.jsonl; else SBV_LIB=consensus_amd/libsbv_$v.so python tools/ab_variants.py ed25519 2>/dev/null | tail -1 >> $O/ab_ed_one_$v.jsonl; fi
  done
done
for f in $O/*.jsonl; do echo "== $f"; cut -c30-420 $f; done
