cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ac; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ed25519.py tests/test_gpu_edge_scale.py -x -q -k "ed25519 or ed_" > $O/pytest_ed.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ed.log
tail -4 $O/pytest_ed.log
for r in 1 2 3; do
  SBV_ED_UNGROUPED_QUAD=0 python tools/ab_variants.py ed25519 2>/dev/null | tail -1 >> $O/ab_ed_ungrouped_quad_off.jsonl
  python tools/ab_variants.py ed25519 2>/dev/null | tail -1 >> $O/ab_ed_ungrouped_quad_on.jsonl
done
for f in $O/*.jsonl; do echo "== $f"; cut -c30-420 $f; done
