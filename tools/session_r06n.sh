cd $GRAFT_REPO_ROOT; O=gpurun_out/r06n; mkdir -p $O
python tools/ab_step.py > /dev/null 2>&1
for r in 1 2; do
  python tools/ab_step.py 2>/dev/null | tail -1 >> $O/ab_step_default.jsonl
  SBV_GROUP_CHUNKS=1 python tools/ab_step.py 2>/dev/null | tail -1 >> $O/ab_step_chunks1.jsonl
  python tools/ab_variants.py ed25519 2>/dev/null | tail -1 >> $O/ab_ed_pitch128.jsonl
  SBV_ED_B_PITCH=96 python tools/ab_variants.py ed25519 2>/dev/null | tail -1 >> $O/ab_ed_pitch96.jsonl
done
python tools/replay_pieces.py 262144,1048576 1 > $O/replay_pieces.jsonl 2>/dev/null
for f in $O/*.jsonl; do echo "== $f"; cat $f | cut -c1-420; done
