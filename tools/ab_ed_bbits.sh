#!/bin/bash
# A/B of the Ed25519 grouped step's comb of B: fresh process per variant, two passes, cold and warm
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r05l; mkdir -p $OUT
python tools/bench_ed25519.py > /dev/null 2>&1   # batch cache in /tmp
for rep in 1 2; do for bits in 16 20 18; do
  echo "# bits=$bits cold rep $rep" >> $OUT/ab_ed_bbits.jsonl
  SBV_ED_B_BITS=$bits SBV_ED_COLD=1 SBV_BENCH_PRIMARY_ONLY=1 timeout 300 python tools/bench_ed25519.py >> $OUT/ab_ed_bbits.jsonl 2>> $OUT/ab_ed_bbits.err
  echo "# bits=$bits warm rep $rep" >> $OUT/ab_ed_bbits.jsonl
  SBV_ED_B_BITS=$bits SBV_BENCH_PRIMARY_ONLY=1 timeout 300 python tools/bench_ed25519.py >> $OUT/ab_ed_bbits.jsonl 2>> $OUT/ab_ed_bbits.err
done; done
cut -c1-260 $OUT/ab_ed_bbits.jsonl
