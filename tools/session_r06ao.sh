cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ao; mkdir -p $O
python tools/ab_step.py > /dev/null 2>&1
bash tools/ab_lib.sh consensus_amd/libsbv_s0.so 3 tools/ab_step.py > $O/ab_p256_sort_stride8.jsonl
cut -c1-330 $O/ab_p256_sort_stride8.jsonl
for lib in default s0 default s0; do
  if [ $lib = default ]; then python tools/key_sweep.py 20 2048,4096 2>&1 >/dev/null | grep -E "keys|hot" | sed "s/^/$lib /" >> $O/key_sweep_sort_stride8.txt
  else SBV_LIB=consensus_amd/libsbv_s0.so python tools/key_sweep.py 20 2048,4096 2>&1 >/dev/null | grep -E "keys|hot" | sed "s/^/$lib /" >> $O/key_sweep_sort_stride8.txt; fi
done
cat $O/key_sweep_sort_stride8.txt | cut -c1-210
python tools/ed_hot_sweep.py 20 1024 2>/dev/null | tee $O/ed_hot_sweep_stride8.jsonl
SBV_LIB=consensus_amd/libsbv_s0.so python tools/ed_hot_sweep.py 20 1024 2>/dev/null | tee $O/ed_hot_sweep_stride0.jsonl
