cd $GRAFT_REPO_ROOT; O=gpurun_out/r06an; mkdir -p $O
python tools/ab_step.py > /dev/null 2>&1
bash tools/ab_lib.sh consensus_amd/libsbv_chunk.so 3 tools/ab_step.py > $O/ab_p256_chunked_xcd.jsonl
cut -c1-330 $O/ab_p256_chunked_xcd.jsonl
for lib in default chunk default chunk; do
  if [ $lib = default ]; then python tools/key_sweep.py 20 2048,4096 2>&1 >/dev/null | grep -E "keys|hot" | sed "s/^/$lib /" >> $O/key_sweep_chunked_xcd.txt
  else SBV_LIB=consensus_amd/libsbv_chunk.so python tools/key_sweep.py 20 2048,4096 2>&1 >/dev/null | grep -E "keys|hot" | sed "s/^/$lib /" >> $O/key_sweep_chunked_xcd.txt; fi
done
cat $O/key_sweep_chunked_xcd.txt
