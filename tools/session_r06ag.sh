cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ag; mkdir -p $O
python tools/ab_variants.py ed25519 >/dev/null 2>&1
AB_ARGS=ed25519 bash tools/ab_lib.sh consensus_amd/libsbv_base.so 4 tools/ab_variants.py > $O/ab_ed_dbl_no_t.jsonl
bash tools/ab_lib.sh consensus_amd/libsbv_base.so 2 tools/ed_one_lane.py > $O/ab_ed_one_lane_dbl_no_t.jsonl
cut -c1-470 $O/ab_ed_dbl_no_t.jsonl; cat $O/ab_ed_one_lane_dbl_no_t.jsonl
