cd $GRAFT_REPO_ROOT; O=gpurun_out/r06w; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ed25519.py -x -q -s -k hot_keys > $O/pytest_ed_hot.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ed_hot.log
tail -8 $O/pytest_ed_hot.log
python tools/ab_variants.py ed25519 > /dev/null 2>&1
AB_ARGS=ed25519 bash tools/ab_lib.sh consensus_amd/libsbv_base.so 3 tools/ab_variants.py > $O/ab_ed_hotbuild_vs_base.jsonl
cut -c1-600 $O/ab_ed_hotbuild_vs_base.jsonl
