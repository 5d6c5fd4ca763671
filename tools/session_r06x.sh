cd $GRAFT_REPO_ROOT; O=gpurun_out/r06x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o edhot -- python $GRAFT_REPO_ROOT/tools/ab_variants.py ed25519 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-200
find $O/prof -name "*kernel_trace.csv" -size +60M -delete
