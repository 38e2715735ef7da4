export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/idxprof3 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/index_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r3_idx.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r3_idx.log | grep -v "^W\|^E\|rocprof" | tail -8
find $GRAFT_REPO_ROOT/gpurun_out/idxprof3 -name "*kernel_stats.csv" | head -1 | xargs cat
