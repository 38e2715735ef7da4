cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/idxprof53 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/index_probe1.py > $GRAFT_REPO_ROOT/gpurun_out/idx53.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/idxprof53/run_kernel_stats.csv
