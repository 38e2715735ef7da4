export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/seedprof3 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/seed_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r3_seed.log 2>&1
tail -6 $GRAFT_REPO_ROOT/gpurun_out/r3_seed.log
