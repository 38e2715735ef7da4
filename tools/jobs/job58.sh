cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -o "SQ_[A-Z0-9_]*" gpurun_out/counters_list.txt | sort -u | tr '\n' ' ' | cut -c1-6000
