python bench.py --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2>&1   # (bring the box into its sustained state first)
python tools/profile_round.py --tag r03 > gpurun_out/profile_round.log 2>&1; tail -1 gpurun_out/profile_round.log | cut -c1-200
bash tools/other_workloads.sh > gpurun_out/other_workloads.txt 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu67.log 2>&1; tail -2 gpurun_out/gpu67.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke67.log 2>&1; tail -1 gpurun_out/smoke67.log
