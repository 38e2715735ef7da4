python bench.py --no-cpu-baseline --steps 100 --warmup 10 > /dev/null 2>&1
python tools/profile_round.py --tag r03 > gpurun_out/profile_round.log 2>&1; tail -1 gpurun_out/profile_round.log | cut -c1-200
bash tools/other_workloads.sh > gpurun_out/other_workloads.txt 2>&1
bash tools/slab_probe.sh > gpurun_out/slab_probe.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke77.log 2>&1; tail -1 gpurun_out/smoke77.log
