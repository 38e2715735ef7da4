timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu60.log 2>&1; tail -2 gpurun_out/gpu60.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke60.log 2>&1; tail -1 gpurun_out/smoke60.log
python bench.py > gpurun_out/bench60.json 2> gpurun_out/bench60.err; tail -c 1500 gpurun_out/bench60.json
