echo "== trig tests"; timeout -k 10 600 python -m pytest tests/test_trig.py tests/test_gpu_parity.py -m gpu -x -q -k "trig or gyroid or sin or cos" > gpurun_out/trig81.log 2>&1; tail -2 gpurun_out/trig81.log
echo "== gyroid 1024"; bash tools/ab_commits.sh run --workload gyroid
echo "== gyroid 1024, 3 in flight"; AB_IN_FLIGHT=3 AB_STEPS=20 bash tools/ab_commits.sh run --workload gyroid
echo "== gyroid 1/8 slab"; bash tools/ab_commits.sh run --workload gyroid --slab-of 8
