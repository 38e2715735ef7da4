echo "== torus 512"; bash tools/ab_commits.sh run --workload torus
echo "== sphere 512"; bash tools/ab_commits.sh run --grid-res 512
echo "== goursat isosweep"; bash tools/ab_commits.sh run --mode isosweep
echo "== sphere 1024"; bash tools/ab_commits.sh run
echo "== gyroid"; bash tools/ab_commits.sh run --workload gyroid
echo "== sphere 1024, 3 in flight"; AB_IN_FLIGHT=3 AB_STEPS=90 bash tools/ab_commits.sh run
echo "== torus, 3 in flight"; AB_IN_FLIGHT=3 AB_STEPS=90 bash tools/ab_commits.sh run --workload torus
echo "== quick parity"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/parity69.log 2>&1; tail -2 gpurun_out/parity69.log
