echo "== quick parity"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/parity54.log 2>&1; tail -2 gpurun_out/parity54.log
echo "== goursat isosweep"; bash tools/ab_commits.sh run --mode isosweep
echo "== sphere 1024"; bash tools/ab_commits.sh run
echo "== sphere 512"; bash tools/ab_commits.sh run --grid-res 512
echo "== torus 512"; bash tools/ab_commits.sh run --workload torus
echo "== gyroid"; bash tools/ab_commits.sh run --workload gyroid
