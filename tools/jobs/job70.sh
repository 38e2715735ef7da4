echo "== torus 512"; bash tools/ab_commits.sh run --workload torus
echo "== goursat isosweep"; bash tools/ab_commits.sh run --mode isosweep
echo "== sphere 512"; bash tools/ab_commits.sh run --grid-res 512
echo "== sphere 1024"; bash tools/ab_commits.sh run
echo "== torus, 3 in flight"; AB_IN_FLIGHT=3 AB_STEPS=90 bash tools/ab_commits.sh run --workload torus
echo "== full gpu suite"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu70.log 2>&1; tail -2 gpurun_out/gpu70.log
