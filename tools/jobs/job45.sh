export AB_IN_FLIGHT=3 AB_STEPS=90
echo "== torus 512 (3 in flight)"; bash tools/ab_commits.sh run --workload torus
echo "== rational 512 (3 in flight)"; bash tools/ab_commits.sh run --grid-res 512 --equation "x^2+y^2+z^2-1/(x^2+4)"
export AB_IN_FLIGHT=1 AB_STEPS=40
echo "== torus 512 (1 in flight)"; bash tools/ab_commits.sh run --workload torus
echo "== full gpu suite"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu45.log 2>&1; tail -3 gpurun_out/gpu45.log
