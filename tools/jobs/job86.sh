echo "== torus 512"; bash tools/ab_commits.sh run --workload torus
echo "== rational 512"; bash tools/ab_commits.sh run --grid-res 512 --equation "x^2+y^2+z^2-1/(x^2+4)"
echo "== torus 256"; bash tools/ab_commits.sh run --workload torus --grid-res 256
echo "== gyroid 1024"; bash tools/ab_commits.sh run --workload gyroid
echo "== gyroid 1/8 slab"; bash tools/ab_commits.sh run --workload gyroid --slab-of 8
echo "== parity"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_trig.py -m gpu -x -q > gpurun_out/parity86.log 2>&1; tail -2 gpurun_out/parity86.log
