echo "== sphere 1024"; bash tools/ab_commits.sh run
echo "== sphere 512"; bash tools/ab_commits.sh run --grid-res 512
echo "== parity subset"; python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/parity34.log 2>&1; tail -2 gpurun_out/parity34.log
