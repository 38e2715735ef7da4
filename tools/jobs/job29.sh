export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_trig.py -x -q 2>&1 | tail -3
bash tools/other_workloads.sh 2>&1 | grep -v amdgpu.ids | cut -c1-200 | head -12
