export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
timeout -k 10 400 python tools/ab.py --reps 2 w4= "w1=env:MC_WPB_CLASSIFY=1" "w2=env:MC_WPB_CLASSIFY=2" "w8=env:MC_WPB_CLASSIFY=8" > gpurun_out/r3_ab17.log 2>&1; cat gpurun_out/r3_ab17.log
timeout -k 10 300 python tools/ab.py --reps 1 --bench-args "--grid-res|512" w4= "w1=env:MC_WPB_CLASSIFY=1" "w2=env:MC_WPB_CLASSIFY=2" >> gpurun_out/r3_ab17.log 2>&1; tail -3 gpurun_out/r3_ab17.log
timeout -k 10 300 python tools/ab.py --reps 1 --bench-args "--mode|isosweep|--steps|30" w4= "w1=env:MC_WPB_CLASSIFY=1" "w2=env:MC_WPB_CLASSIFY=2" >> gpurun_out/r3_ab17.log 2>&1; tail -3 gpurun_out/r3_ab17.log
timeout -k 10 300 python tools/ab.py --reps 1 --bench-args "--workload|torus" w4= "w1=env:MC_WPB_CLASSIFY=1" "w2=env:MC_WPB_CLASSIFY=2" >> gpurun_out/r3_ab17.log 2>&1; tail -3 gpurun_out/r3_ab17.log
