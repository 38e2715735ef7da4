export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
timeout -k 10 400 python tools/ab.py --reps 2 base= "m6=#define MC_CLASSIFY_MINW 6" "m7=#define MC_CLASSIFY_MINW 7" "m8=#define MC_CLASSIFY_MINW 8" > gpurun_out/r3_ab18.log 2>&1; cat gpurun_out/r3_ab18.log
timeout -k 10 300 python tools/ab.py --reps 1 --bench-args "--mode|isosweep|--steps|30" base= "m6=#define MC_CLASSIFY_MINW 6" "m7=#define MC_CLASSIFY_MINW 7" "m8=#define MC_CLASSIFY_MINW 8" >> gpurun_out/r3_ab18.log 2>&1; tail -4 gpurun_out/r3_ab18.log
timeout -k 10 300 python tools/ab.py --reps 1 --bench-args "--workload|torus" base= "m6=#define MC_CLASSIFY_MINW 6" "m7=#define MC_CLASSIFY_MINW 7" "m8=#define MC_CLASSIFY_MINW 8" >> gpurun_out/r3_ab18.log 2>&1; tail -4 gpurun_out/r3_ab18.log
timeout -k 10 300 python tools/ab.py --reps 1 --bench-args "--workload|gyroid|--steps|5" base= "m6=#define MC_CLASSIFY_MINW 6" "m8=#define MC_CLASSIFY_MINW 8" >> gpurun_out/r3_ab18.log 2>&1; tail -3 gpurun_out/r3_ab18.log
