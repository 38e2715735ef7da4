set -x
export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3_t1.log 2>&1; tail -3 gpurun_out/r3_t1.log
python tools/ab.py --reps 2 old="#define MC_OLD_DIRECT 1" new= c160="#define MC_CCAP 160" c128="#define MC_CCAP 128" "w4=env:MC_WPB_EMIT=4" "w4c128=env:MC_WPB_EMIT=4;#define MC_CCAP 128" "w16c128=env:MC_WPB_EMIT=16;#define MC_CCAP 128" > gpurun_out/r3_ab1.log 2>&1
cat gpurun_out/r3_ab1.log
python tools/ab.py --reps 1 --bench-args "--workload|torus" th63= "th57=env:MC_TILE_H=57" "th32=env:MC_TILE_H=32" "th21=env:MC_TILE_H=21" "th16=env:MC_TILE_H=16" "th8=env:MC_TILE_H=8" > gpurun_out/r3_ab2.log 2>&1
cat gpurun_out/r3_ab2.log
python tools/ab.py --reps 1 --bench-args "--mode|isosweep|--steps|30" th63= "th57=env:MC_TILE_H=57" "th32=env:MC_TILE_H=32" "th21=env:MC_TILE_H=21" "th16=env:MC_TILE_H=16" "th8=env:MC_TILE_H=8" > gpurun_out/r3_ab3.log 2>&1
cat gpurun_out/r3_ab3.log
python tools/ab.py --reps 1 --bench-args "--grid-res|512" th63= "th32=env:MC_TILE_H=32" "th16=env:MC_TILE_H=16" > gpurun_out/r3_ab4.log 2>&1
cat gpurun_out/r3_ab4.log
