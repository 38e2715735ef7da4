export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
B="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_IFETCH SQ_INST_LEVEL_VMEM"
for w in sphere goursat torus; do
  case $w in sphere) BA="--in-flight|1";; goursat) BA="--mode|isosweep|--steps|10|--in-flight|1";; torus) BA="--workload|torus|--in-flight|1";; esac
  python tools/pmc.py gpurun_out/r3_stall_${w}_A.json --bench-args "$BA" $A
  python tools/pmc.py gpurun_out/r3_stall_${w}_B.json --bench-args "$BA" $B
  python tools/pmc.py gpurun_out/r3_stall_${w}_C.json --bench-args "$BA" $C
done
