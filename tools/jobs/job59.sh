B="--in-flight|1"
python tools/pmc.py gpurun_out/pmc59_a.json --bench-args "$B" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA > gpurun_out/pmc59_a.log 2>&1
python tools/pmc.py gpurun_out/pmc59_b.json --bench-args "$B" SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS > gpurun_out/pmc59_b.log 2>&1
python tools/pmc.py gpurun_out/pmc59_c.json --bench-args "$B" SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_INSTS_SMEM > gpurun_out/pmc59_c.log 2>&1
python tools/pmc.py gpurun_out/pmc59_d.json --bench-args "$B" SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_CYCLES > gpurun_out/pmc59_d.log 2>&1
tail -3 gpurun_out/pmc59_*.log
