cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_IFETCH SQ_INSTS" \
           "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcb_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcb_$i -o p -- python tools/run_bank_only.py > gpurun_out/pmcb_$i.log 2>&1
  python tools/rocpd_pmc.py $(ls gpurun_out/pmcb_$i/*.db | head -1) | grep -A8 "bank_w64c"
done
