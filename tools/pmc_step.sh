#!/bin/bash
# Instruction-mix counters of every kernel of the bench step (serial order): gpurun -- 'bash tools/pmc_step.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcs_1 gpurun_out/pmcs_2
NMX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/pmcs_1 -o p -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-cold-start > gpurun_out/pmcs_1.log 2>&1
NMX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d gpurun_out/pmcs_2 -o p -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-cold-start > gpurun_out/pmcs_2.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/pmcs_1/*.db | head -1)
python tools/rocpd_pmc.py $(ls gpurun_out/pmcs_2/*.db | head -1)
rm -rf gpurun_out/pmcs_3
NMX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_ANY -d gpurun_out/pmcs_3 -o p -- python bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-cold-start --no-mode-a > gpurun_out/pmcs_3.log 2>&1
python tools/rocpd_pmc.py $(ls gpurun_out/pmcs_3/*.db | head -1)
