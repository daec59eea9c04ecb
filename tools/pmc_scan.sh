#!/bin/bash
# Instruction-mix / LDS counters of the Mode-A scan (tools/bench_scan.py): gpurun -- 'bash tools/pmc_scan.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmca_1 gpurun_out/pmca_2 gpurun_out/pmca_3
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/pmca_1 -o p -- python tools/bench_scan.py --steps 1 > gpurun_out/pmca_1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d gpurun_out/pmca_2 -o p -- python tools/bench_scan.py --steps 1 > gpurun_out/pmca_2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_ANY -d gpurun_out/pmca_3 -o p -- python tools/bench_scan.py --steps 1 > gpurun_out/pmca_3.log 2>&1
for i in 1 2 3; do python tools/rocpd_pmc.py $(ls gpurun_out/pmca_$i/*.db | head -1) | grep -A8 "timeosc_w1000"; done
