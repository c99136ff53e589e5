#!/bin/bash
# usage: tools/pmc.sh <tag> <cmd...> ; collects three PMC passes into gpurun_out/pmc_<tag>_{a,b,c}
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_${tag}_a -o p -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${tag}_b -o p -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_${tag}_c -o p -- "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr --output-format csv -d $R/gpurun_out/pmc_${tag}_d -o p -- "$@" > /dev/null 2>&1
ls $R/gpurun_out/pmc_${tag}_*
