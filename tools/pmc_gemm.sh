#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> <shape>   (env selects the kernel variant)
set -e
tag=$1; shape=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
            "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
            "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" \
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  d=$R/gpurun_out/pmcg_${tag}_$i
  mkdir -p $d
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $ctrs -d $d -o run -- python $R/tools/gemm_stream_bench.py --only $shape --iters 12 > $d/log.txt 2>&1 || echo "pass $i failed"
  python $R/tools/pmc_dump.py $(find $d -name "*.db" | sort | tail -1) 2>&1 | grep gemm || true
done
