#!/bin/bash
# usage (on the GPU box): tools/pmc_ffn.sh <tag> [expert|dense]    SQ / LDS / TA counters of the fused FFN kernel,
# one rocprofv3 --pmc pass per counter group (no trace domains besides --kernel-trace)
set -e
tag=$1; mode=${2:-expert}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
            "SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES" \
            "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  d=$R/gpurun_out/pmcf_${tag}_$i
  mkdir -p $d
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $ctrs -d $d -o run -- python $R/tools/ffn_bench.py --only $mode --rounds 1 --iters 8 > $d/log.txt 2>&1 || echo "pass $i failed"
  python $R/tools/pmc_dump.py $(find $d -name "*.db" | sort | tail -1) 2>&1 | grep -E "ffn_fwd|gemm_nt_dma" || true
done
