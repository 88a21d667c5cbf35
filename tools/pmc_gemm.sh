#!/bin/bash
# SQ wait / busy counters of the block GEMMs alone (tools/bench_gemm.py), one rocprofv3 --pmc pass per counter group.
# usage (on the GPU box, from the repo root): bash tools/pmc_gemm.sh gpurun_out/pmc_gemm
set -e
OUT=$(realpath -m ${1:-gpurun_out/pmc_gemm}); ROOT=$(pwd)
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_IFETCH" \
           "TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/tools/bench_gemm.py > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 $ROOT/tools/pmc_table.py $OUT k_gemm_pipe > $OUT/table.txt
