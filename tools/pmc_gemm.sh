#!/bin/bash
# PMC passes over the stand-alone GEMM bench (counters only: no trace domains besides kernel-trace).
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
V=${1:-2}
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
            "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$tag -o g --output-format csv -- $R/tools/bin/gemm_bench 1024 2 $V > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmc_${tag}_v$V.csv || tail -5 /tmp/pmc_$tag.log > $R/gpurun_out/pmc_${tag}_v$V.err
done
ls -la $R/gpurun_out | tail -8
