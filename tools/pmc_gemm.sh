#!/bin/bash
# PMC passes over the stand-alone GEMM bench (each counter set in its own run, kernel-trace only: the guide's recipe).
#   usage: tools/pmc_gemm.sh <tag>      -> gpurun_out/pmc_<set>_<tag>.csv
TAG=${1:-r1}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 90 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_${TAG}_$i -o g --output-format csv -- $R/tools/bin/gemm_bench 1024 1 2 > /tmp/pmc_${TAG}_$i.log 2>&1
  f=$(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmc_${name}_$TAG.csv || tail -3 /tmp/pmc_${TAG}_$i.log > $R/gpurun_out/pmc_${name}_$TAG.err
  i=$((i+1))
done
ls $R/gpurun_out | grep "_$TAG" | tr '\n' ' '
