#!/bin/bash
# PMC passes over the attention kernels of the stand-alone bench (ONLY_ATTN=1), one counter set per run, kernel-trace only.
#   usage: tools/pmc_attn.sh <tag> [LD_LIBRARY_PATH for a variant]     -> gpurun_out/pmc_attn_<tag>.md
TAG=${1:-r2}; VAR=$2
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  LD_LIBRARY_PATH=$VAR ONLY_ATTN=1 timeout 120 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmca_${TAG}_$i -o g --output-format csv -- $R/tools/bin/gemm_bench 1024 2 2 > /tmp/pmca_${TAG}_$i.log 2>&1
  f=$(find /tmp/pmca_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmca_${i}_$TAG.csv || tail -3 /tmp/pmca_${TAG}_$i.log > $R/gpurun_out/pmca_${i}_$TAG.err
  i=$((i+1))
done
cd $R
python tools/pmc_attn_summary.py $TAG > gpurun_out/pmc_attn_$TAG.md; cat gpurun_out/pmc_attn_$TAG.md
