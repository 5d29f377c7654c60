#!/bin/bash
# PMC passes over the weight-gradient kernels of the ModifiedResNet-50 training step (tools/rn_train_profile.py, bf16, 256 images, 2 steps):
# LDS bank conflicts of the transpose reads, wait / issue shares, HBM-side bytes.  One counter set per run, kernel-trace only.
#   usage: tools/pmc_rn_wgrad.sh <tag>     -> gpurun_out/pmc_rn_wgrad_<tag>.md
TAG=${1:-r5}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
i=0
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  RN_PROFILE_STEPS=2 timeout 200 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmcrn_${TAG}_$i -o g --output-format csv -- python $R/tools/rn_train_profile.py > /tmp/pmcrn_${TAG}_$i.log 2>&1
  f=$(find /tmp/pmcrn_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep "rn_wgrad3x3\|rn_tn_skinny\|gemm_tn" $f) > $R/gpurun_out/pmca_${i}_rnw$TAG.csv || tail -3 /tmp/pmcrn_${TAG}_$i.log > $R/gpurun_out/pmca_${i}_rnw$TAG.err
  i=$((i+1))
done
cd $R
PMC_FILTER="rn_wgrad3x3|rn_tn_skinny|gemm_tn" python tools/pmc_attn_summary.py rnw$TAG > gpurun_out/pmc_rn_wgrad_$TAG.md; grep -c . gpurun_out/pmc_rn_wgrad_$TAG.md
