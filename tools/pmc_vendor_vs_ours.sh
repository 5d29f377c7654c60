#!/bin/bash
# Round 4: the vendor GEMM (hipBLASLt through torch.nn.functional.linear -- CALIBRATION ONLY, never on the product
# path) and the hand-written 8-phase kernel under the SAME rocprofv3 counter passes, on the same shapes and the same operand fills.
#   usage: tools/pmc_vendor_vs_ours.sh <tag>     -> gpurun_out/pmcvo_<side>_<pass>_<tag>.csv   (summarise: tools/pmc_vendor_vs_ours.py <tag>)
# Each counter set in its own run, kernel-trace only (the guide's recipe; FETCH_SIZE and WRITE_SIZE cannot share a pass).
TAG=${1:-r4}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
PASSES=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
        "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
        "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_LOAD_BANDWIDTH GRBM_GUI_ACTIVE"
        "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
i=0
for pass in "${PASSES[@]}"; do
  for side in ours vendor; do
    if [ $side = ours ]; then CMD="$R/tools/bin/gemm_bench 1024 1 2"; export NT_SHAPES=6; else CMD="python $R/tools/vendor_calibration.py"; export ITERS=1 NO_ATTN=1; fi
    timeout 240 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmcvo_${TAG}_${side}_$i -o g --output-format csv -- $CMD > /tmp/pmcvo_${TAG}_${side}_$i.log 2>&1
    f=$(find /tmp/pmcvo_${TAG}_${side}_$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then cp $f $R/gpurun_out/pmcvo_${side}_p${i}_$TAG.csv; else tail -5 /tmp/pmcvo_${TAG}_${side}_$i.log > $R/gpurun_out/pmcvo_${side}_p${i}_$TAG.err; fi
  done
  i=$((i+1))
done
ls $R/gpurun_out | grep "pmcvo_.*_$TAG" | tr '\n' ' '
