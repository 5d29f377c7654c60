#!/bin/bash
# PMC passes over the attention kernels as the TRAINING STEP launches them (bench.py, 1024 pairs, 2 + 1 steps): forward (full-line
# stores), the fused backward at 197 tokens, the score-tile-once backward at <= 128 tokens (BERT).  One counter set per run, kernel-trace
# only (the guide's recipe).   usage: tools/pmc_attn_train.sh <tag>     -> gpurun_out/pmc_attn_<tag>.md
TAG=${1:-r5}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  EZCLIP_NO_CANARY=1 EZCLIP_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmcat_${TAG}_$i -o g --output-format csv -- python $R/bench.py --workload bf16_b1024_train --steps 1 --warmup 1 --no-also --no-cpu-baseline --no-recall --sustained-steps 0 > /tmp/pmcat_${TAG}_$i.log 2>&1
  f=$(find /tmp/pmcat_${TAG}_$i -name "*counter_collection.csv" | head -1)
  # keep the attention kernels' rows only (the whole step is ~100 MB of CSV)
  [ -n "$f" ] && (head -1 $f; grep "attn_" $f) > $R/gpurun_out/pmca_${i}_$TAG.csv || tail -3 /tmp/pmcat_${TAG}_$i.log > $R/gpurun_out/pmca_${i}_$TAG.err
  i=$((i+1))
done
cd $R
python tools/pmc_attn_summary.py $TAG > gpurun_out/pmc_attn_$TAG.md; grep -c . gpurun_out/pmc_attn_$TAG.md
