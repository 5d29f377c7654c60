#!/bin/bash
# Round 3 (late): first run of the staged four-wave GEMM (gemm4q.hip, variant 4): bit comparison with the 128x128 kernel and timing
# beside the 8-phase kernel (variant 2), random and all-zero operands.
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/gemm4q_r3af.log; : > $OUT
echo "## correctness (v0 = 128x128 reference, v2 = 8-phase, v4 = four-wave; 20 launches each)" >> $OUT
NT_SHAPES=9 timeout 120 tools/bin/gemm_bench 1024 20 0,2,4 2>&1 | grep -v "^batch" >> $OUT
for sc in 1 0; do
  echo "## OPERAND_SCALE=$sc, 600 launches per shape" >> $OUT
  OPERAND_SCALE=$sc NT_SHAPES=9 timeout 200 tools/bin/gemm_bench 1024 600 2,4 2>&1 | grep "vit.qkv\|fc+qgelu\|ffn1+gelu\|patch\|train.fc" >> $OUT
done
cat $OUT
