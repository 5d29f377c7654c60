#!/bin/bash
# Round 3 (late): staged four-wave GEMM with the hand-placed issue order (EZ_Q_INTERLEAVE 2): bit comparison + timing beside the 8-phase kernel.
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/gemm4q_r3ah.log; : > $OUT
echo "## correctness (v0 = 128x128 reference, v4 = four-wave; 10 launches)" >> $OUT
NT_SHAPES=9 timeout 120 tools/bin/gemm_bench 1024 10 0,4 2>&1 | grep "vit.qkv\|fc+qgelu\|ffn1+gelu\|patch\|train.fc" >> $OUT
for sc in 1 0; do
  echo "## OPERAND_SCALE=$sc, 600 launches per shape (v2 = 8-phase, v4 = four-wave)" >> $OUT
  OPERAND_SCALE=$sc NT_SHAPES=9 timeout 200 tools/bin/gemm_bench 1024 600 2,4 2>&1 | grep "vit.qkv\|fc+qgelu\|ffn1+gelu\|patch\|train.fc" >> $OUT
done
cat $OUT
