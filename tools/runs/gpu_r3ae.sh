#!/bin/bash
# Round 3: the vendor library under the same two regimes as profiles/r3_gemm_zero_operands_power.log (random / all-zero operands, 1 500 launches).
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/vendor_power_r3ae.log; : > $OUT
for sc in 1 0; do
  echo "## OPERAND_SCALE=$sc (hipBLASLt through torch.nn.functional.linear, bias epilogue; 1500 launches per shape)" >> $OUT
  OPERAND_SCALE=$sc ITERS=1500 timeout 300 python tools/vendor_calibration.py 2>&1 | grep "torch linear" >> $OUT
  echo "## OPERAND_SCALE=$sc (hand-written 8-phase kernel with its fused epilogues; 1500 launches per shape)" >> $OUT
  OPERAND_SCALE=$sc NT_SHAPES=6 timeout 300 tools/bin/gemm_bench 1024 1500 2 2>&1 | grep -v "^batch" >> $OUT
done
cat $OUT
