#!/bin/bash
# Round 3: how much of the sustained GEMM rate is the power cap?  The same kernel, same shapes, 400 launches back to back, with random
# operands and with all-zero operands (no datapath toggling), rocm-smi clock / power sampled while each runs.
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/gemm_power_r3u.log
: > $OUT
for sc in 1 0; do
  echo "## OPERAND_SCALE=$sc" >> $OUT
  OPERAND_SCALE=$sc NT_SHAPES=4 timeout 300 tools/bin/gemm_bench 1024 1500 2 > gpurun_out/gb_scale$sc.log 2>&1 &
  BP=$!
  sleep 2
  for i in 1 2 3 4; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | sed -e 's/=*//' | tr '\n' ' ' >> $OUT; echo >> $OUT
    sleep 1
  done
  wait $BP
  cat gpurun_out/gb_scale$sc.log >> $OUT
done
cat $OUT
