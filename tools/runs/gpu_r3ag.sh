#!/bin/bash
# Round 3 (late): staged four-wave GEMM, reads issued before the MFMAs (default build) vs interleaved under them (-DEZ_Q_INTERLEAVE).
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/gemm4q_r3ag.log; : > $OUT
for v in base qil; do
  if [ $v = base ]; then LP=easynlp_amd/csrc; else LP=tools/bin/var_$v; fi
  for sc in 1 0; do
    echo "## lib=$v OPERAND_SCALE=$sc, 600 launches per shape (v2 = 8-phase, v4 = four-wave)" >> $OUT
    LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH OPERAND_SCALE=$sc NT_SHAPES=9 timeout 200 tools/bin/gemm_bench 1024 600 2,4 2>&1 | grep "vit.qkv\|fc+qgelu\|patch" >> $OUT
  done
done
cat $OUT
