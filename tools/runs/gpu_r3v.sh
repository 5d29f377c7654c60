#!/bin/bash
# Round 3: is the epilogue's store tail the in-order vmcnt of the DMA-waiting wave?  rolex: DMA by wave row 0, stores by wave row 1;
# roley: both by wave row 0 (same traffic; results wrong by construction, timing only).  tools/build_variants.py rolex:-DEZ_ROLE_X roley:-DEZ_ROLE_Y
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/gemm_roles_r3v.log
: > $OUT
for sc in 0 1; do
  for v in base rolex roley rolex roley base; do
    echo "## OPERAND_SCALE=$sc variant=$v" >> $OUT
    if [ $v = base ]; then LP=easynlp_amd/csrc; else LP=tools/bin/var_$v; fi
    LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH OPERAND_SCALE=$sc NT_SHAPES=4 timeout 120 tools/bin/gemm_bench 1024 300 2 2>&1 | grep -v "^batch" >> $OUT
  done
done
cat $OUT
