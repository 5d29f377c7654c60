#!/bin/bash
# Round 4, call 13: the direct epilogue's 64-byte half-line stores without the non-temporal hint (can L2 merge the two halves of a line?)
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4m
for v in base direct_nont new base direct_nont; do
  L=tools/bin/var_$v; [ $v = new ] && L=easynlp_amd/csrc
  echo "== $v: gemm_bench 1024 300 2"; LD_LIBRARY_PATH=$L NT_SHAPES=11 timeout 300 tools/bin/gemm_bench 1024 300 2 2>&1 | grep -v "^batch"
done > gpurun_out/gb_epi_ab_$T.log 2>&1
cat gpurun_out/gb_epi_ab_$T.log
