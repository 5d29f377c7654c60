#!/bin/bash
TAG=${1:-r2q}
mkdir -p gpurun_out
for rep in 1 2; do
  for v in base st16; do
    echo "== $v"; L=""; [ $v != base ] && L=tools/bin/var_$v
    LD_LIBRARY_PATH=$L ATTN_PROBE=1 ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2 2>&1 | grep "bwd"
  done
done > gpurun_out/gb_attn_$TAG.log 2>&1; cat gpurun_out/gb_attn_$TAG.log
bash tools/pmc_attn.sh $TAG 2>&1 | grep -A40 "attn_bwd_short_kernel<false" | head -60
