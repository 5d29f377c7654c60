#!/bin/bash
TAG=${1:-r2r}
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
  for pf in 0 256 128 512; do
    echo "== prefetch $pf"
    ATTN_PF=$pf ATTN_PROBE=1 ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2 2>&1 | grep "bwd"
  done
done > gpurun_out/gb_attn_$TAG.log 2>&1; cat gpurun_out/gb_attn_$TAG.log
timeout 600 python -m pytest tests -m gpu -q --maxfail=12 -k "attention or packed or golden" 2>&1 | tail -5
