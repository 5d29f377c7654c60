#!/bin/bash
TAG=${1:-r2v}
mkdir -p gpurun_out
for rep in 1 2; do
  for v in base pipea; do
    echo "== $v"; L=""; [ $v != base ] && L=tools/bin/var_$v
    LD_LIBRARY_PATH=$L ATTN_PROBE=1 ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2 2>&1 | grep "bwd"
  done
done > gpurun_out/gb_attn_$TAG.log 2>&1; cat gpurun_out/gb_attn_$TAG.log
timeout 600 python -m pytest tests -m gpu -q --maxfail=12 -k "attention or folded or bench_regime" 2>&1 | tail -3
