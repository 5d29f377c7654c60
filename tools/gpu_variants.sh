#!/bin/bash
# usage: tools/gpu_variants.sh <tag> <nshapes> <variant names...>   -- gemm_bench against the experiment libs of tools/build_variants.py
TAG=$1; N=$2; shift 2
mkdir -p gpurun_out
OUT=gpurun_out/variants_$TAG.log; : > $OUT
for rep in 1 2; do
  echo "== base (rep $rep)" >> $OUT
  NT_SHAPES=$N timeout 120 tools/bin/gemm_bench 1024 20 2 >> $OUT 2>&1
  for v in "$@"; do
    echo "== $v (rep $rep)" >> $OUT
    NT_SHAPES=$N LD_LIBRARY_PATH=tools/bin/var_$v timeout 120 tools/bin/gemm_bench 1024 20 2 >> $OUT 2>&1
  done
done
cat $OUT
