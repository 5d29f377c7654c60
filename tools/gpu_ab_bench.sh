#!/bin/bash
# usage: tools/gpu_ab_bench.sh <tag> <workload> <variant...>  -- bench.py with the in-tree lib vs experiment libs
# (tools/build_variants.py), alternating on one box; the in-tree library is never touched (EZCLIP_LIB selects the build)
TAG=$1; WL=$2; shift 2
mkdir -p gpurun_out; OUT=gpurun_out/ab_$TAG.log; : > $OUT
for rep in 1 2; do
  for v in base "$@"; do
    if [ $v = base ]; then unset EZCLIP_LIB; else export EZCLIP_LIB=$PWD/tools/bin/var_$v/libezclip_hip.so; fi
    echo "== $v (rep $rep)" >> $OUT
    timeout 300 python bench.py --steps 10 --warmup 3 --workload $WL --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])" >> $OUT
  done
done
unset EZCLIP_LIB
cat $OUT
