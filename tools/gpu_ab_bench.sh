#!/bin/bash
# usage: tools/gpu_ab_bench.sh <tag> <workload> <variant...>  -- bench.py with the in-tree lib vs experiment libs, alternating on one box
TAG=$1; WL=$2; shift 2
mkdir -p gpurun_out; OUT=gpurun_out/ab_$TAG.log; : > $OUT
L=easynlp_amd/csrc/libezclip_hip.so
cp $L /tmp/base.so
for rep in 1 2; do
  for v in base "$@"; do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp tools/bin/var_$v/libezclip_hip.so $L; fi
    echo "== $v (rep $rep)" >> $OUT
    timeout 300 python bench.py --steps 10 --warmup 3 --workload $WL --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])" >> $OUT
  done
done
cp /tmp/base.so $L
cat $OUT
