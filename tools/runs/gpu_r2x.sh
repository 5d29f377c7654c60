#!/bin/bash
TAG=${1:-r2x}
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["value"], d["ms_per_step"], d["loss"])'
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
for rep in 1 2; do
  for pace in 0 1; do
    for wl in bf16_b1024_fwd_loss bf16_b1024_train; do
      echo -n "pace=$pace "; EZCLIP_PACE=$pace timeout 300 $B --workload $wl 2>&1 | tail -1 | python -c "$P"
    done
  done
done > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
