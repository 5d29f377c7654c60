#!/bin/bash
TAG=${1:-r2w}
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["time_share"], d.get("text_tower_rows"))'
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
{ timeout 300 $B --workload bf16_b1024_fwd_loss 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --workload bf16_b1024_train 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --workload bf16_b1024_train_autograd 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --workload bf16_b1024_train --text-dropout 0.1 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --workload bf16_b1024_train_autograd --text-dropout 0.1 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --workload bf16_b1024_train 2>&1 | tail -1 | python -c "$P"; } > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
