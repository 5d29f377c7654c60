#!/bin/bash
TAG=${1:-r2y}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -k "dropout or packed or attention or golden or overfit" 2>&1 | tail -150 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "FAILED\|Error" gpurun_out/pytest_$TAG.log | head
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["value"], d["ms_per_step"], d["loss"], d.get("text_tower_rows"))'
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
{ timeout 300 $B --workload bf16_b1024_train --text-dropout 0.1 2>&1 | tail -1 | python -c "$P"
  EZCLIP_PACK_TEXT=0 timeout 300 $B --workload bf16_b1024_train --text-dropout 0.1 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --workload bf16_b1024_train 2>&1 | tail -1 | python -c "$P"; } > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
