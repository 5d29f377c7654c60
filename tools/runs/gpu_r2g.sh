#!/bin/bash
TAG=${1:-r2g}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -80 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "FAILED" gpurun_out/pytest_$TAG.log | head
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["time_share"], d.get("text_tower_rows"), d["gflop_per_pair"], d["model_mfma_frac"])'
for rep in 1 2; do
  echo "== packed rep $rep"; timeout 300 $B 2>&1 | tail -1 | python -c "$P"
  echo "== padded"; EZCLIP_PACK_TEXT=0 timeout 300 $B 2>&1 | tail -1 | python -c "$P"
done > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
