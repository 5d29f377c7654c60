#!/bin/bash
TAG=${1:-r2j}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -80 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "FAILED" gpurun_out/pytest_$TAG.log | head
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["time_share"], d["model_mfma_frac"])'
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
for rep in 1 2; do
  for c in 2 1 4; do echo "== chunks $c"; EZCLIP_IMAGE_CHUNKS=$c timeout 300 $B 2>&1 | tail -1 | python -c "$P"; done
done > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
