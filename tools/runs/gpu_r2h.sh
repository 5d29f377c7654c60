#!/bin/bash
TAG=${1:-r2h}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -80 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "FAILED" gpurun_out/pytest_$TAG.log | head
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["time_share"], d.get("text_tower_rows"), d["gflop_per_pair"], d["model_mfma_frac"])'
for wl in bf16_b1024_train bf16_b1024_train_autograd; do
  B="python bench.py --no-also --no-cpu-baseline --steps 5 --warmup 2 --workload $wl"
  echo "== $wl packed"; timeout 300 $B 2>&1 | tail -1 | python -c "$P"
  echo "== $wl padded"; EZCLIP_PACK_TEXT=0 timeout 300 $B 2>&1 | tail -1 | python -c "$P"
done > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
