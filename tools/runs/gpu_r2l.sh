#!/bin/bash
TAG=${1:-r2l}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -80 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "FAILED" gpurun_out/pytest_$TAG.log | head
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["time_share"], d["model_mfma_frac"])'
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
for rep in 1 2; do timeout 300 $B 2>&1 | tail -1 | python -c "$P"; done > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
ATTN_PROBE=1 ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2 2>&1 | grep attn > gpurun_out/gb_attn_$TAG.log; cat gpurun_out/gb_attn_$TAG.log
