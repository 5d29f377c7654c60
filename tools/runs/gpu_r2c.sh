#!/bin/bash
TAG=${1:-r2c}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -60 > gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2 > gpurun_out/gb_$TAG.log 2>&1
grep -v "^batch" gpurun_out/gb_$TAG.log
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["time_share"], d["attention_tflops"])'
for rep in 1 2 3; do timeout 300 $B 2>&1 | tail -1 | python -c "$P"; done > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
timeout 300 python bench.py --no-also --no-cpu-baseline --steps 5 --warmup 2 --workload bf16_b1024_train 2>&1 | tail -1 | python -c "$P"
