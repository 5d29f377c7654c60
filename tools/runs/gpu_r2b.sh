#!/bin/bash
# Round 2, second GPU pass: full parity suite, streamed attention forward A/B, bench.
TAG=${1:-r2b}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 2>&1 | tail -120 > gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
for st in 0 1; do echo "== gemm_bench ATTN_STREAM=$st"; ATTN_STREAM=$st ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2; done > gpurun_out/gb_$TAG.log 2>&1
grep -v "^batch" gpurun_out/gb_$TAG.log
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["time_share"], d["attention_tflops"])'
for rep in 1 2; do
  echo "== base rep $rep"; timeout 300 $B 2>&1 | tail -1 | python -c "$P"
  echo "== attention load-then-compute"; EZCLIP_ATTN_STREAM=0 timeout 300 $B 2>&1 | tail -1 | python -c "$P"
done > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.log | cut -c1-300
