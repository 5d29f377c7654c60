#!/bin/bash
# One GPU-box pass: parity tests, bench lines, rocprofv3 kernel trace (summaries under gpurun_out/).
# usage: tools/gpu_round.sh <tag> [pytest|nopytest]
TAG=${1:-r1}
R=$(pwd)
mkdir -p gpurun_out
if [ "${2:-pytest}" = "pytest" ]; then
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_$TAG.log
fi
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --workload bf16_b1024_train --no-cpu-baseline > gpurun_out/bench_train_$TAG.log 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --workload fp32_b256_fwd_sim --no-cpu-baseline > gpurun_out/bench_fp32_$TAG.log 2>&1
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_$TAG.md > /dev/null 2>&1
find /tmp/prof_$TAG -name "*kernel_stats*.csv" -exec cp {} gpurun_out/kernel_stats_$TAG.csv \;
tail -3 gpurun_out/pytest_$TAG.log; cat gpurun_out/bench_$TAG.log | tail -1 | cut -c1-400; tail -1 gpurun_out/bench_train_$TAG.log | cut -c1-300
