#!/bin/bash
# rocprofv3 kernel trace of the training workload -> gpurun_out/kernel_stats_train_<tag>.md
TAG=${1:-r1}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/proft_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --workload bf16_b1024_train --no-cpu-baseline > $R/gpurun_out/proft_$TAG.log 2>&1
cd $R
DB=$(find /tmp/proft_$TAG -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_train_$TAG.md > /dev/null 2>&1
head -24 $R/gpurun_out/kernel_stats_train_$TAG.md
