#!/bin/bash
# rocprofv3 kernel trace of the training workload + FETCH_SIZE / WRITE_SIZE passes over the GEMM bench
TAG=${1:-r1}
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/proft_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --workload bf16_b1024_train --no-cpu-baseline > $R/gpurun_out/proft_$TAG.log 2>&1
cd $R
DB=$(find /tmp/proft_$TAG -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_train_$TAG.md > /dev/null 2>&1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 90 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc2_$c -o g --output-format csv -- $R/tools/bin/gemm_bench 1024 1 2 > /tmp/pmc2_$c.log 2>&1
  f=$(find /tmp/pmc2_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmc_${c}_$TAG.csv || tail -3 /tmp/pmc2_$c.log > $R/gpurun_out/pmc_${c}_$TAG.err
done
head -30 $R/gpurun_out/kernel_stats_train_$TAG.md
