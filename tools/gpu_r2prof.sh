#!/bin/bash
# Round 2 profiles: rocprofv3 kernel traces of the forward and training workloads (towers on ONE stream, so that
# per-kernel durations are not inflated by the other tower's kernels sharing the CUs) + PMC passes over gemm_bench.
TAG=${1:-r2}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
for wl in fwd train; do
  W=""; [ $wl = train ] && W="--workload bf16_b1024_train"
  cd /tmp && EZCLIP_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${wl}_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also $W > $R/gpurun_out/prof_${wl}_$TAG.log 2>&1
  cd $R
  DB=$(find /tmp/prof_${wl}_$TAG -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${TAG}_${wl}_kernel_stats.md > /dev/null 2>&1
  tail -1 gpurun_out/prof_${wl}_$TAG.log | cut -c1-200
  head -16 gpurun_out/${TAG}_${wl}_kernel_stats.md
done
bash tools/pmc_gemm.sh $TAG
python tools/pmc_summary.py $TAG gpurun_out 2>&1 | tail -3
