#!/bin/bash
# Round 3: kernel trace of the final tree's headline step, long enough (20 timed steps) that the cold first launches are diluted, with the
# dominant kernel's average split by position in the timeline next to bench.py's own roofline.avg_launch_us of the SAME run.
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
for wl in fwd train; do
  W=""; [ $wl = train ] && W="--workload bf16_b1024_train"
  cd /tmp && EZCLIP_NO_CANARY=1 EZCLIP_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${wl}_ac -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also $W > $R/gpurun_out/prof_${wl}_r3ac.log 2>&1
  cd $R
  DB=$(find /tmp/prof_${wl}_ac -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/r3ac_${wl}_kernel_stats.md "gemm_(nt|tn)(_8p)?_kernel" > /dev/null 2>&1
  head -9 gpurun_out/r3ac_${wl}_kernel_stats.md | cut -c1-150; tail -9 gpurun_out/r3ac_${wl}_kernel_stats.md
  grep '^{"metric"' gpurun_out/prof_${wl}_r3ac.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("bench line of this run: ms_per_step", d["ms_per_step"], "roofline avg_launch_us", r["avg_launch_us"], "launches_per_step", r["launches_per_step"], "-> GEMM ms per step", round(r["avg_launch_us"]*r["launches_per_step"]/1e3,3), "frac", r["frac"])'
done
