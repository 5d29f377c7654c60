#!/bin/bash
# Round 4, call 16: where the autograd training path loses its 4 % to the fused step: kernel traces of both, busy time and gaps
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4p; R=$(pwd)
for wl in bf16_b1024_train bf16_b1024_train_autograd; do
  cd /tmp && EZCLIP_NO_CANARY=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_${wl}_$T -o bench -- python $R/bench.py --steps 12 --warmup 3 --sustained-steps 0 --no-cpu-baseline --no-also --workload $wl > $R/gpurun_out/prof_${wl}_$T.log 2>&1
  cd $R
  DB=$(find /tmp/prof_${wl}_$T -name "*.db" | head -1)
  echo "== $wl"; grep '^{"metric"' gpurun_out/prof_${wl}_$T.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "value", d["value"])'
  python tools/trace_busy.py $DB 0.4 40
done 2>&1 | tee gpurun_out/train_vs_autograd_trace_$T.log
