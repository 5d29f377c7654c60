#!/bin/bash
# Round 5, call 13: a kernel trace of the default also-workload order (ten steps each): busy share and largest gaps per workload segment --
# where does the autograd training step idle when it runs as the 7th workload?
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5m}; R=$(pwd)
cd /tmp && EZCLIP_NO_CANARY=1 timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_line_$T -o bench -- python $R/bench.py --no-cpu-baseline --sustained-steps 0 --steps 10 --also bf16_b1024_fwd_loss_padded_text,bf16_b1024_train,bf16_b1024_train_padded_text,bf16_b1024_train_opt,bf16_b1024_fwd_loss_autograd,bf16_b1024_train_autograd > $R/gpurun_out/prof_line_$T.log 2>&1
cd $R
DB=$(find /tmp/prof_line_$T -name "*.db" | head -1)
grep '^{"metric"' gpurun_out/prof_line_$T.log | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print(" | ".join("%s %.1f" % (k.replace("bf16_b1024_", ""), v["ms_per_step"]) for k, v in d["also"].items()))'
python tools/trace_segments.py $DB 8 2>&1 | tee gpurun_out/trace_segments_$T.log | cut -c1-200
