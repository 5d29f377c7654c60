#!/bin/bash
# Round 3: what the caller's optimizer costs per training step: torch's fused AdamW vs a per-parameter Python loop (tools/soak_train.py).
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/soak_opt_r3z.log
for k in fused loop fused loop; do timeout 600 python tools/soak_train.py 100 0.0 $k 2>&1 | grep "optimizer\|soak OK\|last 25" >> gpurun_out/soak_opt_r3z.log; done
cat gpurun_out/soak_opt_r3z.log
