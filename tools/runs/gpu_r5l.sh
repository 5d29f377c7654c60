#!/bin/bash
# Round 5, call 12: does gc.freeze() (bench.py) remove the autograd training step's slowdown in the default workload order?
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5l}
run() { EZCLIP_NO_CANARY=1 timeout 900 python bench.py --no-cpu-baseline --sustained-steps 0 --steps 10 --also bf16_b1024_fwd_loss_padded_text,bf16_b1024_train,bf16_b1024_train_padded_text,bf16_b1024_train_opt,bf16_b1024_fwd_loss_autograd,bf16_b1024_train_autograd 2> /dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("   " + " | ".join("%s %.1f (%s MHz, busy %.3f)" % (k.replace("bf16_b1024_", ""), v["ms_per_step"], v.get("clock_mhz_timed_steps"), sum((v.get("time_share") or {}).values())) for k, v in d["also"].items()))'; }
echo "gc.freeze (default):"; run 2>&1 | tee gpurun_out/gc_freeze_$T.log
echo "EZCLIP_BENCH_NO_GC_FREEZE=1:"; EZCLIP_BENCH_NO_GC_FREEZE=1 run 2>&1 | tee -a gpurun_out/gc_freeze_$T.log
echo "gc.freeze (default), again:"; run 2>&1 | tee -a gpurun_out/gc_freeze_$T.log
