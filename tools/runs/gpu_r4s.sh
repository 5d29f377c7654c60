#!/bin/bash
# Round 4, call 19: why the autograd training step reads 4 % slow inside the default line and not alone
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4s
run() { EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-cpu-baseline --sustained-steps 0 "$@" 2> gpurun_out/err_$T.log | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("  headline", d["config"]["workload"], d["ms_per_step"])
for k, v in (d.get("also") or {}).items(): print("  also", k, v.get("ms_per_step"), v.get("error"))'; }
echo "A: autograd train alone as the headline (3 + 8 steps)"; run --workload bf16_b1024_train_autograd --no-also --steps 8 --warmup 3
echo "B: forward headline, then autograd train as the only also-workload"; run --steps 10 --also bf16_b1024_train_autograd
echo "C: forward headline, then fused train, then autograd train"; run --steps 10 --also bf16_b1024_train,bf16_b1024_train_autograd
echo "D: forward headline, then autograd train, then fused train"; run --steps 10 --also bf16_b1024_train_autograd,bf16_b1024_train
