#!/bin/bash
# Round 4, call 21 (last GPU minutes): the two most suspect predecessors of the autograd training step
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4u
run() { EZCLIP_NO_CANARY=1 timeout 200 python bench.py --no-cpu-baseline --sustained-steps 0 --steps 6 --also-steps 6 "$@" 2> gpurun_out/err_$T.log | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("   ", " | ".join("%s %.1f" % (k.replace("bf16_b1024_", ""), v.get("ms_per_step") or -1) for k, v in (d.get("also") or {}).items()))'; }
A=bf16_b1024_train_autograd
echo "after fwd_loss_autograd:"; run --also bf16_b1024_fwd_loss_autograd,$A
echo "after train_opt:"; run --also bf16_b1024_train_opt,$A
