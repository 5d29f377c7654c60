#!/bin/bash
# Round 4, call 22: the whole default sequence without the two sustained legs (thermal / power history or cumulative state?)
mkdir -p gpurun_out; export TMPDIR=/tmp
EZCLIP_NO_CANARY=1 timeout 200 python bench.py --no-cpu-baseline --sustained-steps 0 --steps 6 --also-steps 6 --also bf16_b1024_fwd_loss_padded_text,bf16_b1024_train,bf16_b1024_train_padded_text,bf16_b1024_train_opt,bf16_b1024_fwd_loss_autograd,bf16_b1024_train_autograd,bf16_b1024_train 2> gpurun_out/err_r4v.log | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("   ", " | ".join("%s %.1f" % (k.replace("bf16_b1024_", ""), v.get("ms_per_step") or -1) for k, v in (d.get("also") or {}).items()))'
