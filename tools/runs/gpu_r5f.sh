#!/bin/bash
# Round 5, call 6: (1) is the ModifiedResNet training backward deterministic? (three passes on one saved state, per-stage checksums);
# (2) the autograd training step reads 3-4 % slow as the 7th workload of the default line -- does it with the caching allocator kept
# warm (EZCLIP_BENCH_KEEP_CACHE=1) / in the default order but with only the training workloads?
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5f}
timeout 900 python tools/rn_train_determinism.py 2>&1 | tee gpurun_out/rn_train_determinism_$T.log | cut -c1-200 | head -60
run() { EZCLIP_NO_CANARY=1 timeout 900 python bench.py --no-cpu-baseline --sustained-steps 0 --steps 10 "$@" 2> gpurun_out/err_$T.log | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("   ", " | ".join("%s %.1f" % (k.replace("bf16_b1024_", ""), v.get("ms_per_step") or -1) for k, v in (d.get("also") or {}).items()))'; }
ALL=bf16_b1024_fwd_loss_padded_text,bf16_b1024_train,bf16_b1024_train_padded_text,bf16_b1024_train_opt,bf16_b1024_fwd_loss_autograd,bf16_b1024_train_autograd
echo "default order, empty_cache between workloads:"; run --also $ALL 2>&1 | tee gpurun_out/autograd_order_$T.log
echo "default order, allocator cache kept:"; EZCLIP_BENCH_KEEP_CACHE=1 run --also $ALL 2>&1 | tee -a gpurun_out/autograd_order_$T.log
echo "autograd first, then fused:"; run --also bf16_b1024_train_autograd,bf16_b1024_train 2>&1 | tee -a gpurun_out/autograd_order_$T.log
