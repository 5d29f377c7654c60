#!/bin/bash
# (written in round 4, run in round 5 call r5c: profiles/r5_autograd_bisect.log) which predecessor makes bf16_b1024_train_autograd read 135-137 ms instead of 131.8 in the
# default line?  Alone, or after the headline / the fused train workload, it does not (tools/runs/gpu_r4s.sh).  Each line: forward
# headline (10 steps), then the listed also-workloads in order; prints ms per step of each.  ~25 s per line.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5a}
run() { EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-cpu-baseline --sustained-steps 0 --steps 10 "$@" 2> gpurun_out/err_$T.log | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("   ", " | ".join("%s %.1f" % (k.replace("bf16_b1024_", ""), v.get("ms_per_step") or -1) for k, v in (d.get("also") or {}).items()))'; }
A=bf16_b1024_train_autograd
for pre in bf16_b1024_fwd_loss_padded_text bf16_b1024_train_padded_text bf16_b1024_train_opt bf16_b1024_fwd_loss_autograd; do
  echo "after $pre:"; run --also $pre,$A
done 2>&1 | tee gpurun_out/autograd_bisect_$T.log
echo "after train_opt + fwd_loss_autograd:"; run --also bf16_b1024_train_opt,bf16_b1024_fwd_loss_autograd,$A 2>&1 | tee -a gpurun_out/autograd_bisect_$T.log
echo "PYTHONMALLOC / allocator view of the slow case: torch.cuda.memory_stats before the timed steps"; EZCLIP_BENCH_MEMSTATS=1 run --also bf16_b1024_train_opt,bf16_b1024_fwd_loss_autograd,$A 2>&1 | tee -a gpurun_out/autograd_bisect_$T.log
