#!/bin/bash
TAG=${1:-r2s}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --maxfail=12 -k "attention or packed or golden or bench_regime" 2>&1 | tail -5
ATTN_PROBE=1 ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2 2>&1 | grep "attn" > gpurun_out/gb_attn_$TAG.log 2>&1; cat gpurun_out/gb_attn_$TAG.log
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["time_share"], d.get("layernorm_gbps"))'
B="python bench.py --no-also --no-cpu-baseline --steps 6 --warmup 2"
for wl in bf16_vitl14_b512_train bf16_vitl14_b512_fwd_loss bf16_b1024_train; do timeout 300 $B --workload $wl 2>&1 | tail -1 | python -c "$P"; done > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
