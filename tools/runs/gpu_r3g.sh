#!/bin/bash
# Round 3, seventh call: same-box A/B of the short last-tile path of the attention forward; HIP runtime calls per bench step.
TAG=${1:-r3g}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/attn_ab.py 9 > gpurun_out/attn_ab_$TAG.log 2>&1; cat gpurun_out/attn_ab_$TAG.log | tail -6
timeout 600 python tools/hip_api_per_step.py bf16_b1024_fwd_loss gpurun_out/${TAG}_hip_api_per_step.md > gpurun_out/hipapi_$TAG.log 2>&1; tail -25 gpurun_out/hipapi_$TAG.log
