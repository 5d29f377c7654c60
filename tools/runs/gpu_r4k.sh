#!/bin/bash
# Round 4, call 11: op-level A/B of the two-block attention forward (ezclip_debug_set(11, 1 / 0)) on the towers' shapes incl. ViT-L/14
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/attn_ab.py 11 1 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_fwd_two_block_ab_r4k.log
EZCLIP_NO_CANARY=1 timeout 600 python bench.py --workload bf16_vitl14_b512_fwd_loss --no-also --no-cpu-baseline --steps 10 --sustained-steps 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("vitl14 three=1", d["value"], d["ms_per_step"], d.get("time_share"))' | tee -a gpurun_out/attn_fwd_two_block_ab_r4k.log
EZCLIP_ATTN_FWD_THREE=0 EZCLIP_NO_CANARY=1 timeout 600 python bench.py --workload bf16_vitl14_b512_fwd_loss --no-also --no-cpu-baseline --steps 10 --sustained-steps 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("vitl14 three=0", d["value"], d["ms_per_step"], d.get("time_share"))' | tee -a gpurun_out/attn_fwd_two_block_ab_r4k.log
