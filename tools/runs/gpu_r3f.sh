#!/bin/bash
# Round 3, sixth call: attention forward with the short last-tile path: op tests, stand-alone timing, bench.
TAG=${1:-r3f}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_canary_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "canary or attention or golden or folded or packed" 2>&1 | tail -30 > gpurun_out/pytest_focus_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_focus_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR\|Error\|assert " gpurun_out/pytest_focus_$TAG.log | head -20
ATTN_PROBE=1 ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2 2>&1 | grep attn > gpurun_out/gb_attn_$TAG.log; cat gpurun_out/gb_attn_$TAG.log
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["value"], d["ms_per_step"], d["loss"], d.get("model_mfma_frac"), d.get("attention_tflops"))'
B="python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 5"
{ for v in 1 2; do EZCLIP_NO_CANARY=1 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"; done; } > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
