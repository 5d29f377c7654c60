#!/bin/bash
# Round 3: attention forward row sums on the matrix pipe: op / model tests, same-box A/B.
TAG=${1:-r3i}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_canary_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_bench_regime_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "canary or attention or golden or folded or vitb16" 2>&1 | tail -20 > gpurun_out/pytest_focus_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_focus_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR\|Error\|assert " gpurun_out/pytest_focus_$TAG.log | head -20
timeout 300 python tools/attn_ab.py 9 3 1 > gpurun_out/attn_ab_$TAG.log 2>&1; tail -5 gpurun_out/attn_ab_$TAG.log
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["value"], d["ms_per_step"], d["loss"], d.get("model_mfma_frac"))'
B="python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 5"
{ for v in 1 2; do EZCLIP_NO_CANARY=1 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"; done; } > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
