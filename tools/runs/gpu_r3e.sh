#!/bin/bash
# Round 3, fifth call: last ViT block with CLS-row queries only (A/B), whole suite.
TAG=${1:-r3e}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_canary_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "canary or cls or golden or folded" 2>&1 | tail -40 > gpurun_out/pytest_focus_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_focus_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR\|Error\|assert " gpurun_out/pytest_focus_$TAG.log | head -20
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["value"], d["ms_per_step"], d["loss"], d.get("model_mfma_frac"))'
B="python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 5"
{ for v in 1 0 1 0; do EZCLIP_NO_CANARY=1 EZCLIP_CLS_Q_ONLY=$v timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"; done; } > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$TAG.log | head -30
