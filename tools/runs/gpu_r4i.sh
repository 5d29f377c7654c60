#!/bin/bash
# Round 4, call 9: short attention forward with ra-row images and one wave per two query blocks (three workgroups per CU at 197 tokens):
# tests, op-level A/B (tools/attn_ab.py 11), headline A/B; the full-depth ViT-L/14 gradient check.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4i
timeout 900 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py -k "canary or attention" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_attn_$T.log
timeout 900 python -m pytest -x -q -m gpu tests/test_model_gpu.py tests/test_dropout.py tests/test_openclip_gpu.py -k "forward or golden or dropout or packed" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_model_$T.log
timeout 600 python tools/attn_ab.py 11 1 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_fwd_three_ab_$T.log
for v in 1 0 1 0; do
  EZCLIP_ATTN_FWD_THREE=$v EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 200 > gpurun_out/bench_tmp_$T.json 2> gpurun_out/bench_tmp_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_tmp_$T.json").read().strip().splitlines()[-1])
print("three=$v", d["value"], d["ms_per_step"], "sustained", d["sustained"]["ms_per_step_second_half"], d.get("time_share"), d.get("attention_tflops"))
PY
done 2>&1 | tee gpurun_out/bench_attn_three_ab_$T.log
timeout 1500 python -m pytest -x -q -s -m gpu tests/test_bench_regime_gpu.py -k "directional" 2>&1 | grep -E "passed|failed|Error|assert|FAILED|directional derivatives" | cut -c1-600 | tee gpurun_out/pytest_dd_$T.log
