#!/bin/bash
# Round 4, call 7: the score-tile-once attention backward after the pad fix (NaN on ragged lengths in call 6): tests, op-level A/B,
# training-step A/B (default = once up to 128 tokens; 0 = two-pass only; 2 = once wherever eligible).
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4g
timeout 900 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py -k "canary or attention" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_attn_$T.log
timeout 900 python -m pytest -x -q -m gpu tests/test_dropout.py tests/test_openclip_gpu.py 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_drop_$T.log
timeout 1200 python -m pytest -x -q -m gpu tests/test_model_gpu.py tests/test_hf_gpu.py -k "backward or packed or overfit or accumulation" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_model_$T.log
timeout 600 python tools/attn_bwd_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_bwd_ab_$T.log
for v in 1 0 2 1 0; do
  EZCLIP_ATTN_BWD_ONCE=$v EZCLIP_NO_CANARY=1 timeout 600 python bench.py --workload bf16_b1024_train --no-also --no-cpu-baseline --steps 10 --sustained-steps 0 > gpurun_out/bench_train_once${v}_$T.json 2> gpurun_out/bench_train_once${v}_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_train_once${v}_$T.json").read().strip().splitlines()[-1])
print("once=$v train", d["value"], d["ms_per_step"], d["model_mfma_frac"], d.get("time_share"), d["loss"])
PY
done 2>&1 | tee gpurun_out/bench_train_once_ab_$T.log
