#!/bin/bash
# Round 3: inference towers as captured hipGraphs: tests, serving-size latency; chunked ModifiedResNet batch test.
TAG=${1:-r3k}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_canary_gpu.py tests/test_engine_state_gpu.py tests/test_resnet_gpu.py tests/test_model_gpu.py tests/test_openclip_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "canary or graph or chunks or predictor or cls_rows" 2>&1 | tail -40 > gpurun_out/pytest_focus_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_focus_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR\|Error\|assert " gpurun_out/pytest_focus_$TAG.log | head -20
timeout 300 python tools/predict_latency.py > gpurun_out/latency_$TAG.log 2>&1; tail -7 gpurun_out/latency_$TAG.log
