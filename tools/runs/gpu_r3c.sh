#!/bin/bash
# Round 3, third call: ModifiedResNet tower (tests, throughput), then the whole suite again (attention op options, contrastive-step
# policy of the f32 pipeline).
TAG=${1:-r3c}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_canary_gpu.py tests/test_resnet_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/pytest_rn_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_rn_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR\|Error\|assert " gpurun_out/pytest_rn_$TAG.log | head -30
timeout 300 python tools/rn_bench.py > gpurun_out/rn_bench_$TAG.log 2>&1; tail -5 gpurun_out/rn_bench_$TAG.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --durations=8 -p no:cacheprovider 2>&1 | tail -90 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$TAG.log | head -30
