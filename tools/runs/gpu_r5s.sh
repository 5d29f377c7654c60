#!/bin/bash
# Round 5, call 19: the dedicated 3 x 3 weight-gradient kernel at 64 channels in and out (rn_wgrad3x3_c64: operands read once, the 64 x 576
# result in registers, LDS transpose reads): operator test, the ModifiedResNet test files, tools/rn_bench.py's training leg against
# EZCLIP_RN_EXPLICIT_IM2COL=2 (the generic kernel gathering the neighbourhoods) and =1 (explicit column matrix).
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5s}
{ timeout 600 python -m pytest tests/test_resnet_train_ops_gpu.py -m gpu -q --maxfail=8 -p no:cacheprovider -k "64_channels or neighbourhoods" 2>&1 | tail -40; } > gpurun_out/pytest_wgrad64_$T.log
grep -n "passed\|failed" gpurun_out/pytest_wgrad64_$T.log | tail -2; grep -n "^FAILED\|^ERROR\|Error\|assert " gpurun_out/pytest_wgrad64_$T.log | head -12
if grep -q "failed\|error" gpurun_out/pytest_wgrad64_$T.log; then exit 1; fi
{ timeout 900 python -m pytest tests/test_resnet_train_ops_gpu.py tests/test_resnet_train_gpu.py tests/test_resnet_gpu.py -m gpu -q --maxfail=8 -p no:cacheprovider 2>&1 | tail -30; } > gpurun_out/pytest_rn_$T.log
grep -n "passed\|failed" gpurun_out/pytest_rn_$T.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_rn_$T.log | head
for mode in 2 0 2 0; do
  echo "== EZCLIP_RN_EXPLICIT_IM2COL=$mode"; RN_BENCH_TRAIN_ONLY=1 EZCLIP_RN_EXPLICIT_IM2COL=$mode timeout 300 python tools/rn_bench.py 2>&1 | grep TRAIN
done 2>&1 | tee gpurun_out/rn_bench_ab_$T.log
