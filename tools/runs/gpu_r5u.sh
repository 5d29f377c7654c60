#!/bin/bash
# Round 5, call 21: the operands-once weight-gradient kernels widened -- 3 x 3 at 128 channels (sub-problems of 64 x 64), the 1 x 1 / stem
# products with a small result (rn_tn_skinny): operator tests, the ModifiedResNet test files, tools/rn_bench.py's training leg with
# EZCLIP_RN_EXPLICIT_IM2COL=3 (generic 1 x 1 products) against the default.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5u}
{ timeout 600 python -m pytest tests/test_resnet_train_ops_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "64_channel or small_result" 2>&1 | tail -60; } > gpurun_out/pytest_wgrad_$T.log
grep -n "passed\|failed" gpurun_out/pytest_wgrad_$T.log | tail -2; grep -n "^FAILED\|^ERROR\|^E  " gpurun_out/pytest_wgrad_$T.log | head -24
if grep -q "failed\|error" gpurun_out/pytest_wgrad_$T.log; then exit 1; fi
{ timeout 900 python -m pytest tests/test_resnet_train_ops_gpu.py tests/test_resnet_train_gpu.py tests/test_resnet_gpu.py -m gpu -q --maxfail=8 -p no:cacheprovider 2>&1 | tail -30; } > gpurun_out/pytest_rn_$T.log
grep -n "passed\|failed" gpurun_out/pytest_rn_$T.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_rn_$T.log | head
for mode in 3 0 3 0; do
  echo "== EZCLIP_RN_EXPLICIT_IM2COL=$mode"; RN_BENCH_TRAIN_ONLY=1 EZCLIP_RN_EXPLICIT_IM2COL=$mode timeout 300 python tools/rn_bench.py 2>&1 | grep TRAIN
done 2>&1 | tee gpurun_out/rn_bench_ab_$T.log
