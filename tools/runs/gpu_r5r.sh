#!/bin/bash
# Round 5, call 18: the 3 x 3 weight gradients of the ModifiedResNet training path WITHOUT the explicit column matrix (the generic
# weight-gradient kernel gathers the neighbourhoods itself: GemmTNArgs::conv_H): the operator test (bit identity with the explicit
# route), the tower tests, and tools/rn_bench.py's training leg against EZCLIP_RN_EXPLICIT_IM2COL=1 in the same process order.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5r}
{ timeout 900 python -m pytest tests/test_resnet_train_ops_gpu.py tests/test_resnet_train_gpu.py tests/test_resnet_gpu.py -m gpu -q --maxfail=8 -p no:cacheprovider 2>&1 | tail -30; } > gpurun_out/pytest_rn_$T.log
grep -n "passed\|failed" gpurun_out/pytest_rn_$T.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_rn_$T.log | head
for rep in 1 2; do
  echo "== explicit im2col (rep $rep)"; RN_BENCH_TRAIN_ONLY=1 EZCLIP_RN_EXPLICIT_IM2COL=1 timeout 300 python tools/rn_bench.py 2>&1 | grep TRAIN
  echo "== neighbourhoods gathered by the product (rep $rep)"; RN_BENCH_TRAIN_ONLY=1 timeout 300 python tools/rn_bench.py 2>&1 | grep TRAIN
done 2>&1 | tee gpurun_out/rn_bench_ab_$T.log
