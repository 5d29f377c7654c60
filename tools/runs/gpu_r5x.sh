#!/bin/bash
# Round 5, call 23: the 16-lane partial-sum kernel of rn_tn_skinny (operator tests) and the strip length of rn_tn_skinny (EZCLIP_RN_SKINNY_P:
# shorter strips = fewer load rounds per strip and more workgroups per CU where the registers allow) on tools/rn_bench.py's training leg.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5x}
{ timeout 600 python -m pytest tests/test_resnet_train_ops_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "small_result" 2>&1 | tail -30; } > gpurun_out/pytest_skinny_$T.log
grep -n "passed\|failed" gpurun_out/pytest_skinny_$T.log | tail -2; grep -n "^FAILED\|^ERROR\|^E  " gpurun_out/pytest_skinny_$T.log | head -12
if grep -q "failed\|error" gpurun_out/pytest_skinny_$T.log; then exit 1; fi
for cap in 0 128 64 0 128 64 32; do
  echo "== EZCLIP_RN_SKINNY_P=$cap"; RN_BENCH_TRAIN_ONLY=1 RN_BENCH_BF16_ONLY=1 EZCLIP_RN_SKINNY_P=$cap timeout 300 python tools/rn_bench.py 2>&1 | grep TRAIN
done 2>&1 | tee gpurun_out/rn_bench_skinny_p_$T.log
