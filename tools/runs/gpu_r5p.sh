#!/bin/bash
# Round 5, call 16: the BatchNorm kernels of the ModifiedResNet training path reworked (finalise: one wave per channel; moments / apply:
# 16-byte chunks, full blocks at 64 channels): the RN tests, tools/rn_bench.py, and the kernel breakdown again.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5p}; R=$(pwd)
timeout 900 python -m pytest -q -m gpu tests/test_resnet_train_ops_gpu.py tests/test_resnet_train_gpu.py tests/test_resnet_gpu.py --maxfail=30 > gpurun_out/pytest_rn_$T.log 2>&1
grep -E "^E  |passed|failed|^FAILED|fault" gpurun_out/pytest_rn_$T.log | cut -c1-300 | head -20
timeout 300 python tools/rn_bench.py 2>&1 | tail -6 | tee gpurun_out/rn_bench_$T.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rn_$T -o rn -- python $R/tools/rn_train_profile.py > $R/gpurun_out/prof_rn_$T.log 2>&1
cd $R
DB=$(find /tmp/prof_rn_$T -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${T}_rn_train_kernel_stats.md "gemm" > /dev/null 2>&1
head -16 gpurun_out/${T}_rn_train_kernel_stats.md | cut -c1-170; grep "total kernel time" gpurun_out/${T}_rn_train_kernel_stats.md
