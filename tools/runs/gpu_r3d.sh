#!/bin/bash
# Round 3, fourth call: the three tests that failed in r3c (host-side fixes), kernel trace of the ModifiedResNet-50 tower.
TAG=${1:-r3d}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_canary_gpu.py tests/test_resnet_gpu.py tests/test_zz_global_scope_gpu.py tests/test_zz_bench_contract_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR\|Error\|assert " gpurun_out/pytest_$TAG.log | head -30
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rn_$TAG -o rn -- python $R/tools/rn_bench.py > $R/gpurun_out/prof_rn_$TAG.log 2>&1
cd $R
DB=$(find /tmp/prof_rn_$TAG -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${TAG}_rn_kernel_stats.md > /dev/null 2>&1
head -24 gpurun_out/${TAG}_rn_kernel_stats.md | cut -c1-200
