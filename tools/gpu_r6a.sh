#!/bin/bash
# Round 6, call A: the ModifiedResNet training parity edges first, then the whole -m gpu suite.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_resnet_train_gpu.py tests/test_resnet_train_ops_gpu.py tests/test_resnet_gpu.py -q -x -s -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r6a_pytest_rn.log
tail -15 gpurun_out/r6a_pytest_rn.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r6a_pytest_all.log
grep -n "passed\|failed" gpurun_out/r6a_pytest_all.log | tail -3; grep -n "^FAILED\|^ERROR" gpurun_out/r6a_pytest_all.log | head
