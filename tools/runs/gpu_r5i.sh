#!/bin/bash
# Round 5, call 9: the ModifiedResNet training tests with the ReLU-decision-aware comparison (fp32) and the torch-bf16-relative bound (bf16)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_resnet_train_ops_gpu.py tests/test_resnet_train_gpu.py tests/test_resnet_gpu.py --maxfail=30 > gpurun_out/pytest_rn_train_${1:-r5i}.log 2>&1
grep -E "^E  |passed|failed|^FAILED|fault" gpurun_out/pytest_rn_train_${1:-r5i}.log | cut -c1-400 | head -30
