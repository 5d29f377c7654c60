#!/bin/bash
# Round 5, call 5: the ModifiedResNet training tower's gradient errors, every parameter above 2e-4, four configurations (which width /
# which layer injects the 0.3-0.9 % error seen at width 64 in fp32?)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/rn_train_diag.py 2e-4 2>&1 | grep -v "^\[ezclip\]" > gpurun_out/rn_train_diag_full_${1:-r5e}.log; wc -l gpurun_out/rn_train_diag_full_${1:-r5e}.log
