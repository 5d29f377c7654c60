#!/bin/bash
# Round 5, call 7: ModifiedResNet training backward, stage by stage against the oracle's explicit backward (tools/rn_train_stage_diff.py)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/rn_train_stage_diff.py 2>&1 | tee gpurun_out/rn_train_stage_diff_${1:-r5g}.log | cut -c1-220 | head -80
