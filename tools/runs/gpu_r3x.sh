#!/bin/bash
# Round 3: soak run of the training path at the headline size (tools/soak_train.py): 400 optimizer steps, dropout 0.1.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/soak_train.py 400 0.1 2>&1 | grep -v amdgpu.ids > gpurun_out/soak_r3x.log
tail -24 gpurun_out/soak_r3x.log
