#!/bin/bash
# Round 3: image tower as k concurrent sub-batches (tail-round filling) A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/split_image_ab.py 40 > gpurun_out/split_image_r3t.log 2>&1
cat gpurun_out/split_image_r3t.log | grep -v amdgpu.ids
