#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/rn_train_where.py 2>&1 | tee gpurun_out/rn_train_where_${1:-r5h}.log | cut -c1-260 | head -60
