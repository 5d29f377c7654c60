#!/bin/bash
# Round 5, call 15: where does the ModifiedResNet-50 training step spend its time? (rocprofv3 kernel trace, bf16, 256 images, 6 steps)
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5o}; R=$(pwd)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rn_$T -o rn -- python $R/tools/rn_train_profile.py > $R/gpurun_out/prof_rn_$T.log 2>&1
cd $R
DB=$(find /tmp/prof_rn_$T -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${T}_rn_train_kernel_stats.md "gemm" > /dev/null 2>&1
head -34 gpurun_out/${T}_rn_train_kernel_stats.md | cut -c1-190
