#!/bin/bash
TAG=${1:-r2k}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -80 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "FAILED" gpurun_out/pytest_$TAG.log | head
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["time_share"], d["model_mfma_frac"])'
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
for rep in 1 2 3; do timeout 300 $B 2>&1 | tail -1 | python -c "$P"; done > gpurun_out/ab_$TAG.log 2>&1
for wl in bf16_b1024_train bf16_b1024_train_autograd; do timeout 300 $B --workload $wl --steps 5 2>&1 | tail -1 | python -c "$P"; done >> gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
cd /tmp && EZCLIP_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > $OLDPWD/gpurun_out/prof_$TAG.log 2>&1
cd $OLDPWD
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${TAG}_fwd_kernel_stats.md > /dev/null 2>&1
sed -n 3,22p gpurun_out/${TAG}_fwd_kernel_stats.md
