#!/bin/bash
# Round 2, closing pass: full parity suite, the default bench line (every BASELINE config), clean kernel traces, PMC passes,
# vendor calibration.
TAG=${1:-r2}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -80 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "FAILED" gpurun_out/pytest_$TAG.log | head
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.json | cut -c1-300
for wl in fwd train; do
  W=""; [ $wl = train ] && W="--workload bf16_b1024_train"
  cd /tmp && EZCLIP_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${wl}_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also $W > $R/gpurun_out/prof_${wl}_$TAG.log 2>&1
  cd $R
  DB=$(find /tmp/prof_${wl}_$TAG -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${TAG}_${wl}_kernel_stats.md > /dev/null 2>&1
  head -14 gpurun_out/${TAG}_${wl}_kernel_stats.md
done
bash tools/pmc_gemm.sh $TAG > /dev/null 2>&1
bash tools/pmc_attn.sh $TAG > /dev/null 2>&1; head -30 gpurun_out/pmc_attn_$TAG.md | tail -24
python tools/pmc_summary.py $TAG gpurun_out > /dev/null 2>&1; head -16 gpurun_out/${TAG}_gemm_pmc.md | tail -10
ATTN_PROBE=1 ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2 2>&1 | grep attn > gpurun_out/gb_attn_$TAG.log; cat gpurun_out/gb_attn_$TAG.log
timeout 300 tools/bin/gemm_bench 1024 10 2 2>&1 | grep -v "^batch" > gpurun_out/gb_$TAG.log; head -14 gpurun_out/gb_$TAG.log
timeout 300 python tools/vendor_calibration.py > gpurun_out/vendor_$TAG.log 2>&1; cat gpurun_out/vendor_$TAG.log | tail -8
