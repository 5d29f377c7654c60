#!/bin/bash
# Round 2, first GPU pass: the full parity suite (incl. the bench-regime tests), the bench line with every BASELINE
# config, A/B runs of this round's switches, LayerNorm-backward A/B, rocprofv3 kernel traces.
TAG=${1:-r2a}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -60 > gpurun_out/pytest_$TAG.log
tail -5 gpurun_out/pytest_$TAG.log
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.log | cut -c1-600
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
for rep in 1 2; do
  echo "== base rep $rep"; timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  echo "== one stream";    EZCLIP_TWO_STREAMS=0 timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  echo "== raster 4";      EZCLIP_RASTER_GM=4 timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  echo "== raster 8";      EZCLIP_RASTER_GM=8 timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  echo "== no qkv fusion"; EZCLIP_FUSE_QKV=0 timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
for gm in 0 4 8; do echo "== gemm_bench RASTER_GM=$gm"; RASTER_GM=$gm NT_SHAPES=8 timeout 300 tools/bin/gemm_bench 1024 10 2; done > gpurun_out/gb_$TAG.log 2>&1
grep -v "^batch" gpurun_out/gb_$TAG.log
( echo "== ln_bwd default"; timeout 120 python tools/lnbwd_bench.py; echo "== ln_bwd packed"; EZCLIP_LIB=tools/bin/var_lnpacked/libezclip_hip.so timeout 120 python tools/lnbwd_bench.py;
  EZCLIP_LIB=tools/bin/var_lnpacked/libezclip_hip.so timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "layernorm" 2>&1 | tail -2 ) > gpurun_out/lnb_$TAG.log 2>&1
cat gpurun_out/lnb_$TAG.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_$TAG.md > /dev/null 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/proft_$TAG -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --workload bf16_b1024_train > $R/gpurun_out/proft_$TAG.log 2>&1
cd $R
DB=$(find /tmp/proft_$TAG -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_train_$TAG.md > /dev/null 2>&1
head -14 gpurun_out/kernel_stats_$TAG.md; head -16 gpurun_out/kernel_stats_train_$TAG.md
