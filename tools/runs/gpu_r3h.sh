#!/bin/bash
# Round 3: packing-metadata kernel with eight sentences in flight per wave, launched on the text stream: tests, kernel time, bench.
TAG=${1:-r3h}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_canary_gpu.py tests/test_pack_meta_gpu.py tests/test_model_gpu.py tests/test_hf_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "canary or pack or golden" 2>&1 | tail -20 > gpurun_out/pytest_focus_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_focus_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR\|Error\|assert " gpurun_out/pytest_focus_$TAG.log | head -20
cd /tmp && EZCLIP_NO_CANARY=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${TAG}_kernel_stats.md > /dev/null 2>&1
grep "pack_meta\|nce_tile" gpurun_out/${TAG}_kernel_stats.md | cut -c1-160
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["value"], d["ms_per_step"], d["loss"], d.get("model_mfma_frac"))'
B="python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 5"
{ for v in 1 2 3; do EZCLIP_NO_CANARY=1 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"; done; EZCLIP_NO_CANARY=1 timeout 300 $B --workload bf16_b1024_train 2>/dev/null | tail -1 | python -c "$P"; } > gpurun_out/ab_$TAG.log 2>&1
cat gpurun_out/ab_$TAG.log
