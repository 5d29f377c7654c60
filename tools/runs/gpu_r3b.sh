#!/bin/bash
# Round 3, second call: the new kernels (tiled InfoNCE, packing metadata, polynomial GELU, attention options) -- focused tests first,
# then the whole suite, GEMM shapes, contrastive-step timing, the bench line with rotating batches.
TAG=${1:-r3b}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_canary_gpu.py tests/test_pack_meta_gpu.py tests/test_ops_gpu.py tests/test_dropout.py -m gpu -q --maxfail=10 -p no:cacheprovider \
   -k "canary or pack or infonce or attention or gemm_nt" 2>&1 | tail -60 > gpurun_out/pytest_focus_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_focus_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR\|Error" gpurun_out/pytest_focus_$TAG.log | head -20
timeout 300 python tools/nce_bench.py > gpurun_out/nce_$TAG.log 2>&1; cat gpurun_out/nce_$TAG.log | tail -12
timeout 300 tools/bin/gemm_bench 1024 10 2 2>&1 | grep -v "^batch" > gpurun_out/gb_$TAG.log; head -16 gpurun_out/gb_$TAG.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --durations=8 -p no:cacheprovider 2>&1 | tail -70 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$TAG.log | head -30
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.json | cut -c1-300; tail -3 gpurun_out/bench_$TAG.err
