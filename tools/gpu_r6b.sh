#!/bin/bash
# Round 6, call B: RN training parity edges (second pass), bench contract + recall leg, attention-backward phase trace, GEMM de-phase A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_resnet_train_gpu.py -q -s -p no:cacheprovider 2>&1 | grep -v "^E    +\|^E   *where" | tail -150 > gpurun_out/r6b_pytest_rn.log
grep -n "passed\|failed\|^FAILED\|EXPLICIT_IM2COL" gpurun_out/r6b_pytest_rn.log | tail -12
timeout 1500 python -m pytest tests/test_zz_bench_contract_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r6b_pytest_bench.log
tail -5 gpurun_out/r6b_pytest_bench.log
EZCLIP_LIB=tools/bin/var_attntrace/libezclip_hip.so timeout 300 python tools/attn_bwd_trace.py > gpurun_out/r6b_attn_bwd_trace.log 2>&1
cat gpurun_out/r6b_attn_bwd_trace.log | tail -40
EZCLIP_LIB=tools/bin/var_attntrace/libezclip_hip.so timeout 300 python tools/attn_bwd_trace.py 512 257 16 > gpurun_out/r6b_attn_bwd_trace_l257.log 2>&1
tail -28 gpurun_out/r6b_attn_bwd_trace_l257.log
bash tools/gpu_variants.sh r6b_dephase 4 dephase1 dephase2 > /dev/null 2>&1
grep -n "==\|TF\|ms" gpurun_out/variants_r6b_dephase.log | head -60
