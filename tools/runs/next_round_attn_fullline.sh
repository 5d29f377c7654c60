#!/bin/bash
# READY FOR THE NEXT ROUND (not run yet).  Attention forward: output rows written as full 128-byte lines through the dead K image
# (tools/experiments/attn_fwd_fullline_store.patch; the product writes every line in four 32-byte pieces).  Before calling this on the box:
#     git apply tools/experiments/attn_fwd_fullline_store.patch tools/experiments/attn_bwd_wide_store.patch
#     python tools/build_variants.py "fullline@attention_short.hip+attention_short_bwd.hip:-DEZ_ATTN_FULLLINE_STORE -DEZ_ATTN_BWD_WIDE_STORE"
#   (both patches are inert without their flags.  The second one widens the backward kernels' dq / dk / dv stores from 8 to 16 bytes per lane --
#    v_permlane32_swap pairs, as the forward epilogue already does: every 128-byte line in four pieces instead of eight.)
# Then: correctness of the variant library (attention tests), op-level timing, and the forward step A/B.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5b}
EZCLIP_LIB=tools/bin/var_fullline/libezclip_hip.so timeout 900 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py tests/test_dropout.py tests/test_openclip_gpu.py -k "canary or attention or dropout or causal or backward" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head | tee gpurun_out/pytest_fullline_$T.log
for v in product fullline product fullline; do
  L=easynlp_amd/csrc; [ $v = fullline ] && L=tools/bin/var_fullline
  echo "== $v"; LD_LIBRARY_PATH=$L ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 50 2 2>&1 | grep "attn"
done 2>&1 | tee gpurun_out/attn_fullline_ab_$T.log
for v in product fullline product fullline; do
  L=easynlp_amd/csrc/libezclip_hip.so; [ $v = fullline ] && L=tools/bin/var_fullline/libezclip_hip.so
  EZCLIP_LIB=$L EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 150 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v fwd', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'])"
done 2>&1 | tee -a gpurun_out/attn_fullline_ab_$T.log
for v in product fullline product fullline; do
  L=easynlp_amd/csrc/libezclip_hip.so; [ $v = fullline ] && L=tools/bin/var_fullline/libezclip_hip.so
  EZCLIP_LIB=$L EZCLIP_NO_CANARY=1 timeout 600 python bench.py --workload bf16_b1024_train --no-also --no-cpu-baseline --steps 12 --warmup 3 --sustained-steps 0 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v train', d['value'], d['ms_per_step'])"
done 2>&1 | tee -a gpurun_out/attn_fullline_ab_$T.log
