#!/bin/bash
# Round 5, first GPU call: (1) the two prepared attention store experiments (full-line forward stores, 16-byte backward stores) --
# tests on the variant library, op-level A/B, forward / train step A/B; (2) the ModifiedResNet training path (operator tests,
# tower-level fixture tests) -- verify-and-merge or delete; (3) the new parity tests (GradScaler / --use_amp, bf16 gradient error
# table).
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5a}
V=tools/bin/var_fullline/libezclip_hip.so
echo "== new parity tests"; timeout 900 python -m pytest -q -m gpu tests/test_00_canary_gpu.py tests/test_amp_and_grad_error_gpu.py --maxfail=10 2>&1 | tail -30 | tee gpurun_out/pytest_amp_graderr_$T.log
echo "== RN training"; timeout 900 python -m pytest -q -m gpu tests/test_resnet_train_ops_gpu.py tests/test_resnet_train_gpu.py tests/test_resnet_gpu.py --maxfail=30 2>&1 | tail -60 | tee gpurun_out/pytest_rn_train_$T.log
echo "== attention variant tests"; EZCLIP_LIB=$V timeout 900 python -m pytest -x -q -m gpu tests/test_ops_gpu.py tests/test_dropout.py tests/test_openclip_gpu.py -k "attention or dropout or causal or backward" 2>&1 | tail -15 | tee gpurun_out/pytest_fullline_$T.log
for v in product fullline product fullline; do
  L=easynlp_amd/csrc; [ $v = fullline ] && L=tools/bin/var_fullline
  echo "== $v"; LD_LIBRARY_PATH=$L ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 50 2 2>&1 | grep "attn"
done 2>&1 | tee gpurun_out/attn_fullline_ab_$T.log
for v in product fullline product fullline; do
  L=easynlp_amd/csrc/libezclip_hip.so; [ $v = fullline ] && L=$V
  EZCLIP_LIB=$L EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 150 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v fwd', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'])"
done 2>&1 | tee -a gpurun_out/attn_fullline_ab_$T.log
for v in product fullline product fullline; do
  L=easynlp_amd/csrc/libezclip_hip.so; [ $v = fullline ] && L=$V
  EZCLIP_LIB=$L EZCLIP_NO_CANARY=1 timeout 600 python bench.py --workload bf16_b1024_train --no-also --no-cpu-baseline --steps 12 --warmup 3 --sustained-steps 0 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v train', d['value'], d['ms_per_step'])"
done 2>&1 | tee -a gpurun_out/attn_fullline_ab_$T.log
