#!/bin/bash
# Round 4, call 2: the bf16 GEMM kernels on v_mfma_f32_16x16x32_bf16 (EZ_MI16=1, base) against the 32x32x16 build (tools/bin/var_mi32):
# correctness (operator + regime tests, incl. bit equality 8-phase == 128x128), sustained gemm_bench A/B (1500 launches per shape,
# random operands), the headline bench A/B with its sustained leg; then the tests call 1 did not reach (config 5 at full depth).
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4b
timeout 900 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py 2>&1 | tail -4 | tee gpurun_out/pytest_ops_$T.log
timeout 900 python -m pytest -x -q -m gpu tests/test_bench_regime_gpu.py -k "not config5 and not directional" 2>&1 | tail -6 | tee gpurun_out/pytest_regime_$T.log
OUT=gpurun_out/gemm_mi16_ab_$T.log; : > $OUT
for v in base mi32 base mi32; do
  echo "## variant=$v (OPERAND_SCALE=1, 1500 launches per shape)" >> $OUT
  if [ $v = base ]; then LP=easynlp_amd/csrc; else LP=tools/bin/var_$v; fi
  LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH NT_SHAPES=7 timeout 200 tools/bin/gemm_bench 1024 1500 2 2>&1 | grep -v "^batch" >> $OUT
done
cat $OUT
for v in base mi32 base; do
  if [ $v = base ]; then unset EZCLIP_LIB; else export EZCLIP_LIB=$PWD/tools/bin/var_$v/libezclip_hip.so; fi
  EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 > gpurun_out/bench_${v}_$T.json 2> gpurun_out/bench_${v}_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_${v}_$T.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["model_mfma_frac"], d["roofline"]["frac"], d.get("sustained"))
PY
done 2>&1 | tee gpurun_out/bench_ab_$T.log
unset EZCLIP_LIB
timeout 1500 python -m pytest -x -q -s -m gpu tests/test_bench_regime_gpu.py -k "config5" 2>&1 | tail -12 | tee gpurun_out/pytest_c5_$T.log
