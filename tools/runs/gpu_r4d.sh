#!/bin/bash
# Round 4, call 4: the activation as a template argument of the 8-phase NT kernel (straight-line epilogues) -- correctness, then
# same-box A/B against the run-time-activation build (tools/bin/var_actrt); distributed tests with their summary captured.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4d
timeout 900 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py -k "canary or gemm" 2>&1 | grep -E "passed|failed|Error" | tee gpurun_out/pytest_ops_$T.log
timeout 1200 python -m pytest -x -q -m gpu tests/test_bench_regime_gpu.py -k "not config5 and not directional and not headline" 2>&1 | grep -E "passed|failed|Error" | tee gpurun_out/pytest_regime_$T.log
timeout 900 python -m pytest -x -q -m gpu tests/test_engine_state_gpu.py tests/test_zz_two_ranks_one_gpu.py tests/test_resnet_gpu.py 2>&1 | grep -E "passed|failed|Error|assert" | tee gpurun_out/pytest_dist_$T.log
OUT=gpurun_out/gemm_act_template_ab_$T.log; : > $OUT
for v in base actrt base actrt; do
  echo "## variant=$v (OPERAND_SCALE=1, 1500 launches per shape)" >> $OUT
  if [ $v = base ]; then LP=easynlp_amd/csrc; else LP=tools/bin/var_$v; fi
  LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH NT_SHAPES=12 timeout 300 tools/bin/gemm_bench 1024 1500 2 2>&1 | grep -v "^batch" | sed -e 's/maxdiff.*//' >> $OUT
done
cat $OUT
for v in base actrt; do
  if [ $v = base ]; then unset EZCLIP_LIB; else export EZCLIP_LIB=$PWD/tools/bin/var_$v/libezclip_hip.so; fi
  EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 200 > gpurun_out/bench_${v}_$T.json 2> gpurun_out/bench_${v}_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_${v}_$T.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["model_mfma_frac"], d["roofline"]["frac"], d["sustained"]["ms_per_step_second_half"], d["sustained"]["telemetry"])
PY
done 2>&1 | tee gpurun_out/bench_ab_$T.log
