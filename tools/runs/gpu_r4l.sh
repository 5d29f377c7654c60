#!/bin/bash
# Round 4, call 12: the direct (register -> global, v_permlane16_swap) epilogue of the 8-phase NT kernel against the LDS round-trip
# epilogue of the previous commit (tools/bin/var_base = libezclip_hip.so of 54bb785): correctness first, then A/B.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4l
DIFFMAP=1 timeout 300 tools/bin/gemm_bench 1024 3 0,2 2>&1 | grep -v "^batch\|attn" | head -60 > gpurun_out/gb_check_$T.log; head -40 gpurun_out/gb_check_$T.log
timeout 900 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py -k "canary or gemm or linear or ln" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_gemm_$T.log
timeout 900 python -m pytest -x -q -m gpu tests/test_bench_regime_gpu.py 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_regime_$T.log
for rep in 1 2; do for v in base new; do
  L=tools/bin/var_base; [ $v = new ] && L=easynlp_amd/csrc
  echo "== $v (rep $rep): gemm_bench 1024 300 2"; LD_LIBRARY_PATH=$L NT_SHAPES=14 timeout 300 tools/bin/gemm_bench 1024 300 2 2>&1 | grep -v "^batch"
done; done > gpurun_out/gb_epi_ab_$T.log 2>&1
cat gpurun_out/gb_epi_ab_$T.log
for v in base new base new; do
  L=tools/bin/var_base/libezclip_hip.so; [ $v = new ] && L=easynlp_amd/csrc/libezclip_hip.so
  EZCLIP_LIB=$L EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 150 > gpurun_out/bench_epi_${v}_$T.json 2> gpurun_out/bench_epi_${v}_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_epi_${v}_$T.json").read().strip().splitlines()[-1])
print("$v fwd", d["value"], d["ms_per_step"], d["model_mfma_frac"], d["roofline"]["frac"], d["sustained"]["ms_per_step"], d["sustained"]["telemetry"]["shader_clock_mhz_mean"], d["sustained"]["telemetry"]["socket_power_w_mean"])
PY
done 2>&1 | tee gpurun_out/bench_epi_ab_$T.log
