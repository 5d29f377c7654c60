#!/bin/bash
# Round 4, call 17: block-0 loads of the epilogue one K-tile early for the residual + row-stat and the act'(U) + erf-GELU instantiations too
# (fragment addresses recomputed per tile free the registers): base = ad2.. product, fa = recomputed addresses only, early = + early loads
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4q
for v in base fa early base fa early; do
  EZCLIP_LIB=tools/bin/var_$v/libezclip_hip.so EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 150 > gpurun_out/bench_${v}_$T.json 2> gpurun_out/bench_${v}_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_${v}_$T.json").read().strip().splitlines()[-1])
print("$v fwd", d["value"], d["ms_per_step"], d["model_mfma_frac"], d["roofline"]["frac"], d["sustained"]["ms_per_step"], d["sustained"]["telemetry"]["shader_clock_mhz_mean"], d["sustained"]["telemetry"]["socket_power_w_mean"])
PY
done 2>&1 | tee gpurun_out/bench_early_ab_$T.log
for v in base early base early; do
  EZCLIP_LIB=tools/bin/var_$v/libezclip_hip.so EZCLIP_NO_CANARY=1 timeout 600 python bench.py --workload bf16_b1024_train --no-also --no-cpu-baseline --steps 12 --warmup 3 --sustained-steps 0 > gpurun_out/bench_train_${v}_$T.json 2> gpurun_out/bench_train_${v}_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_train_${v}_$T.json").read().strip().splitlines()[-1])
print("$v train", d["value"], d["ms_per_step"], d["model_mfma_frac"], d["roofline"]["frac"])
PY
done 2>&1 | tee -a gpurun_out/bench_early_ab_$T.log
EZCLIP_LIB=tools/bin/var_early/libezclip_hip.so timeout 600 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py tests/test_bench_regime_gpu.py -k "canary or gemm or linear or ln or regime or headline" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_early_$T.log
