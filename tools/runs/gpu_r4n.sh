#!/bin/bash
# Round 4, call 14: ten-slot LDS ring, DMA eight half-tiles ahead (-DEZ_RING10=1) against the product build: check, then A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4n
LD_LIBRARY_PATH=tools/bin/var_ring10 NT_SHAPES=14 timeout 300 tools/bin/gemm_bench 1024 3 0,2 2>&1 | grep -v "^batch" | cut -c1-200 > gpurun_out/gb_check_$T.log; cat gpurun_out/gb_check_$T.log
EZCLIP_LIB=tools/bin/var_ring10/libezclip_hip.so timeout 600 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py -k "canary or gemm or linear or ln" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_gemm_$T.log
for v in base ring10 base ring10; do
  echo "== $v: gemm_bench 1024 300 2"; LD_LIBRARY_PATH=tools/bin/var_$v NT_SHAPES=14 timeout 300 tools/bin/gemm_bench 1024 300 2 2>&1 | grep -v "^batch"
done > gpurun_out/gb_ring_ab_$T.log 2>&1
cat gpurun_out/gb_ring_ab_$T.log
for v in base ring10 base ring10; do
  EZCLIP_LIB=tools/bin/var_$v/libezclip_hip.so EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 150 > gpurun_out/bench_ring_${v}_$T.json 2> gpurun_out/bench_ring_${v}_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_ring_${v}_$T.json").read().strip().splitlines()[-1])
print("$v fwd", d["value"], d["ms_per_step"], d["model_mfma_frac"], d["roofline"]["frac"], d["sustained"]["ms_per_step"], d["sustained"]["telemetry"]["shader_clock_mhz_mean"], d["sustained"]["telemetry"]["socket_power_w_mean"])
PY
done 2>&1 | tee gpurun_out/bench_ring_ab_$T.log
