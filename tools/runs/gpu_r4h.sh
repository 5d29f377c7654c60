#!/bin/bash
# Round 4, call 8: cache-policy bits on the LDS-DMA loads (A streamed with nt / sc1, B with nt) -- sustained gemm_bench A/B; tile raster
# and two-stream A/B of the headline step in the sustained regime under the 16x16x32 kernels; the global-scope forward's logits.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4h
timeout 600 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_zz_global_scope_gpu.py 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head | tee gpurun_out/pytest_$T.log
OUT=gpurun_out/gemm_dma_policy_$T.log; : > $OUT
for v in base dma_a_nt dma_a_sc1 dma_b_nt dma_ab_nt base; do
  echo "## variant=$v (OPERAND_SCALE=1, 1500 launches per shape)" >> $OUT
  if [ $v = base ]; then LP=easynlp_amd/csrc; else LP=tools/bin/var_$v; fi
  LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH NT_SHAPES=4 timeout 200 tools/bin/gemm_bench 1024 1500 2 2>&1 | grep -v "^batch" | sed -e 's/maxdiff.*//' >> $OUT
done
cat $OUT
for cfg in "base" "EZCLIP_RASTER_GM=4" "EZCLIP_RASTER_GM=8" "EZCLIP_TWO_STREAMS=0" "base"; do
  if [ "$cfg" = base ]; then E=""; else E="$cfg"; fi
  env $E EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 200 > gpurun_out/bench_tmp_$T.json 2> gpurun_out/bench_tmp_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_tmp_$T.json").read().strip().splitlines()[-1])
print("$cfg", d["value"], d["ms_per_step"], "sustained", d["sustained"]["ms_per_step_second_half"], d["sustained"]["telemetry"].get("shader_clock_mhz_mean"), d["sustained"]["telemetry"].get("socket_power_w_mean"))
PY
done 2>&1 | tee gpurun_out/bench_raster_ab_$T.log
