#!/bin/bash
# Round 4, call 18: HIP stream priority of the text tower's stream (two-stream forward step): 0 (default) / -1 (high)
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4r
for v in 0 -1 0 -1; do
  EZCLIP_SIDE_PRIORITY=$v EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 150 > gpurun_out/bench_prio_$T.json 2> gpurun_out/bench_prio_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_prio_$T.json").read().strip().splitlines()[-1])
print("side priority $v fwd", d["value"], d["ms_per_step"], d["model_mfma_frac"], d["roofline"]["frac"], d["sustained"]["ms_per_step"], d["sustained"]["telemetry"]["shader_clock_mhz_mean"], d["sustained"]["telemetry"]["socket_power_w_mean"])
PY
done 2>&1 | tee gpurun_out/bench_prio_ab_$T.log
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" 2>&1 | tail -1 | tee -a gpurun_out/bench_prio_ab_$T.log
