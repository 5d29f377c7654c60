#!/bin/bash
# Round 4, call 20: the default line once more after bench.py collects the previous workloads' garbage before building the next model
mkdir -p gpurun_out; export TMPDIR=/tmp
EZCLIP_NO_CANARY=1 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_r4t.json 2> gpurun_out/bench_r4t.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r4t.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k, v in d["also"].items(): print(k, v.get("value"), v.get("ms_per_step"))
PY
