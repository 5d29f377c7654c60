#!/bin/bash
# Round 3: steady-state (300-step, power-capped) A/B of the step-level switches: two streams, packed text, raster.
mkdir -p gpurun_out
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
B="python bench.py --no-also --no-cpu-baseline --steps 300 --warmup 20"
{ for rep in 1 2; do
    echo -n "default            : "; EZCLIP_NO_CANARY=1 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
    echo -n "EZCLIP_TWO_STREAMS=0: "; EZCLIP_NO_CANARY=1 EZCLIP_TWO_STREAMS=0 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
    echo -n "EZCLIP_RASTER_GM=4  : "; EZCLIP_NO_CANARY=1 EZCLIP_RASTER_GM=4 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
  done; } > gpurun_out/ab_steady_r3s.log 2>&1
cat gpurun_out/ab_steady_r3s.log
