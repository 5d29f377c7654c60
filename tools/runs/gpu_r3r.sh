#!/bin/bash
# Round 3: clock and power of the chip while the headline workload runs (evidence for "power-limited")
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/power_r3r.log
{ echo "# idle"; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction)" | head -8; } > $OUT
EZCLIP_NO_CANARY=1 python bench.py --no-also --no-cpu-baseline --steps 600 --warmup 5 > gpurun_out/bench_power_r3r.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24; do
  { echo "# sample $i (t = $((6 + i)) s)"; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Package Power|sclk|Sensor junction" | sed -e 's/=*//' | tr '\n' ' '; echo; } >> $OUT
  sleep 1
done
wait $BP
tail -1 gpurun_out/bench_power_r3r.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("bench:", d["value"], d["ms_per_step"])' >> $OUT
cat $OUT | tail -32
