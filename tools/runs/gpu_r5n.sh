#!/bin/bash
# Round 5, call 14: the complete GPU suite a second time on a fresh box (flakiness check of the new tests), smoke, and the default bench line.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5n}
{ timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=6 -p no:cacheprovider 2>&1 | tail -40; } > gpurun_out/pytest_$T.log
grep -n "passed\|failed" gpurun_out/pytest_$T.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$T.log | head
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -2
timeout 1500 python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$T.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["model_mfma_frac"], "roofline", d["roofline"]["frac"], "clock", d.get("clock_mhz_timed_steps"), "padded", d.get("value_padded_text"), d.get("model_mfma_frac_padded_text"), "sustained clock", d["sustained"]["telemetry"].get("shader_clock_mhz_mean"))
for k, v in d["also"].items():
    print("  %-36s %9.1f pairs/s %8.2f ms  frac %s" % (k, v.get("value", -1), v.get("ms_per_step", -1), v.get("model_mfma_frac")))
PY
