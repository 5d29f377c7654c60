#!/bin/bash
# Round 5, closing pass on the final tree: the complete GPU suite, smoke, the default bench line, tools/rn_bench.py and the kernel trace of the
# ModifiedResNet-50 training step.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5v}; R=$(pwd)
{ timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=6 -p no:cacheprovider 2>&1 | tail -40; } > gpurun_out/pytest_$T.log
grep -n "passed\|failed" gpurun_out/pytest_$T.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$T.log | head
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke_$T.log
timeout 900 python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$T.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["model_mfma_frac"], "roofline", d["roofline"]["frac"], "clock", d.get("clock_mhz_timed_steps"), "padded", d.get("value_padded_text"), d.get("model_mfma_frac_padded_text"), "sustained clock", d["sustained"]["telemetry"].get("shader_clock_mhz_mean"))
for k, v in d["also"].items():
    print("  %-36s %9.1f pairs/s %8.2f ms  frac %s" % (k, v.get("value", -1), v.get("ms_per_step", -1), v.get("model_mfma_frac")))
PY
timeout 300 python tools/rn_bench.py 2>&1 | grep RN50 | tee gpurun_out/rn_bench_$T.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rn_$T -o rn -- python $R/tools/rn_train_profile.py > $R/gpurun_out/prof_rn_$T.log 2>&1
cd $R
DB=$(find /tmp/prof_rn_$T -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${T}_rn_train_kernel_stats.md "gemm" > /dev/null 2>&1
head -24 gpurun_out/${T}_rn_train_kernel_stats.md | cut -c1-170
