#!/bin/bash
# Round 6, closing pass on the final tree: the complete parity suite, smoke, the default bench line (every BASELINE config + the round's new
# fields), kernel traces of the forward and the training step, per-shape GEMM / attention / contrastive-step / ResNet timings.
TAG=${1:-r6}; HEAD=${2:-unknown}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
{ echo "# pytest tests -m gpu on HEAD $HEAD ($(date -u +%FT%TZ))";
  timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 -p no:cacheprovider 2>&1 | grep -v "^E    +\|^E   *where" | tail -100; } > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$TAG.log | head
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -3 gpurun_out/smoke_$TAG.log
timeout 1800 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.json | cut -c1-300
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "events", d.get("ms_per_step_hip_events"), "frac", d["model_mfma_frac"], "roofline", d["roofline"]["frac"], "avg us", d["roofline"]["avg_launch_us"], "clock", d.get("clock_mhz_timed_steps"), "padded", d.get("value_padded_text"), d.get("model_mfma_frac_padded_text"))
print("recall", d.get("recall_at_1"), d.get("recall_at_1_oracle"), d.get("recall_at_1_abs_diff"))
print("sustained", d["sustained"]["ms_per_step"], d["sustained"].get("ms_per_step_hip_events"), d["sustained"]["telemetry"], d["sustained"].get("model_mfma_frac_second_half"))
for k, v in d["also"].items():
    print("  %-36s %9.1f pairs/s %8.2f ms  host %7.2f  frac %s  clock %s MHz %s" % (k, v.get("value", -1), v.get("ms_per_step", -1), v.get("host_ms_per_step", -1), v.get("model_mfma_frac"), v.get("clock_mhz_timed_steps"), (v.get("sustained") or {}).get("model_mfma_frac_second_half", "")))
print(d["cpu_baseline"])
PY
for wl in fwd train; do
  W=""; [ $wl = train ] && W="--workload bf16_b1024_train"
  cd /tmp && EZCLIP_NO_CANARY=1 EZCLIP_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${wl}_$TAG -o bench -- python $R/bench.py --steps 20 --warmup 3 --sustained-steps 0 --no-cpu-baseline --no-recall --no-also $W > $R/gpurun_out/prof_${wl}_$TAG.log 2>&1
  cd $R
  DB=$(find /tmp/prof_${wl}_$TAG -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/${TAG}_${wl}_kernel_stats.md "gemm_(nt|tn)(_8p)?_kernel" > /dev/null 2>&1
  head -12 gpurun_out/${TAG}_${wl}_kernel_stats.md | cut -c1-160; tail -9 gpurun_out/${TAG}_${wl}_kernel_stats.md
  grep '^{"metric"' gpurun_out/prof_${wl}_$TAG.log > gpurun_out/${TAG}_${wl}_bench_line_under_rocprof.json
  python -c 'import sys,json; d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]; print("bench line of this run: ms_per_step", d["ms_per_step"], "roofline avg_launch_us", r["avg_launch_us"], "launches_per_step", r["launches_per_step"], "frac", r["frac"])' gpurun_out/${TAG}_${wl}_bench_line_under_rocprof.json
done
ATTN_PROBE=1 ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 20 2 2>&1 | grep attn > gpurun_out/gb_attn_$TAG.log; cat gpurun_out/gb_attn_$TAG.log
timeout 300 tools/bin/gemm_bench 1024 10 2 2>&1 | grep -v "^batch" > gpurun_out/gb_$TAG.log; head -22 gpurun_out/gb_$TAG.log
timeout 300 python tools/nce_bench.py > gpurun_out/nce_$TAG.log 2>&1; tail -9 gpurun_out/nce_$TAG.log
timeout 300 python tools/rn_bench.py > gpurun_out/rn_bench_$TAG.log 2>&1; tail -6 gpurun_out/rn_bench_$TAG.log
