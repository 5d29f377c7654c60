#!/bin/bash
# Round 4, call 1: (1) MFMA shape power probe (32x32x16 vs 16x16x32 under the socket cap), (2) the round's new parity tests,
# (3) vendor-vs-ours PMC passes, (4) the bench line with the `sustained` block (does the telemetry source work on the box?)
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4a
{ ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>&1 | head -40; } > gpurun_out/hwmon_$T.log
( for sc in 1 0; do
    tools/bin/mfma_power_probe 4 $sc &
    BP=$!; sleep 2.5
    for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | sed -e 's/=*//' | tr '\n' ' '; echo; sleep 2; done
    wait $BP
  done ) > gpurun_out/mfma_power_probe_$T.log 2>&1
tail -30 gpurun_out/mfma_power_probe_$T.log
timeout 1500 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py -k "canary or gelu_polynomial or contrastive_loss_one or 8phase" 2>&1 | tail -5 | tee gpurun_out/pytest_a_$T.log
timeout 1500 python -m pytest -x -q -m gpu tests/test_model_gpu.py tests/test_hf_gpu.py tests/test_resnet_gpu.py -k "large_text or from_config" 2>&1 | tail -8 | tee gpurun_out/pytest_b_$T.log
timeout 1500 python -m pytest -x -q -s -m gpu tests/test_bench_regime_gpu.py -k "headline or config5" 2>&1 | tail -15 | tee gpurun_out/pytest_c_$T.log
bash tools/pmc_vendor_vs_ours.sh $T > gpurun_out/pmcvo_$T.log 2>&1
python tools/pmc_vendor_vs_ours.py $T gpurun_out/vendor_vs_ours_pmc_$T.md > /dev/null 2>gpurun_out/pmcvo_summary_$T.err
EZCLIP_NO_CANARY=1 timeout 900 python bench.py --also bf16_b1024_fwd_loss_padded_text,bf16_hf_vitl14_large_b512_train --no-cpu-baseline --steps 20 > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
tail -c 3000 gpurun_out/bench_$T.json
