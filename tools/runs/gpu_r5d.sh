#!/bin/bash
# Round 5, call 4: the complete GPU suite on the tree with the ModifiedResNet training path (scratch sizing fixed) -- merge-or-delete
# decision --, the super-column width for nine tile columns (w = 5 vs 6), PMC passes over the attention kernels of the training step,
# and the default bench line.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5d}
echo "== RN training tests"
EZCLIP_SYNC_LAUNCHES=1 timeout 600 python -m pytest -q -m gpu tests/test_resnet_train_ops_gpu.py tests/test_resnet_train_gpu.py tests/test_resnet_gpu.py --maxfail=30 > gpurun_out/pytest_rn_train_$T.log 2>&1; grep -E "^E  |passed|failed|^FAILED|fault" gpurun_out/pytest_rn_train_$T.log | cut -c1-300 | head -30
timeout 300 python tools/rn_train_diag.py 2>&1 | grep -v "^\[ezclip\]" | tail -40 | tee gpurun_out/rn_train_diag_$T.log
echo "== complete suite"
{ timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 -p no:cacheprovider 2>&1 | tail -60; } > gpurun_out/pytest_$T.log
grep -n "passed\|failed" gpurun_out/pytest_$T.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$T.log | head
echo "== super-column width, 9 and 12 tile columns"
for g in 106 105 108 106 105; do
  echo "RASTER_GM=$g"; RASTER_GM=$g NT_SHAPES=3 LD_LIBRARY_PATH=easynlp_amd/csrc timeout 300 tools/bin/gemm_bench 1024 200 2 2>&1 | grep -E "qkv|fc"
done 2>&1 | tee gpurun_out/gemm_supercolumn_w_$T.log
echo "== attention PMC inside the training step"
bash tools/pmc_attn_train.sh $T; grep -E "^## |dur_us|WAIT_ANY|WAIT_INST_ANY|ACTIVE_INST_ANY|MFMA_BUSY|WAVE_CYCLES" gpurun_out/pmc_attn_$T.md | head -60
echo "== default bench line"
timeout 1500 python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r5d.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["model_mfma_frac"], "roofline", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "padded", d.get("value_padded_text"), d.get("model_mfma_frac_padded_text"))
print("sustained", d["sustained"]["ms_per_step"], d["sustained"]["telemetry"])
for k, v in d["also"].items():
    print("  %-36s %9.1f pairs/s %8.2f ms  frac %s" % (k, v.get("value", -1), v.get("ms_per_step", -1), v.get("model_mfma_frac")))
print(d["cpu_baseline"])
PY
