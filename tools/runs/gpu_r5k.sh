#!/bin/bash
# Round 5, call 11: the fused attention backward with the OVERLAPPED prologue (K, V + own rows first; pass A runs while the Q / dO images
# land) against the load-everything-first build (var_bwdnoovl): tests, op level (stand-alone and between GEMM bursts), training step;
# and the default line's also-workloads with the clock beside every number.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5k}
echo "== attention tests"; timeout 900 python -m pytest -x -q -m gpu tests/test_ops_gpu.py tests/test_dropout.py tests/test_openclip_gpu.py tests/test_model_gpu.py -k "attention or dropout or causal or backward" 2>&1 | tail -3 | tee gpurun_out/pytest_attn_bwd_ovl_$T.log
for v in old new old new; do
  L=easynlp_amd/csrc; [ $v = old ] && L=tools/bin/var_bwdnoovl
  echo "== $v"; LD_LIBRARY_PATH=$L ONLY_ATTN=1 ATTN_PROBE=1 timeout 300 tools/bin/gemm_bench 1024 50 2 2>&1 | grep "bwd" | grep -v bert
done 2>&1 | tee gpurun_out/attn_bwd_ovl_ab_$T.log
for v in old new old new; do
  L=easynlp_amd/csrc/libezclip_hip.so; [ $v = old ] && L=tools/bin/var_bwdnoovl/libezclip_hip.so
  EZCLIP_LIB=$L EZCLIP_NO_CANARY=1 timeout 600 python bench.py --workload bf16_b1024_train --no-also --no-cpu-baseline --steps 12 --warmup 3 --sustained-steps 0 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v train', d['value'], d['ms_per_step'], d.get('clock_mhz_timed_steps'))"
done 2>&1 | tee -a gpurun_out/attn_bwd_ovl_ab_$T.log
echo "== also-workloads with clocks"
EZCLIP_NO_CANARY=1 timeout 900 python bench.py --no-cpu-baseline --sustained-steps 0 --steps 10 --also bf16_b1024_fwd_loss_padded_text,bf16_b1024_train,bf16_b1024_train_padded_text,bf16_b1024_train_opt,bf16_b1024_fwd_loss_autograd,bf16_b1024_train_autograd 2> /dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("   %-36s %8.2f ms   clock %s MHz  power %s W" % ("headline", d["ms_per_step"], d.get("clock_mhz_timed_steps"), d.get("power_w_timed_steps")))
for k, v in d["also"].items(): print("   %-36s %8.2f ms   clock %s MHz  power %s W" % (k, v["ms_per_step"], v.get("clock_mhz_timed_steps"), v.get("power_w_timed_steps")))' | tee gpurun_out/also_clocks_$T.log
