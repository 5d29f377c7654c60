#!/bin/bash
# Round 5, call 10: the attention forward as a PERSISTENT kernel (a workgroup walks (sample, head) items; 118 VGPRs) against one workgroup
# per item (ATTN_FWD_OPTS bit 3): tests, op level, forward step; and the autograd step's host time in the default workload order.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5j}
echo "== attention / dropout / packed tests"; timeout 900 python -m pytest -x -q -m gpu tests/test_ops_gpu.py tests/test_dropout.py tests/test_openclip_gpu.py tests/test_pack_meta_gpu.py tests/test_model_gpu.py -k "attention or dropout or causal or packed or pack or ragged or forward_matches" 2>&1 | tail -4 | tee gpurun_out/pytest_attn_persist_$T.log
for o in 3 11 3 11; do
  echo "== ATTN_FWD_OPTS=$o ($([ $o = 3 ] && echo persistent || echo one workgroup per item))"; ATTN_FWD_OPTS=$o LD_LIBRARY_PATH=easynlp_amd/csrc ONLY_ATTN=1 timeout 300 tools/bin/gemm_bench 1024 50 2 2>&1 | grep "attn" | grep -v bwd
done 2>&1 | tee gpurun_out/attn_persist_ab_$T.log
for o in 11 3 11 3; do
  EZCLIP_ATTN_FWD_OPTS=$o EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 150 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('opts $o fwd', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d.get('model_mfma_frac'), 'host', d.get('host_ms_per_step'))"
done 2>&1 | tee -a gpurun_out/attn_persist_ab_$T.log
echo "== host time per step, default order"
EZCLIP_NO_CANARY=1 timeout 900 python bench.py --no-cpu-baseline --sustained-steps 0 --steps 10 --also bf16_b1024_fwd_loss_padded_text,bf16_b1024_train,bf16_b1024_train_padded_text,bf16_b1024_train_opt,bf16_b1024_fwd_loss_autograd,bf16_b1024_train_autograd 2> /dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k, v in d["also"].items(): print("   %-36s %8.2f ms   host %8.2f ms" % (k, v["ms_per_step"], v.get("host_ms_per_step", -1)))' | tee gpurun_out/host_ms_$T.log
