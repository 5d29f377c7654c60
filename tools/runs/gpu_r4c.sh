#!/bin/bash
# Round 4, call 3: where the MI16 8-phase GEMM's sustained rate goes (ablation builds under the power cap) beside hipBLASLt on the same
# box; the weight-gradient kernel on 16x16x32 (tests + training bench A/B); event-log progress + two ranks on one GPU; RCCL first contact.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4c
timeout 900 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py -k "canary or gemm_tn or layernorm_bwd" 2>&1 | tail -3 | tee gpurun_out/pytest_tn_$T.log
timeout 1200 python -m pytest -x -q -m gpu tests/test_bench_regime_gpu.py -k "not config5 and not directional and not headline" 2>&1 | tail -3 | tee gpurun_out/pytest_regime_$T.log
timeout 1200 python -m pytest -x -q -m gpu tests/test_model_gpu.py -k "backward_matches or packed_text_tower_training" 2>&1 | tail -3 | tee gpurun_out/pytest_bwd_$T.log
timeout 900 python -m pytest -x -q -m gpu tests/test_engine_state_gpu.py tests/test_zz_two_ranks_one_gpu.py 2>&1 | tail -3 | tee gpurun_out/pytest_dist_$T.log
OUT=gpurun_out/gemm_ablations_$T.log; : > $OUT
for v in base noepi nostore noldsread nodma mainonly base; do
  echo "## variant=$v (OPERAND_SCALE=1, 1500 launches per shape)" >> $OUT
  if [ $v = base ]; then LP=easynlp_amd/csrc; else LP=tools/bin/var_$v; fi
  LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH NT_SHAPES=4 timeout 200 tools/bin/gemm_bench 1024 1500 2 2>&1 | grep -v "^batch" | sed -e 's/maxdiff.*//' >> $OUT
done
echo "## hipBLASLt (torch.nn.functional.linear, bias), 1500 launches per shape" >> $OUT
ITERS=1500 NO_ATTN=1 timeout 300 python tools/vendor_calibration.py >> $OUT 2>&1
cat $OUT
for v in base mi32; do
  if [ $v = base ]; then unset EZCLIP_LIB; else export EZCLIP_LIB=$PWD/tools/bin/var_$v/libezclip_hip.so; fi
  EZCLIP_NO_CANARY=1 timeout 600 python bench.py --workload bf16_b1024_train --no-also --no-cpu-baseline --steps 10 --sustained-steps 100 > gpurun_out/bench_train_${v}_$T.json 2> gpurun_out/bench_train_${v}_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_train_${v}_$T.json").read().strip().splitlines()[-1])
print("$v train", d["value"], d["ms_per_step"], d["model_mfma_frac"], d["roofline"]["frac"], d.get("time_share"), d.get("sustained"))
PY
done 2>&1 | tee gpurun_out/bench_train_ab_$T.log
unset EZCLIP_LIB
timeout 1500 python -m pytest -x -q -s -m gpu tests/test_bench_regime_gpu.py -k "config5 or headline" 2>&1 | tail -12 | tee gpurun_out/pytest_c5_$T.log
NCCL_DEBUG=INFO timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 tools/rccl_two_ranks_one_gpu.py > gpurun_out/rccl_two_ranks_one_gpu_$T.log 2>&1
tail -25 gpurun_out/rccl_two_ranks_one_gpu_$T.log
