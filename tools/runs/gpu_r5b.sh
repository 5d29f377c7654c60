#!/bin/bash
# Round 5, call 2: (1) full logs of the new parity tests (r5a kept only a tail); (2) the ModifiedResNet training tower tests one
# process each with EZCLIP_SYNC_LAUNCHES=2 (r5a: a GPU fault in the wider-shapes test -- which launch?); (3) the GEMM's direct epilogue
# with full-line stores (DPP row_ror:8 + ds_bpermute) against the LDS round trip (var_epi0): bit-identity, op-level and step A/B;
# (4) the super-column tile order (RASTER_GM = 100 + w) on the N >= 2304 products.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5b}
echo "== parity tests (full log in gpurun_out/pytest_amp_graderr_full_$T.log)"
timeout 600 python -m pytest -q -m gpu tests/test_amp_and_grad_error_gpu.py > gpurun_out/pytest_amp_graderr_full_$T.log 2>&1; grep -E "^E  |passed|failed|^FAILED" gpurun_out/pytest_amp_graderr_full_$T.log | cut -c1-400 | head -40
echo "== RN training tower, one process per test"
for t in "test_training_tower_against_the_reference_fixture" "test_training_tower_wider_shapes_against_the_oracle[layers0-64-128-64-4]" "test_training_tower_wider_shapes_against_the_oracle[layers1-32-64-96-3]" "test_clipapp_trains_the_resnet_tower_when_asked"; do
  echo "-- $t"; EZCLIP_SYNC_LAUNCHES=2 timeout 300 python -m pytest -x -q -m gpu "tests/test_resnet_train_gpu.py::$t" > gpurun_out/rn_$T.tmp 2>&1
  grep -E "^\[ezclip\] launch" gpurun_out/rn_$T.tmp | tail -3; grep -vE "^\[ezclip\] launch" gpurun_out/rn_$T.tmp | grep -E "^E  |passed|failed|Error|error|fault|Abort" | cut -c1-500 | head -12
done 2>&1 | tee gpurun_out/rn_train_diag_$T.log
echo "== GEMM epilogue: epi0 (LDS round trip) vs new (direct, full-line)"
for v in epi0 new epi0 new; do
  L=easynlp_amd/csrc; [ $v = epi0 ] && L=tools/bin/var_epi0
  echo "== $v"; LD_LIBRARY_PATH=$L timeout 300 tools/bin/gemm_bench 1024 200 0,2 2>&1 | grep -E "TF|ln.fold" | grep -v attn
done 2>&1 | tee gpurun_out/gemm_epi_ab_$T.log
echo "== GEMM tests on the new library"
timeout 900 python -m pytest -q -m gpu tests/test_ops_gpu.py tests/test_bench_regime_gpu.py tests/test_model_gpu.py -x 2>&1 | tail -5 | tee gpurun_out/pytest_gemm_epi_$T.log
echo "== tile order (new library): n-fastest vs super-columns of w tile columns"
for g in 0 103 104 106 0 103 106; do
  echo "RASTER_GM=$g"; RASTER_GM=$g NT_SHAPES=3 LD_LIBRARY_PATH=easynlp_amd/csrc timeout 300 tools/bin/gemm_bench 1024 200 2 2>&1 | grep -E "qkv|fc"
done 2>&1 | tee gpurun_out/gemm_supercolumn_$T.log
echo "== forward step"
for v in epi0 new new103 new106 epi0 new new103 new106; do
  L=easynlp_amd/csrc/libezclip_hip.so; [ $v = epi0 ] && L=tools/bin/var_epi0/libezclip_hip.so
  G=0; [ $v = new103 ] && G=103; [ $v = new106 ] && G=106
  EZCLIP_RASTER_GM=$G EZCLIP_LIB=$L EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 150 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v fwd', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d.get('model_mfma_frac'), d['roofline']['frac'])"
done 2>&1 | tee gpurun_out/step_epi_ab_$T.log
