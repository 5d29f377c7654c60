#!/bin/bash
# Round 5, call 3: the corrected parity tests (GradScaler, bf16 gradient-error tables with the activation-rounding floor); the
# ModifiedResNet training tower's per-parameter gradient errors (tools/rn_train_diag.py); the autograd-path bisect prepared in round 4;
# PMC passes over the attention kernels (full-line forward) and over the GEMM with the super-column tile order (traffic).
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5c}
echo "== parity tests"
timeout 600 python -m pytest -q -m gpu tests/test_amp_and_grad_error_gpu.py > gpurun_out/pytest_amp_graderr_full_$T.log 2>&1; grep -E "^E  |passed|failed|^FAILED" gpurun_out/pytest_amp_graderr_full_$T.log | cut -c1-300 | head -20
head -8 gpurun_out/r5_bf16_grad_error_vitb16_bertbase_b4_l64.md | cut -c1-300
echo "== RN training diag"
timeout 600 python tools/rn_train_diag.py 2>&1 | grep -v "^\[ezclip\]" | tail -80 | tee gpurun_out/rn_train_diag2_$T.log
echo "== autograd bisect"
bash tools/runs/autograd_bisect.sh $T 2>&1 | tail -20
echo "== attention PMC"
bash tools/pmc_attn.sh $T > /dev/null 2>&1; head -60 gpurun_out/pmc_attn_$T.md | cut -c1-200
echo "== GEMM PMC (super-column order)"
bash tools/pmc_gemm.sh $T > /dev/null 2>&1
python tools/pmc_summary.py $T gpurun_out > /dev/null 2>&1; head -30 gpurun_out/${T}_gemm_pmc.md | cut -c1-220; ls gpurun_out | grep "$T"
