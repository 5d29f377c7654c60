#!/bin/bash
# Round 6, call D: attention-backward compile-time variants (setprio / launch bounds / unroll), rg03 gradient table, selective GEMM de-phase step A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r6d_attn_bwd_variants.log; : > $OUT
for rep in 1 2; do
  timeout 120 python tools/attn_bwd_variant.py >> $OUT 2>&1
  for v in ab_prio ab_lb448 ab_lb448u2 ab_prio448; do EZCLIP_LIB=tools/bin/var_$v/libezclip_hip.so timeout 120 python tools/attn_bwd_variant.py >> $OUT 2>&1; done
done
grep " ms " $OUT
timeout 900 python -m pytest tests/test_amp_and_grad_error_gpu.py -q -p no:cacheprovider -k rg03 2>&1 | grep -v "^E    +\|^E   *where" | tail -12 | cut -c1-300
tail -3 gpurun_out/r6_bf16_grad_error_vitb16_bertbase_rg03_b4_l64.md | cut -c1-400
for rep in 1 2 3; do for d in 0 -1; do EZCLIP_NO_CANARY=1 EZCLIP_GEMM_DEPHASE=$d timeout 300 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline --no-recall --sustained-steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dephase $d rep $rep: value', d['value'], 'ms', d['ms_per_step'], 'events', d['ms_per_step_hip_events'], 'sustained', d['sustained']['ms_per_step'], 'clk', d['sustained']['telemetry'].get('shader_clock_mhz_mean'), 'roofline', d['roofline']['frac'])"; done; done 2>&1 | tee gpurun_out/r6d_bench_dephase_selective_ab.log
for rep in 1 2; do for d in 0 -1; do EZCLIP_NO_CANARY=1 EZCLIP_GEMM_DEPHASE=$d timeout 300 python bench.py --workload bf16_b1024_fwd_loss_padded_text --steps 20 --warmup 5 --no-also --no-cpu-baseline --no-recall --sustained-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('padded dephase $d rep $rep: value', d['value'], 'ms', d['ms_per_step'], 'frac', d['model_mfma_frac'])"; done; done 2>&1 | tee -a gpurun_out/r6d_bench_dephase_selective_ab.log
