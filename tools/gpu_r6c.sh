#!/bin/bash
# Round 6, call C: RN tests, the new full-depth bf16 fixture's table, attention-backward phase trace (fixed units), GEMM de-phase sweep + step A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_resnet_train_gpu.py -q -s -p no:cacheprovider 2>&1 | grep -v "^E    +\|^E   *where" | tail -60 > gpurun_out/r6c_pytest_rn.log
grep -n "passed\|failed\|^FAILED\|worst conv" gpurun_out/r6c_pytest_rn.log | tail -12
timeout 1200 python -m pytest tests/test_amp_and_grad_error_gpu.py -q -p no:cacheprovider 2>&1 | grep -v "^E    +\|^E   *where" | tail -40 > gpurun_out/r6c_pytest_graderr.log
tail -6 gpurun_out/r6c_pytest_graderr.log | cut -c1-300
head -8 gpurun_out/r6_bf16_grad_error_vitb16_bertbase_rg03_b4_l64.md | cut -c1-300
EZCLIP_LIB=tools/bin/var_attntrace/libezclip_hip.so timeout 300 python tools/attn_bwd_trace.py > gpurun_out/r6c_attn_bwd_trace.log 2>&1
tail -24 gpurun_out/r6c_attn_bwd_trace.log
EZCLIP_LIB=tools/bin/var_attntrace/libezclip_hip.so timeout 300 python tools/attn_bwd_trace.py 512 257 16 > gpurun_out/r6c_attn_bwd_trace_l257.log 2>&1
tail -26 gpurun_out/r6c_attn_bwd_trace_l257.log | head -12
OUT=gpurun_out/r6c_gemm_dephase.log; : > $OUT
for rep in 1 2; do for d in 0 1 113 126 206 306 402; do echo "== GEMM_DEPHASE=$d (rep $rep)" >> $OUT; NT_SHAPES=8 GEMM_DEPHASE=$d timeout 120 tools/bin/gemm_bench 1024 20 2 2>&1 | grep " v2 " >> $OUT; done; done
python - <<'PY'
import re, collections
t = collections.defaultdict(lambda: collections.defaultdict(list))
d = None
for line in open("gpurun_out/r6c_gemm_dephase.log"):
    m = re.match(r"== GEMM_DEPHASE=(\d+)", line)
    if m: d = int(m.group(1)); continue
    m = re.match(r"(\S+)\s+M=.*\(([\d.]+) ms\)", line)
    if m: t[m.group(1)][d].append(float(m.group(2)))
for shape, by in t.items():
    print("%-16s" % shape, "  ".join("%d: %s" % (k, "/".join("%.3f" % x for x in v)) for k, v in sorted(by.items())))
PY
for rep in 1 2; do for d in 0 1; do EZCLIP_NO_CANARY=1 EZCLIP_GEMM_DEPHASE=$d timeout 300 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline --no-recall --sustained-steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dephase $d rep $rep: value', d['value'], 'ms', d['ms_per_step'], 'events', d['ms_per_step_hip_events'], 'sustained', d['sustained']['ms_per_step'], 'clk', d['sustained']['telemetry'].get('shader_clock_mhz_mean'), 'roofline', d['roofline']['frac'])"; done; done 2>&1 | tee gpurun_out/r6c_bench_dephase_ab.log
