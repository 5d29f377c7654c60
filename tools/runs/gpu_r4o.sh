#!/bin/bash
# Round 4, call 15: fused retrieval ranks (tests), then the spill fix A/B: var_base = library of 071a0eb (six NT instantiations spill 2-6 registers)
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4o
timeout 900 python -m pytest -x -q -m gpu tests/test_00_canary_gpu.py tests/test_ops_gpu.py -k "canary or recall or similarity or gemm" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -20 | tee gpurun_out/pytest_recall_$T.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/recall_fused_timing_$T.log
import time, torch
from easynlp_amd.appzoo.clip.evaluator import recall_ranks
for n, e in ((5000, 512), (30000, 512), (50000, 768)):
    g = torch.Generator().manual_seed(1)
    t = torch.nn.functional.normalize(torch.randn(n, e, generator=g), dim=-1).cuda()
    v = torch.nn.functional.normalize(t + 0.8 * torch.randn(n, e, generator=g).cuda(), dim=-1)
    out = {}
    for name, kw in (("materialise (4096-query blocks)", dict(materialise=True)), ("fused t2i", {}), ("fused both directions", dict(both_directions=True))):
        recall_ranks(t, v, **kw); torch.cuda.synchronize()
        t0 = time.time(); r = recall_ranks(t, v, **kw); torch.cuda.synchronize(); dt = time.time() - t0
        out[name] = r
        print("n=%6d e=%4d  %-32s %8.2f ms  (%.1f TF f32)" % (n, e, name, dt * 1e3, 2.0 * n * n * e / dt / 1e12))
    assert torch.equal(out["materialise (4096-query blocks)"], out["fused t2i"]) and torch.equal(out["fused t2i"], out["fused both directions"][0])
PY
for v in base new base new; do
  L=tools/bin/var_base; [ $v = new ] && L=easynlp_amd/csrc
  echo "== $v: gemm_bench 1024 300 2"; LD_LIBRARY_PATH=$L NT_SHAPES=14 timeout 300 tools/bin/gemm_bench 1024 300 2 2>&1 | grep -v "^batch\|attn\|wgrad\|ln.fold"
done > gpurun_out/gb_spill_ab_$T.log 2>&1
cat gpurun_out/gb_spill_ab_$T.log
for v in base new base new; do
  L=tools/bin/var_base/libezclip_hip.so; [ $v = new ] && L=easynlp_amd/csrc/libezclip_hip.so
  EZCLIP_LIB=$L EZCLIP_NO_CANARY=1 timeout 600 python bench.py --no-also --no-cpu-baseline --steps 20 --sustained-steps 150 > gpurun_out/bench_spill_${v}_$T.json 2> gpurun_out/bench_spill_${v}_$T.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_spill_${v}_$T.json").read().strip().splitlines()[-1])
print("$v fwd", d["value"], d["ms_per_step"], d["model_mfma_frac"], d["roofline"]["frac"], d["sustained"]["ms_per_step"], d["sustained"]["telemetry"]["shader_clock_mhz_mean"], d["sustained"]["telemetry"]["socket_power_w_mean"])
PY
done 2>&1 | tee gpurun_out/bench_spill_ab_$T.log
