#!/bin/bash
# Round 3: attention PMC summary of the final kernels; training steps under the reference's dropout 0.1 (chinese_clip and
# huggingface_clip flavour, packed vs padded text rows); the complete suite on the final HEAD.
TAG=${1:-r3m}; HEAD=${2:-unknown}
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/pmc_attn.sh $TAG > /dev/null 2>&1; head -40 gpurun_out/pmc_attn_$TAG.md | cut -c1-200
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["config"]["text_dropout"], d["value"], d["ms_per_step"], d["loss"], d.get("text_tower_rows"))'
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3"
{ echo "# training steps with the reference's dropout 0.1 (packed by default; EZCLIP_PACK_TEXT=0 / EZCLIP_PACK_HF_DROPOUT=0: padded rows)"
  EZCLIP_NO_CANARY=1 timeout 300 $B --workload bf16_b1024_train --text-dropout 0.1 2>/dev/null | tail -1 | python -c "$P"
  EZCLIP_NO_CANARY=1 EZCLIP_PACK_TEXT=0 timeout 300 $B --workload bf16_b1024_train --text-dropout 0.1 2>/dev/null | tail -1 | python -c "$P"
  EZCLIP_NO_CANARY=1 timeout 300 $B --workload bf16_hf_vitl14_b512_train --text-dropout 0.1 2>/dev/null | tail -1 | python -c "$P"
  EZCLIP_NO_CANARY=1 EZCLIP_PACK_HF_DROPOUT=0 timeout 300 $B --workload bf16_hf_vitl14_b512_train --text-dropout 0.1 2>/dev/null | tail -1 | python -c "$P"
  EZCLIP_NO_CANARY=1 timeout 300 $B --workload bf16_hf_vitl14_b512_train 2>/dev/null | tail -1 | python -c "$P"; } > gpurun_out/ab_dropout_$TAG.log 2>&1
cat gpurun_out/ab_dropout_$TAG.log
{ echo "# pytest tests -m gpu on HEAD $HEAD ($(date -u +%FT%TZ))";
  timeout 1700 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 -p no:cacheprovider 2>&1 | tail -60; } > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$TAG.log | head
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -2 gpurun_out/smoke_$TAG.log
