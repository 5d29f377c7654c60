P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], d["config"]["text_dropout"], d["value"], d["ms_per_step"], d["loss"], d.get("text_tower_rows"))'
B="python bench.py --no-also --no-cpu-baseline --steps 10 --warmup 3 --workload bf16_hf_vitl14_b512_train --text-dropout 0.1"
mkdir -p gpurun_out
{ EZCLIP_NO_CANARY=1 timeout 300 $B 2>&1 | tail -1 | python -c "$P"
  EZCLIP_NO_CANARY=1 EZCLIP_PACK_HF_DROPOUT=0 timeout 300 $B 2>&1 | tail -1 | python -c "$P"; } > gpurun_out/ab_hf_dropout.log 2>&1
cat gpurun_out/ab_hf_dropout.log
