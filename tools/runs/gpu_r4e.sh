#!/bin/bash
# Round 4, call 5: is the LDS-DMA's cost (27 % in call 3) its issue / LDS-write side or the L2 / fabric traffic behind it?  (dmasame: every
# DMA aimed at the operand's first KiB); then the complete GPU suite + smoke + the default bench line on this tree (checkpoint).
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4e
OUT=gpurun_out/gemm_dma_ablation_$T.log; : > $OUT
for v in base dmasame nodma noepi base; do
  echo "## variant=$v (OPERAND_SCALE=1, 1500 launches per shape)" >> $OUT
  if [ $v = base ]; then LP=easynlp_amd/csrc; else LP=tools/bin/var_$v; fi
  LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH NT_SHAPES=4 timeout 200 tools/bin/gemm_bench 1024 1500 2 > gpurun_out/gb_tmp.log 2>&1 &
  BP=$!; sleep 3
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | sed -e 's/=*//' | tr '\n' ' ' >> $OUT; echo >> $OUT
  wait $BP
  grep -v "^batch" gpurun_out/gb_tmp.log | sed -e 's/maxdiff.*//' >> $OUT
done
cat $OUT
{ echo "# pytest tests -m gpu ($(date -u +%FT%TZ))";
  timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 -p no:cacheprovider 2>&1 | tail -40; } > gpurun_out/pytest_$T.log
grep -n "passed\|failed" gpurun_out/pytest_$T.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$T.log | head
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > gpurun_out/smoke_$T.log 2>&1; tail -2 gpurun_out/smoke_$T.log
timeout 1200 python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
tail -c 2500 gpurun_out/bench_$T.json
