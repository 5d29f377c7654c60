#!/bin/bash
# Round 3, first call: the COMPLETE -m gpu suite on the committed HEAD (canary first, small -> large), smoke(), the staged
# two-workgroup attention backward, the default bench line.
TAG=${1:-r3a}; HEAD=${2:-unknown}
mkdir -p gpurun_out
export TMPDIR=/tmp
{ echo "# pytest tests -m gpu on HEAD $HEAD ($(date -u +%FT%TZ))"; 
  timeout 1700 python -m pytest tests -m gpu -q --maxfail=12 --durations=15 -p no:cacheprovider 2>&1 | tail -120; } > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "FAILED\|ERROR" gpurun_out/pytest_$TAG.log | head -20
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -4 gpurun_out/smoke_$TAG.log
timeout 200 tools/bin/attn_bwd_2wg 1024 20 > gpurun_out/attn2wg_$TAG.log 2>&1; tail -12 gpurun_out/attn2wg_$TAG.log
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.json | cut -c1-400
