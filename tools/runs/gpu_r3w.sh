#!/bin/bash
# Round 3: the complete suite + smoke + headline on the final tree (late additions: config-5-shape and 1024-pair tests; bench exactly as the driver runs it).
TAG=${1:-r3j}; HEAD=${2:-unknown}
mkdir -p gpurun_out
export TMPDIR=/tmp
{ echo "# pytest tests -m gpu on HEAD $HEAD ($(date -u +%FT%TZ))";
  timeout 1700 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 -p no:cacheprovider 2>&1 | tail -60; } > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/pytest_$TAG.log | head
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -3 gpurun_out/smoke_$TAG.log
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.json | cut -c1-260
