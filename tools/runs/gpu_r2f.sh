#!/bin/bash
TAG=${1:-r2f}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 2>&1 | tail -60 > gpurun_out/pytest_$TAG.log
grep -n "passed\|failed" gpurun_out/pytest_$TAG.log | tail -2; grep -n "FAILED" gpurun_out/pytest_$TAG.log | head
