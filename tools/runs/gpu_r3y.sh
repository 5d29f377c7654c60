#!/bin/bash
# Round 3: the PCIe-inclusive rate of the headline step (tools/pcie_inclusive.py); box canary first.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as e; print('canary:', e.run_canary())" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python -m pytest tests/test_prefetch.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python tools/pcie_inclusive.py 12 2>&1 | grep -v amdgpu.ids > gpurun_out/pcie_r3y.log
cat gpurun_out/pcie_r3y.log | tail -12
