#!/bin/bash
# READY FOR THE NEXT ROUND (not run yet).  The row-matrix kernels of the ModifiedResNet training path (BatchNorm training forward /
# backward, average-pool backward, im2col, input-gradient weight pack, weight-gradient unpack) + their operator-level GPU tests against
# the oracle's steps (tools/experiments/rn_train_kernels.patch; DESIGN.md 4.8 "Training the tower -- the plan").  Before calling this:
#     git apply tools/experiments/rn_train_kernels.patch && python -c "import __graft_entry__ as g; g.build()"
#     python -m pytest tests/test_cabi.py -q -m "not gpu"          (the new symbols are exported, nothing imports the oracle)
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r5c}
timeout 900 python -m pytest -q -m gpu tests/test_00_canary_gpu.py tests/test_resnet_train_ops_gpu.py --maxfail=20 2>&1 | tail -40 | tee gpurun_out/pytest_rn_train_ops_$T.log
# Step 2 (tools/experiments/rn_train_tower.patch INSTEAD of rn_train_kernels.patch: it contains the kernels and adds the tower's training
# forward / backward -- ezclip_rn_encode_image_train, ezclip_rn_backward, RnEngine.encode_image_train / backward -- and its engine-level tests)
if [ -f tests/test_resnet_train_gpu.py ]; then
  timeout 900 python -m pytest -q -m gpu tests/test_resnet_train_gpu.py tests/test_resnet_gpu.py --maxfail=20 2>&1 | tail -40 | tee gpurun_out/pytest_rn_train_tower_$T.log
fi
