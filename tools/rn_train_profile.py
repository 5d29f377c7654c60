#!/usr/bin/env python
"""RN50 training step (bf16, 256 images) a few times: run under `rocprofv3 --kernel-trace --stats` for the per-kernel breakdown."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easynlp_amd import lib as L  # noqa: E402
from easynlp_amd.appzoo.clip.rn_tower import RnEngine  # noqa: E402
from oracle import resnet_oracle as RO  # noqa: E402  (weights only)

layers, width, e, res, B = (3, 4, 6, 3), 64, 1024, 224, 256
sd = RO.make_state_dict(layers, width, e, res, 1)
eng = RnEngine(layers, width, e, res, L.DTYPE_BF16)
dev = {n: sd[n].cuda() for n in eng.names}
eng.sync_train(dev)
px = torch.randn(B, 3, res, res, device="cuda")
probe = torch.randn(B, e, device="cuda")
grads = {n: torch.zeros(eng.shapes[n], dtype=torch.float32, device="cuda") for n in eng.names if not eng.is_statistic(n)}
for _ in range(int(os.environ.get("RN_PROFILE_STEPS", "6"))):
    out = eng.encode_image_train(px)
    eng.backward(out, probe, grads)
torch.cuda.synchronize()
print("done")
