#!/usr/bin/env python
"""ModifiedResNet-50 image tower (csrc/resnet.hip), forward throughput at 224 x 224."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easynlp_amd import lib as L  # noqa: E402
from easynlp_amd.appzoo.clip.rn_tower import RnEngine  # noqa: E402
from oracle import resnet_oracle as RO  # noqa: E402  (weights only: test infrastructure, as in bench.py's baseline leg)

layers, width, e, res = (3, 4, 6, 3), 64, 1024, 224
sd = RO.make_state_dict(layers, width, e, res, 1)
for dtype in (() if os.environ.get("RN_BENCH_TRAIN_ONLY") else ("bf16", "fp32")):
    eng = RnEngine(layers, width, e, res, L.dtype_code(dtype))
    dev = {n: sd[n].cuda() for n in eng.names}
    eng.sync(dev)
    for B in ((256, 1024) if dtype == "bf16" else (64,)):
        px = torch.randn(B, 3, res, res, device="cuda")
        for _ in range(2):
            eng.encode_image(px)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            eng.encode_image(px)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("RN50 %s B=%4d: %.2f ms  %.0f images/s" % (dtype, B, ms, B / ms * 1e3), flush=True)

# training step of the tower (round 5): BatchNorm on batch statistics + the whole backward pass, 224 x 224
for dtype, B in ((("bf16", 128), ("bf16", 256)) if os.environ.get("RN_BENCH_BF16_ONLY") else (("bf16", 128), ("bf16", 256), ("fp32", 32))):
    eng = RnEngine(layers, width, e, res, L.dtype_code(dtype))
    dev = {n: sd[n].cuda() for n in eng.names}
    eng.sync_train(dev)
    px = torch.randn(B, 3, res, res, device="cuda")
    probe = torch.randn(B, e, device="cuda")
    grads = {n: torch.zeros(eng.shapes[n], dtype=torch.float32, device="cuda") for n in eng.names if not eng.is_statistic(n)}

    def step():
        out = eng.encode_image_train(px)
        eng.backward(out, probe, grads)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    finite = all(bool(torch.isfinite(g).all()) for g in grads.values())
    print("RN50 %s B=%4d TRAIN (fwd on batch statistics + backward): %.2f ms  %.0f images/s  grads finite: %s" % (dtype, B, ms, B / ms * 1e3, finite), flush=True)
