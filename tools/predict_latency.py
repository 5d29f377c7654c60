#!/usr/bin/env python
"""Serving-size batches (what CLIPPredictor / an online service sends): latency of one no-grad dual-encoder pass, ViT-B/16 +
BERT-base, bf16 -- eager launches against the captured hipGraphs (clip_hip_graphs=1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn  # noqa: E402
from easynlp_amd.appzoo.clip import CLIPApp  # noqa: E402

apps = {}
for name, flag in (("eager", 0), ("graph", 1)):
    apps[name] = CLIPApp.from_config(Bn.VITB16_BERTBASE, seed=1234, device="cuda", compute_dtype="bf16")
    apps[name].eval()
    apps[name].use_graphs = bool(flag)
print("| pairs | eager ms | graph ms | eager pairs/s | graph pairs/s |\n|---|---|---|---|---|")
for B in (1, 4, 16, 64):
    px, ids = Bn.synth_batch(B, 64, Bn.VITB16_BERTBASE["vocab_size"], torch.device("cuda"), seed=7)
    res = {}
    for name, app in apps.items():
        with torch.no_grad():
            for _ in range(5):
                out = app({"pixel_values": px, "input_ids": ids}, feat=True)
            torch.cuda.synchronize()
            n = 50
            t0 = time.perf_counter()
            for _ in range(n):
                out = app({"pixel_values": px, "input_ids": ids}, feat=True)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / n * 1e3
    print("| %d | %.3f | %.3f | %.0f | %.0f |" % (B, res["eager"], res["graph"], B / res["eager"] * 1e3, B / res["graph"] * 1e3), flush=True)
