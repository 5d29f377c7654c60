#!/usr/bin/env python
"""Same-box A/B of switches of the short attention forward kernel (ezclip_debug_set): op-level timing on the towers' shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easynlp_amd import lib as L  # noqa: E402

lib = L.load()
key = int(sys.argv[1]) if len(sys.argv) > 1 else 9
ON, OFF = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 0)
for name, B, Lq, H in (("vit-b/16", 1024, 197, 12), ("vit-l/14", 512, 257, 16), ("len 200", 1024, 200, 12), ("len 224", 1024, 224, 12)):
    qkv = (torch.randn(B * Lq, 3 * H * 64, device="cuda") * 0.5).bfloat16()
    outs, times = {}, {0: [], 1: []}
    for rep in range(3):
        for v in (1, 0):
            L.check(lib.ezclip_debug_set(key, ON if v else OFF))
            for _ in range(5):
                c = L.op_attention(qkv, B, Lq, H)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                c = L.op_attention(qkv, B, Lq, H)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 40)
            outs[v] = c
    L.check(lib.ezclip_debug_set(key, 3 if key == 9 else 1))
    print("%-9s switch on: %s ms   off: %s ms   outputs equal: %s" % (
        name, " ".join("%.4f" % t for t in times[1]), " ".join("%.4f" % t for t in times[0]), bool(torch.equal(outs[0], outs[1]))) + "  max |diff| %.3g" % float((outs[0].float() - outs[1].float()).abs().max()), flush=True)
