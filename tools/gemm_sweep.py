#!/usr/bin/env python
"""Per-shape throughput of the NT GEMM variants on the path's real shapes (GPU).
usage: python tools/gemm_sweep.py [bf16|fp32] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easynlp_amd import lib as L  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
td = torch.bfloat16 if dtype == "bf16" else torch.float32
lib = L.load()
Mv, Mt = B * 197, B * 64
SHAPES = [("vit.qkv", Mv, 2304, 768, 0, False), ("vit.out+res", Mv, 768, 768, 0, True),
          ("vit.fc+qgelu", Mv, 3072, 768, 1, False), ("vit.proj+res", Mv, 768, 3072, 0, True),
          ("bert.qkvo+res", Mt, 768, 768, 0, True), ("bert.ffn1+gelu", Mt, 3072, 768, 2, False),
          ("bert.ffn2+res", Mt, 768, 3072, 0, True), ("patch", B * 196, 768, 768, 0, False)]
dev = "cuda"
print("dtype", dtype, "batch", B)
for name, M, N, K, act, res in SHAPES:
    a = torch.randn(M, K, device=dev, dtype=torch.float32).to(td)
    w = (torch.randn(N, K, device=dev, dtype=torch.float32) * K ** -0.5).to(td)
    bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev, dtype=torch.float32).to(td) if res else None
    outs = {}
    line = "%-16s M=%7d N=%5d K=%5d :" % (name, M, N, K)
    for v in (0, 1):
        L.check(lib.ezclip_debug_set(0, v))
        c = L.op_gemm_nt(a, w, bias=bias, residual=r, act=act)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            L.op_gemm_nt(a, w, bias=bias, residual=r, act=act, out=c)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        outs[v] = c.float()
        line += "  v%d %7.1f TF (%.3f ms)" % (v, 2.0 * M * N * K / ms / 1e9, ms)
    line += "  maxdiff %.3g" % float((outs[0] - outs[1]).abs().max())
    print(line, flush=True)
L.check(lib.ezclip_debug_set(0, -1))
