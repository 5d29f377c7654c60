#!/usr/bin/env python
"""Calibration only (never on the product path): what the vendor libraries reach on this box for the shapes of the path --
torch.matmul (hipBLASLt / rocBLAS behind it) for the big bf16 GEMMs and torch SDPA for the attention shapes -- next to
tools/bin/gemm_bench's numbers for the hand-written kernels.  Prints TFLOP/s; used for DESIGN.md section 6."""
import os
import time

import torch

dev = "cuda"
OPSCALE = float(os.environ.get("OPERAND_SCALE", "1"))     # 0: all-zero operands (no datapath toggling: the chip keeps its full clock)
ITERS = int(os.environ.get("ITERS", "20"))


def bench(fn, iters=ITERS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


for name, M, N, K in (("vit.qkv", 201728, 2304, 768), ("vit.out", 201728, 768, 768), ("vit.fc", 201728, 3072, 768),
                      ("vit.proj", 201728, 768, 3072), ("bert.ffn1", 65536, 3072, 768)):
    a = ((torch.rand(M, K, device=dev) * 2 - 1) * OPSCALE).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05 * OPSCALE).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    t = bench(lambda: torch.nn.functional.linear(a, w, bias))
    print("%-10s M=%d N=%d K=%d : torch linear (bias) %.1f TF (%.3f ms)" % (name, M, N, K, 2.0 * M * N * K / t / 1e12, t * 1e3))
    del a, w

for name, B, H, Lq in (() if os.environ.get("NO_ATTN") else (("attn.vit", 1024, 12, 197), ("attn.bert", 1024, 12, 64))):
    q = torch.randn(B, H, Lq, 64, device=dev).bfloat16()
    k, v = torch.randn_like(q), torch.randn_like(q)
    try:
        t = bench(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
        print("%-10s B=%d H=%d L=%d : torch SDPA fwd %.3f ms (%.0f TF)" % (name, B, H, Lq, t * 1e3, 4.0 * B * H * Lq * Lq * 64 / t / 1e12))
    except Exception as e:      # noqa: BLE001
        print(name, "SDPA failed:", e)
