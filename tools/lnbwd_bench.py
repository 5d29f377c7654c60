#!/usr/bin/env python
"""LayerNorm-backward micro-benchmark (ViT shape of the training workload): time + effective bandwidth.

A/B against the packed-staging experiment (rowops.hip, -DEZ_LNBWD_PACKED: four bf16 rows in flight per wave):
    python tools/build_variants.py lnpacked@rowops.hip:-DEZ_LNBWD_PACKED
    python tools/lnbwd_bench.py;  EZCLIP_LIB=tools/bin/var_lnpacked/libezclip_hip.so python tools/lnbwd_bench.py
(the results must be bit-identical: run tests/test_ops_gpu.py -k layernorm_bwd with the same EZCLIP_LIB)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easynlp_amd import lib as L  # noqa: E402

rows, D = 201728, 768
lib = L.load()
x = torch.randn(rows, D, device="cuda").bfloat16()
dy = torch.randn(rows, D, device="cuda").bfloat16()
g = torch.randn(D, device="cuda")
b = torch.randn(D, device="cuda")
y, mean, rstd = L.op_layernorm(x, g, b, 1e-5, want_stats=True)
dx = torch.empty_like(x)
dg = torch.zeros(D, device="cuda")
db = torch.zeros(D, device="cuda")


def run():
    L.check(lib.ezclip_op_layernorm_bwd(L.ptr(x), L.ptr(dy), L.ptr(g), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dg), L.ptr(db),
                                        rows, D, L.DTYPE_BF16, L.stream_ptr()))


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(n):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("ln_bwd %d x %d bf16: %.1f us, %.2f TB/s (x + dy read, dx written)" % (rows, D, dt * 1e6, 3 * rows * D * 2 / dt / 1e12))
for _ in range(3):
    yy = L.op_layernorm(x, g, b, 1e-5)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    yy = L.op_layernorm(x, g, b, 1e-5)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("ln_fwd %d x %d bf16: %.1f us, %.2f TB/s" % (rows, D, dt * 1e6, 2 * rows * D * 2 / dt / 1e12))
