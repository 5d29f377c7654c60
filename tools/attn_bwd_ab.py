#!/usr/bin/env python
"""Same-box A/B of the fused attention backward: the score-tile-once kernel of round 4 (ezclip_debug_set(11, 1), default) against the
two-pass kernel of rounds 2-3 (ezclip_debug_set(11, 0)) -- op-level timing on the towers' shapes (with the q / k / v bias gradients, as
the towers call it), interleaved rounds, and the difference of the outputs (summation order of dQ differs: not bit for bit)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easynlp_amd import lib as L  # noqa: E402

lib = L.load()
for name, B, Lq, H, masked in (("vit-b/16", 1024, 197, 12, False), ("bert 64", 1024, 64, 12, True), ("bert 40", 1024, 40, 12, True),
                               ("len 256", 512, 256, 12, False), ("len 128", 1024, 128, 12, False)):
    D = H * 64
    qkv = (torch.randn(B * Lq, 3 * D, device="cuda") * 0.5).bfloat16()
    dctx = torch.randn(B * Lq, D, device="cuda").bfloat16()
    kb = None
    if masked:
        lens = torch.randint(8, Lq + 1, (B,), device="cuda")
        kb = torch.where(torch.arange(Lq, device="cuda")[None, :] < lens[:, None], 0.0, -10000.0).reshape(-1).contiguous()
    ctx, lse = L.op_attention(qkv, B, Lq, H, key_bias=kb, want_lse=True)
    esz = 2
    base = qkv.data_ptr()
    outs, times = {}, {0: [], 1: []}

    def run(dqkv, db, scratch):
        dbase, bb = dqkv.data_ptr(), db.data_ptr()
        L.check(lib.ezclip_op_attention_bwd_bias(base, base + D * esz, base + 2 * D * esz, 3 * D, ctx.data_ptr(), dctx.data_ptr(), D,
                                                 L.ptr(kb), lse.data_ptr(), dbase, dbase + D * esz, dbase + 2 * D * esz,
                                                 bb, bb + 4 * D, bb + 8 * D, scratch.data_ptr(), B, Lq, H, L.DTYPE_BF16, None, L.stream_ptr()))

    scratch = torch.empty(B * 3 * D, dtype=torch.float32, device="cuda")
    for rep in range(3):
        for v in (1, 0):
            L.check(lib.ezclip_debug_set(11, 2 if v else 0))
            dqkv = torch.zeros_like(qkv)
            db = torch.zeros(3 * D, device="cuda")
            for _ in range(3):
                run(dqkv, db, scratch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(dqkv, db, scratch)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 20)
            db.zero_()
            run(dqkv, db, scratch)
            torch.cuda.synchronize()
            outs[v] = (dqkv.float().clone(), db.clone())
    L.check(lib.ezclip_debug_set(11, 1))
    d = float((outs[0][0] - outs[1][0]).abs().max())
    s = float(outs[0][0].abs().max())
    dbd = float((outs[0][1] - outs[1][1]).abs().max())
    print("%-9s once: %s ms   two-pass: %s ms   max |d dqkv| %.3g of %.3g   max |d bias grads| %.3g of %.3g" % (
        name, " ".join("%.4f" % t for t in times[1]), " ".join("%.4f" % t for t in times[0]), d, s, dbd, float(outs[0][1].abs().max())), flush=True)
