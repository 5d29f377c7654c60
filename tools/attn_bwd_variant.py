#!/usr/bin/env python
"""One number per process: the fused two-pass attention backward (with the q / k / v bias gradients, as the towers call it) on the ViT-B/16
shape, for A/B runs across variant builds (EZCLIP_LIB=tools/bin/var_<name>/libezclip_hip.so) -- ms per launch + a checksum of dq | dk | dv."""
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easynlp_amd import lib as L  # noqa: E402

lib = L.load()
B, Lq, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), (int(sys.argv[2]) if len(sys.argv) > 2 else 197), (int(sys.argv[3]) if len(sys.argv) > 3 else 12)
D = H * 64
g = torch.Generator(device="cuda").manual_seed(3)
qkv = (torch.randn(B * Lq, 3 * D, device="cuda", generator=g) * 0.5).bfloat16()
dctx = torch.randn(B * Lq, D, device="cuda", generator=g).bfloat16()
ctx, lse = L.op_attention(qkv, B, Lq, H, key_bias=None, want_lse=True)
L.check(lib.ezclip_debug_set(11, 0))
base = qkv.data_ptr()
dqkv = torch.zeros_like(qkv)
db = torch.zeros(3 * D, device="cuda")
scratch = torch.empty(B * 3 * D, dtype=torch.float32, device="cuda")


def run():
    dbase, bb = dqkv.data_ptr(), db.data_ptr()
    L.check(lib.ezclip_op_attention_bwd_bias(base, base + D * 2, base + 2 * D * 2, 3 * D, ctx.data_ptr(), dctx.data_ptr(), D, None, lse.data_ptr(),
                                             dbase, dbase + D * 2, dbase + 2 * D * 2, bb, bb + 4 * D, bb + 8 * D, scratch.data_ptr(), B, Lq, H,
                                             L.DTYPE_BF16, None, L.stream_ptr()))


for _ in range(5):
    run()
torch.cuda.synchronize()
times = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) / 20)
db.zero_()
run()
torch.cuda.synchronize()
crc = zlib.crc32(dqkv.view(torch.int16).cpu().numpy().tobytes())
print("%-28s B %d L %d H %d: %s ms  crc32(dqkv) %08x  |db| %.6g" % (os.path.basename(os.path.dirname(os.environ.get("EZCLIP_LIB", "in-tree/x"))), B, Lq, H,
                                                                   " ".join("%.4f" % t for t in times), crc, float(db.double().norm())), flush=True)
