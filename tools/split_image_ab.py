#!/usr/bin/env python
"""A/B: the image tower of one 1024-image batch as ONE pass on one stream against k sub-batches on k streams in flight
together (each with a workspace of its own; the handle holds no per-call device state).  Question: do the sub-batches fill
each other's partial last rounds of GEMM tiles (2364 tiles on 256 CUs = 9.23 rounds for the N = 768 products)?
usage: split_image_ab.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
import bench                                   # noqa: E402
from easynlp_amd import lib as L               # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["bf16_b1024_fwd_loss"])
app, _ = bench.build_app(wl, dev)
eng = app._engine
eng.sync_params(dict(zip(eng.names, [app._params[n] for n in eng.names])), with_backward=False)
B = 1024
g = torch.Generator(device="cpu").manual_seed(0)
pix = torch.randn(B, 3, 224, 224, generator=g).to(dev)
main = torch.cuda.current_stream()


def run(k):
    n = B // k
    streams = [main] + [torch.cuda.Stream(device=dev) for _ in range(k - 1)]
    nbytes = eng.lib.ezclip_image_workspace_bytes(eng.handle, n, 0)
    wss = [L.alloc_bytes(nbytes, dev) for _ in range(k)]
    out = torch.empty(B, eng.embed_dim, dtype=torch.float32, device=dev)

    def once():
        for s in streams[1:]:
            s.wait_stream(main)
        for i, s in enumerate(streams):
            L.check(eng.lib.ezclip_encode_image(eng.handle, L.ptr(pix[i * n:(i + 1) * n]), n, L.ptr(out[i * n:(i + 1) * n]),
                                                L.ptr(wss[i]), wss[i].numel(), 0, L.stream_ptr(None if s is main else s)), "encode_image")
        for s in streams[1:]:
            main.wait_stream(s)
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


ref = None
for rep in range(2):
    for k in (1, 2, 4):
        ms, out = run(k)
        if ref is None:
            ref = out.clone()
        print("image tower, %d x %4d images on %d stream(s): %7.3f ms  (%6.0f images/s)  max |d emb| vs one pass %.2e"
              % (k, B // k, k, ms, B / ms * 1e3, (out - ref).abs().max().item()), flush=True)
