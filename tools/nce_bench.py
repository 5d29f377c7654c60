#!/usr/bin/env python
"""Contrastive step on embeddings: the tiled kernels (csrc/nce.hip) against the materialising path of rounds 1-2, per call, at
the single-GPU shape (n = N = 1024) and the north star's exchange step (n = 1024 of N = 8192)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easynlp_amd import lib as L  # noqa: E402

lib = L.load()
dev = "cuda"
for n, N, off, e in ((1024, 1024, 0, 512), (1024, 8192, 3072, 512), (512, 4096, 1024, 768)):
    g = torch.Generator().manual_seed(0)
    t = torch.nn.functional.normalize(torch.randn(N, e, generator=g), dim=-1).to(dev)
    i = torch.nn.functional.normalize(t.cpu() + torch.randn(N, e, generator=g), dim=-1).to(dev)
    ls = torch.tensor(2.6593, device=dev)
    loss, dls = torch.empty((), device=dev), torch.empty((), device=dev)
    dT, dI = torch.empty(N, e, device=dev), torch.empty(N, e, device=dev)

    def run(kind, grads):
        a = (L.ptr(dT), L.ptr(dI), L.ptr(dls)) if grads else (None, None, None)
        if kind == "old":
            L.check(lib.ezclip_infonce_fused(L.ptr(t), L.ptr(i), n, N, off, e, L.ptr(ls), 1.0, L.ptr(loss), *a, L.ptr(ws), ws.numel(), L.stream_ptr()))
        else:
            L.check(lib.ezclip_infonce_tiled(L.ptr(t), L.ptr(i), n, N, off, e, L.ptr(ls), 1.0, 1 if kind == "split" else 0, L.ptr(loss), *a,
                                             L.ptr(ws), ws.numel(), L.stream_ptr()))
    for kind in ("old", "split", "bf16"):
        nbytes = lib.ezclip_infonce_workspace_bytes(n, N, e) if kind == "old" else lib.ezclip_infonce_tiled_workspace_bytes(n, N, e)
        ws = L.alloc_bytes(nbytes, dev)
        res = []
        for grads in (False, True):
            for _ in range(3):
                run(kind, grads)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(20):
                run(kind, grads)
            ev1.record()
            torch.cuda.synchronize()
            res.append(ev0.elapsed_time(ev1) / 20)
        print("n=%5d N=%5d e=%4d  %-5s  fwd %.3f ms  fwd+bwd %.3f ms  workspace %.1f MB  loss %.6f" % (n, N, e, kind, res[0], res[1], nbytes / 2**20, loss.item()), flush=True)
