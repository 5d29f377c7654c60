#!/usr/bin/env python
"""Throughput of the GPU image pre-processing (ezclip_preprocess_images) with the decoded batch already resident in HBM,
beside the reference's PIL pipeline on one host core.

    python tools/preprocess_bench.py [n_images] [width] [height]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from easynlp_amd import lib as L  # noqa: E402
from oracle import preprocess_oracle as P  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 375
    lib = L.load()
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    per = (img.size + 15) // 16 * 16
    desc = (L.EzclipImageDesc * n)()
    for i in range(n):
        desc[i].offset, desc[i].width, desc[i].height = i * per, w, h
    one = torch.zeros(per, dtype=torch.uint8)
    one[:img.size] = torch.from_numpy(img.reshape(-1))
    packed = one.repeat(n).cuda()
    packed = torch.cat([packed, torch.zeros(16, dtype=torch.uint8, device="cuda")])
    ws = L.alloc_bytes(lib.ezclip_preprocess_workspace_bytes(desc, n, 224, 224), "cuda")
    out = torch.empty((n, 3, 224, 224), dtype=torch.float32, device="cuda")
    m3, s3 = (C.c_float * 3)(*L.CLIP_MEAN), (C.c_float * 3)(*L.CLIP_STD)

    def run():
        L.check(lib.ezclip_preprocess_images(L.ptr(packed), desc, n, 224, 224, m3, s3, L.ptr(out), L.ptr(ws), ws.numel(),
                                             L.stream_ptr()))
    run()
    torch.cuda.synchronize()
    ref = P.reference_pipeline_pil(__import__("PIL.Image").Image.fromarray(img))
    assert np.array_equal(out[n - 1].cpu().numpy().view(np.uint32), ref.view(np.uint32))
    t0 = time.perf_counter()
    iters = 10
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    src, dst = n * img.size, n * 3 * 224 * 224 * 4
    # device-only time (host window tables excluded): events around the two kernels via the library's profile hooks
    L.check(lib.ezclip_profile_begin())
    run()
    torch.cuda.synchronize()
    ms, work, cnt = C.c_double(), C.c_double(), C.c_int()
    L.check(lib.ezclip_profile_end(2, C.byref(ms), C.byref(work), C.byref(cnt)))
    from PIL import Image
    pil = Image.fromarray(img)
    t1 = time.perf_counter()
    k = 0
    while time.perf_counter() - t1 < 3.0:
        P.reference_pipeline_pil(pil)
        k += 1
    cpu = k / (time.perf_counter() - t1)
    print("preprocess %d x (%d x %d): %.3f ms per call (host planning + upload + device) = %.0f images/s; device kernels %.3f ms = %.2f TB/s "
          "(src %.0f MB + out %.0f MB); PIL pipeline on one core: %.0f images/s"
          % (n, w, h, dt * 1e3, n / dt, ms.value, (src + dst) / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0, src / 1e6, dst / 1e6, cpu))


if __name__ == "__main__":
    main()
