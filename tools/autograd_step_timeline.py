#!/usr/bin/env python
"""Host-side and device-side timeline of the autograd training step (the path the reference Trainer drives: forward() + compute_loss() +
loss.backward(), bench.py workload bf16_b1024_train_autograd) next to the fused step (contrastive_step): where does the host wait, and is
the device ever waiting for the host?  Host clock at every phase boundary, one HIP event per boundary on the launch stream.
    python tools/autograd_step_timeline.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # > 1: the pair again and again in ONE process (does the process age?)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
burned = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("BURN", "0")))]      # streams someone else created first
for s_ in burned:
    with torch.cuda.stream(s_):
        torch.zeros(8, device=dev)
for path in [p_ for _ in range(rounds) for p_ in ("autograd", "fused")]:
    wl = dict(B.WORKLOADS["bf16_b1024_train_autograd" if path == "autograd" else "bf16_b1024_train"])
    app, _ = B.build_app(wl, dev)
    if os.environ.get("TWO_STREAMS") == "0":
        app.two_streams = False
    side = app._engine.side_stream(dev) if app.two_streams else None
    batches = [B.synth_batch(1024, 64, B.VITB16_BERTBASE["vocab_size"], dev, seed=1000 + 97 * k) for k in range(B.NBATCH)]
    params = list(app.parameters())
    rows = []

    def mark(tag, rec):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        rec.append((tag, time.perf_counter(), ev))

    for it in range(steps + 3):
        px, ids = batches[it % B.NBATCH]
        rec = []
        mark("start", rec)
        if path == "autograd":
            for p in params:
                p.grad = None
            mark("zero_grad", rec)
            out = app({"pixel_values": px, "input_ids": ids})
            mark("forward", rec)
            loss = app.compute_loss(out, [])["loss"]
            mark("loss", rec)
            loss.backward()
            mark("backward", rec)
        else:
            loss = app.contrastive_step(px, ids, process_group=False, backward=True, zero_grad=True)
            mark("step", rec)
        if it >= 3:
            rows.append(rec)
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    print("== %s: %d steps" % (path, steps))
    base_h, base_e = rows[0][0][1], rows[0][0][2]
    prev_end_dev = None
    for rec in (rows if rounds == 1 else []):
        h = ["%s +%.1f" % (tag, (t - rec[0][1]) * 1e3) for tag, t, _ in rec[1:]]
        d = ["%s +%.1f" % (tag, rec[0][2].elapsed_time(ev)) for tag, _, ev in rec[1:]]
        print("  host  step at %7.1f ms: %s" % ((rec[0][1] - base_h) * 1e3, "  ".join(h)))
        print("  device     at %7.1f ms: %s" % (base_e.elapsed_time(rec[0][2]), "  ".join(d)))
    n = len(rows)
    host_span = (rows[-1][0][1] - rows[0][0][1]) * 1e3 / (n - 1)
    dev_span = rows[0][0][2].elapsed_time(rows[-1][0][2]) / (n - 1)
    arena = app._engine.grad_arena("autograd", dev) if path == "autograd" else None
    inside = None
    if arena is not None:
        lo, hi = arena.flat.data_ptr(), arena.flat.data_ptr() + arena.flat.numel() * 4
        inside = sum(1 for p in params if p.grad is not None and lo <= p.grad.data_ptr() < hi)
    from easynlp_amd.appzoo.clip import model as M_
    print("  side stream %s (main %s); candidates rejected by the probe so far: %d" % (hex(side.cuda_stream) if side is not None else None,
          hex(torch.cuda.current_stream().cuda_stream), len(M_._SIDE_REJECTED)))
    print("  per step: host %.2f ms, device (start event to start event) %.2f ms; backward passes on a FRESH arena: %s of %d; .grad tensors inside the "
          "persistent arena: %s of %d" % (host_span, dev_span, getattr(app, "_arena_fresh_backwards", 0), steps + 3, inside,
                                          sum(1 for p in params if p.grad is not None)))
    del app, batches, params, rows, loss
    import gc
    gc.collect()
    torch.cuda.empty_cache()
