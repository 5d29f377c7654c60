#!/usr/bin/env python
"""The headline step with the inputs starting in HOST memory, as the reference's boundary hands them over (CLIPApp.forward moves
`pixel_values` / `input_ids` to the device itself: appzoo/clip/model.py:116-123; float32 pixels = 602 KB per pair, 617 MB per 1 024-pair
batch).  bench.py's `value` starts with the inputs resident in HBM; this prints the PCIe-inclusive rates beside it:
  A  pageable host tensors through CLIPApp.forward (what the reference's DataLoader without pin_memory gives)
  B  pinned host tensors through CLIPApp.forward (DataLoader(pin_memory=True): core/trainer.py builds it that way when CUDA is on)
  C  pinned host tensors, batch k+1 copied on a side stream while batch k is computed (non_blocking prefetch), device tensors to forward
  D  device-resident inputs (the bench's regime), same loop
  E  the host tensors of A / B behind easynlp_amd.appzoo.clip.DevicePrefetcher (the product's device loader: a background thread copies
     batch k+1 while batch k is computed), its batches handed to CLIPApp.forward
usage: pcie_inclusive.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
import bench                                   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["bf16_b1024_fwd_loss"])
app, name = bench.build_app(wl, dev)
app.eval()
B, S = wl["batch"], wl["seq"]
dev_batches = [bench.synth_batch(B, S, bench.VITB16_BERTBASE["vocab_size"], dev, seed=1000 + 97 * k) for k in range(4)]
pageable = [(p.cpu(), i.cpu()) for p, i in dev_batches]
pinned = [(p.pin_memory(), i.pin_memory()) for p, i in pageable]
mb = pageable[0][0].numel() * 4 / 1e6 + pageable[0][1].numel() * 8 / 1e6


def fwd(px, ids):
    with torch.no_grad():
        out = app({"pixel_values": px, "input_ids": ids})
        return app.compute_loss(out, [])["loss"]


def timed(label, body):
    for k in range(3):
        body(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        loss = body(k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("%-58s %8.2f ms/step  %7.0f pairs/s   (%.1f GB/s of input if all of it crossed PCIe)  loss %.4f"
          % (label, ms, B / ms * 1e3, mb / ms, float(loss)), flush=True)


timed("D device-resident inputs (bench regime)", lambda k: fwd(*dev_batches[k % 4]))
timed("A pageable host tensors -> CLIPApp.forward", lambda k: fwd(*pageable[k % 4]))
timed("B pinned host tensors -> CLIPApp.forward", lambda k: fwd(*pinned[k % 4]))

copy_stream = torch.cuda.Stream(device=dev)
slots = [None, None]


def prefetch(k):
    with torch.cuda.stream(copy_stream):
        p, i = pinned[k % 4]
        slots[k % 2] = (p.to(dev, non_blocking=True), i.to(dev, non_blocking=True), torch.cuda.Event())
        slots[k % 2][2].record(copy_stream)


def body_c(k):
    if slots[k % 2] is None:
        prefetch(k)
    px, ids, ev = slots[k % 2]
    torch.cuda.current_stream().wait_event(ev)
    prefetch(k + 1)                             # lands in the other slot while this batch is computed
    loss = fwd(px, ids)
    px.record_stream(torch.cuda.current_stream()); ids.record_stream(torch.cuda.current_stream())
    slots[k % 2] = None
    return loss


def run_c(k):
    return body_c(k)


slots[0] = slots[1] = None
timed("C pinned + prefetch of batch k+1 on a copy stream", run_c)
from easynlp_amd.appzoo.clip import DevicePrefetcher   # noqa: E402


def timed_loader(label, host):
    n = steps + 3
    it = iter(DevicePrefetcher(({"pixel_values": host[k % 4][0], "input_ids": host[k % 4][1], "label_ids": []} for k in range(n)), dev))
    for _ in range(3):
        b = next(it)
        fwd(b["pixel_values"], b["input_ids"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in it:
        loss = fwd(b["pixel_values"], b["input_ids"])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("%-58s %8.2f ms/step  %7.0f pairs/s   (%.1f GB/s of input if all of it crossed PCIe)  loss %.4f"
          % (label, ms, B / ms * 1e3, mb / ms, float(loss)), flush=True)


timed_loader("E pageable host tensors behind DevicePrefetcher", pageable)
timed_loader("E pinned host tensors behind DevicePrefetcher", pinned)
timed("D device-resident inputs (bench regime), again", lambda k: fwd(*dev_batches[k % 4]))
