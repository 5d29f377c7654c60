#!/usr/bin/env python
"""First-contact probe (round 4): does RCCL accept two ranks on ONE device, and do init / a collective / teardown
return or hang?  Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 tools/rccl_two_ranks_one_gpu.py
with NCCL_DEBUG=INFO.  Every step is bounded by a watchdog: a hang is reported, not waited out."""
import datetime
import os
import sys
import threading
import time

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])


def watchdog(seconds, what):
    def run():
        time.sleep(seconds)
        print("[rank %d] WATCHDOG: %s did not return within %d s -- exiting" % (rank, what, seconds), flush=True)
        os._exit(17)
    t = threading.Thread(target=run, daemon=True)
    t.start()


torch.cuda.set_device(0)                                        # BOTH ranks on cuda:0
print("[rank %d] init_process_group(nccl) on cuda:0 ..." % rank, flush=True)
watchdog(90, "init / all_reduce / destroy")
t0 = time.time()
try:
    dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=60), device_id=torch.device("cuda", 0))
    print("[rank %d] init returned after %.1f s" % (rank, time.time() - t0), flush=True)
    x = torch.full((1 << 20,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("[rank %d] all_reduce -> %.1f (expected %.1f) after %.1f s" % (rank, float(x[0]), world * (world + 1) / 2, time.time() - t0), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("[rank %d] destroy_process_group returned: RCCL ACCEPTS two ranks per device here" % rank, flush=True)
except Exception as e:      # noqa: BLE001
    print("[rank %d] RCCL refused / failed after %.1f s: %s: %s" % (rank, time.time() - t0, type(e).__name__, str(e)[:600]), flush=True)
    sys.exit(3)
