#!/usr/bin/env python
"""How many blocking / copying HIP runtime calls does ONE step of bench.py issue?

Runs `bench.py --no-also --no-cpu-baseline --steps K` under `rocprofv3 --hip-trace --output-format csv` for two values of K and
divides the difference of the per-function call counts by the difference in steps: everything that belongs to start-up, warm-up,
the fences around the timed region and the roofline leg cancels.  A step that synchronises the stream or copies to the host
shows up as >= 1 per step.  usage: hip_api_per_step.py [workload] [out.md]"""
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload = sys.argv[1] if len(sys.argv) > 1 else "bf16_b1024_fwd_loss"
out_md = sys.argv[2] if len(sys.argv) > 2 else None
WATCH = ("hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize", "hipMemcpy", "hipMemcpyAsync", "hipMemcpyDtoH",
         "hipMemcpyDtoHAsync", "hipMemcpyWithStream", "hipStreamWaitEvent", "hipEventRecord", "hipLaunchKernel", "hipModuleLaunchKernel",
         "hipExtModuleLaunchKernel", "hipMemsetAsync", "hipMalloc", "hipFree", "hipHostMalloc")


def counts(steps):
    d = tempfile.mkdtemp(prefix="hiptrace_")
    cmd = ["rocprofv3", "--hip-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--no-also", "--no-cpu-baseline", "--warmup", "3", "--steps", str(steps), "--workload", workload]
    env = dict(os.environ, EZCLIP_NO_CANARY="1")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    files = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)
    if not files:
        sys.exit("no hip_api_trace.csv under %s\n%s" % (d, (r.stdout + r.stderr)[-2000:]))
    c = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            fn = row.get("Function") or row.get("Name") or ""
            c[fn] = c.get(fn, 0) + 1
    return c


k0, k1 = 10, 30
a, b = counts(k0), counts(k1)
lines = ["| HIP runtime call | calls per step | (%d steps) | (%d steps) |" % (k0, k1), "|---|---|---|---|"]
names = sorted(set(a) | set(b), key=lambda n: -(b.get(n, 0) - a.get(n, 0)))
for n in names:
    per = (b.get(n, 0) - a.get(n, 0)) / float(k1 - k0)
    if per != 0 or n in WATCH:
        lines.append("| `%s` | %.2f | %d | %d |" % (n, per, a.get(n, 0), b.get(n, 0)))
text = ("# HIP runtime calls per step of `bench.py --workload %s` (rocprofv3 --hip-trace, difference of a %d- and a %d-step run)\n\n"
        % (workload, k0, k1)) + "\n".join(lines) + "\n"
print(text)
if out_md:
    open(out_md, "w").write(text)
