#!/usr/bin/env python
"""bench.py's cpu_baseline leg timed both ways on THIS host: the imported reference CLIPApp (`kind: reference`, needs the
/root/reference checkout) and the CPU oracle (`kind: port`, what the GPU box can run).  Prints the ratio port / reference that
bench.py quotes next to a `kind: port` baseline (PORT_VS_REFERENCE)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from oracle import ref_harness as R  # noqa: E402

assert R.reference_available(), "no reference checkout here"
sec = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
ref = B.cpu_baseline(sec)
assert ref["kind"] == "reference", ref
avail = R.reference_available
R.reference_available = lambda: False
try:
    port = B.cpu_baseline(sec)
finally:
    R.reference_available = avail
print(json.dumps({"reference": ref, "port": port,
                  "port_vs_reference": {"fwd": round(port["value"] / ref["value"], 3),
                                        "train": round(port["train_value"] / ref["train_value"], 3)}}, indent=1))
