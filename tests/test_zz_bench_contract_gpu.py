"""bench.py's one JSON line (the driver's contract) on a small batch: required keys, types, the roofline and cpu_baseline
objects.  Ordered last: written after the round's GPU minutes were spent (bench.py itself ran on hardware all round; the
`config` key rename is what this guards)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["unit"] == "pairs/s" and out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "bf16" and out["data"] == "synthetic" and out["value"] > 0 and out["ms_per_step"] > 0
    assert out["config"]["workload"] == "bf16_b1024_fwd_loss" and "model" not in out["config"] and out["config"]["pairs_per_gpu"] == 64
    roof = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0
    assert 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
