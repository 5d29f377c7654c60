"""Sorts first in the GPU suite: is the box itself healthy?

A child process that imports nothing but torch allocates, multiplies and synchronises on cuda:0. libezclip_hip.so is
not loaded by it (nor by this module), so a failure here cannot be a fault of this repository's kernels: the message says
so in the first line the driver's `gpu_test_tail` shows.  The second test is the first to load the library at all and
launches its smallest kernel, so "box healthy, library loads, one launch completes" is on the record before any heavy
test starts."""
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

from __graft_entry__ import BOX_FAULT, run_canary  # noqa: E402,F401  (one definition, shared with smoke())


def test_00_box_is_healthy_pure_torch_child_process():
    ok, text = run_canary()
    print(text)
    assert ok, text


def test_01_library_loads_and_launches_one_kernel():
    import torch

    from easynlp_amd import lib as L
    lib = L.load()
    assert b"gfx950" in lib.ezclip_version()
    x = torch.randn(64, 768, device="cuda")
    g = torch.randn(768, device="cuda")
    b = torch.randn(768, device="cuda")
    y = L.op_layernorm(x, g, b, 1e-5)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double().cpu(), (768,), g.double().cpu(), b.double().cpu(), 1e-5)
    assert float((y.double().cpu() - ref).abs().max()) < 1e-5
    print("LIBEZCLIP_LOADED", L.LIB_PATH)
