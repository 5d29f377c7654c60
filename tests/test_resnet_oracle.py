"""oracle/resnet_oracle.py (the ModifiedResNet image tower, eval mode) pinned to the real reference: through the committed fixture
tests/golden/rn_tiny_b3.npz (generated from the reference by tools/make_golden_resnet.py) and, where the reference checkout is
importable, live -- also for a second shape the fixture does not cover.  CPU only; no HIP tower consumes this oracle yet."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ref_harness as R
from oracle import resnet_oracle as RO

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_matches_the_reference_fixture():
    z = np.load(os.path.join(HERE, "golden", "rn_tiny_b3.npz"))
    c = json.loads(bytes(z["meta"]).decode())
    sd = RO.make_state_dict(c["layers"], c["width"], c["output_dim"], c["resolution"], c["wseed"])
    taps = {}
    with torch.no_grad():
        out = RO.modified_resnet_forward(sd, c["layers"], c["width"], torch.from_numpy(z["pixels"]), taps)
    assert out.shape == (c["batch"], c["output_dim"])
    assert float((taps["stem"] - torch.from_numpy(z["stem"])).abs().max()) < 1e-5
    assert float((taps["layer4"] - torch.from_numpy(z["layer4"])).abs().max()) < 2e-5
    assert float((out - torch.from_numpy(z["image_features"])).abs().max()) < 2e-5


def test_parameter_table_and_sensitivity():
    """Every entry of the table is used: perturbing any one tensor moves the output (a restatement that skipped a BatchNorm
    statistic or a downsample branch would not notice)."""
    layers, width, e, res = (1, 1, 1, 1), 8, 16, 32
    sd = RO.make_state_dict(layers, width, e, res, 5)
    g = torch.Generator().manual_seed(1)
    px = torch.randn(2, 3, res, res, generator=g)
    with torch.no_grad():
        base = RO.modified_resnet_forward(sd, layers, width, px)
        for name in sd:
            if name.endswith("attnpool.k_proj.bias"):
                continue          # (a constant added to every key moves all scores of a query alike: softmax does not see it)
            sd2 = dict(sd)
            sd2[name] = sd[name] + (0.05 if name.endswith("running_var") else 0.1) * torch.ones_like(sd[name])
            moved = RO.modified_resnet_forward(sd2, layers, width, px)
            assert float((moved - base).abs().max()) > 1e-7, name


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
@pytest.mark.parametrize("layers,width,e,res,B", [((1, 2, 1, 1), 16, 24, 64, 3), ((2, 1, 2, 1), 8, 32, 96, 2)])
def test_oracle_matches_the_live_reference(layers, width, e, res, B):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import make_golden_resnet as G
    cfg = dict(layers=layers, width=width, output_dim=e, resolution=res)
    sd = RO.make_state_dict(layers, width, e, res, 11)
    m = G.reference_tower(cfg, sd)
    # names and shapes of the table == the reference module's state dict (minus the step counters)
    ref_sd = {("visual." + k): tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")}
    assert ref_sd == RO.param_shapes(layers, width, e, res)
    g = torch.Generator().manual_seed(4)
    px = torch.randn(B, 3, res, res, generator=g)
    with torch.no_grad():
        want = m(px)
        got = RO.modified_resnet_forward(sd, layers, width, px)
    assert float((got - want).abs().max()) < 3e-5 * max(1.0, float(want.abs().max()))
