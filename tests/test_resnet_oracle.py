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


def test_train_mode_oracle_matches_the_reference_fixture():
    """BatchNorm in training mode (batch statistics, running-statistic update) and the backward of the whole tower: features, every
    parameter gradient and every updated running statistic of oracle/resnet_oracle.py against the reference module's own
    (tests/golden/rn_tiny_train_b4.npz: nn.BatchNorm2d.train() + torch autograd on the REFERENCE ModifiedResNet)."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import make_golden_resnet as G
    z = np.load(os.path.join(HERE, "golden", "rn_tiny_train_b4.npz"))
    c = json.loads(bytes(z["meta"]).decode())
    sd = RO.make_state_dict(c["layers"], c["width"], c["output_dim"], c["resolution"], c["wseed"])
    out, grads, stats = RO.train_step_grads(sd, c["layers"], c["width"], torch.from_numpy(z["pixels"]), torch.from_numpy(z["probe"]))
    assert float((out - torch.from_numpy(z["image_features"])).abs().max()) < 3e-5
    names = [k for k in sd if not k.endswith(("running_mean", "running_var"))]
    assert set(grads) == set(names) and len(names) == 75
    seen = 0
    for k in names:
        g = grads[k]
        if ("grad:" + k) in z.files:
            ref = torch.from_numpy(z["grad:" + k])
            err = float((g - ref).norm())
            assert err <= 2e-4 * float(ref.norm()) + 1e-5, (k, err, float(ref.norm()))
        else:
            ref_norm = float(z["gnorm:" + k])
            samp = torch.from_numpy(z["gsamp:" + k])
            idx = torch.from_numpy(G.sample_index(k, g.numel()))
            assert abs(float(g.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-6, k
            assert float((g.reshape(-1)[idx] - samp).abs().max()) <= 2e-4 * ref_norm / (g.numel() ** 0.5) * 30 + 1e-6, k
        seen += 1
    assert seen == 75
    stat_names = [k for k in sd if k.endswith(("running_mean", "running_var"))]
    assert set(stats) == set(stat_names) and len(stat_names) == 44
    for k in stat_names:
        ref = torch.from_numpy(z["stat:" + k])
        assert float((stats[k] - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), k
        assert float((stats[k] - sd[k]).abs().max()) > 1e-4, k          # (the update moved it: the fixture is not the eval path)
    # eval mode on the same weights is a different function (running statistics instead of the batch's)
    with torch.no_grad():
        ev = RO.modified_resnet_forward(sd, c["layers"], c["width"], torch.from_numpy(z["pixels"]))
    assert float((ev - out).abs().max()) > 1e-3


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_train_mode_oracle_matches_the_live_reference():
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import make_golden_resnet as G
    cfg = dict(layers=(2, 1, 1, 2), width=8, output_dim=16, resolution=64)
    sd = RO.make_state_dict(cfg["layers"], cfg["width"], cfg["output_dim"], cfg["resolution"], 21)
    g = torch.Generator().manual_seed(8)
    px = torch.randn(5, 3, 64, 64, generator=g)
    probe = torch.randn(5, 16, generator=g)
    want_out, want_g, want_s = G.reference_train_step(cfg, sd, px, probe)
    out, grads, stats = RO.train_step_grads(sd, cfg["layers"], cfg["width"], px, probe)
    assert float((out - want_out).abs().max()) < 3e-5 * max(1.0, float(want_out.abs().max()))
    assert set(grads) == set(want_g) and set(stats) == set(want_s)
    for k in grads:        # (attnpool.k_proj.bias: a constant added to every key -- its exact gradient is 0, both sides hold rounding noise)
        assert float((grads[k] - want_g[k]).norm()) <= 2e-4 * float(want_g[k].norm()) + 1e-5, k
    for k in stats:
        assert float((stats[k] - want_s[k]).abs().max()) <= 1e-5 * max(1.0, float(want_s[k].abs().max())), k


@pytest.mark.parametrize("layers,width,e,res,B", [((1, 2, 1, 1), 16, 24, 64, 4), ((2, 1, 1, 2), 8, 16, 96, 3)])
def test_backward_by_steps_equals_autograd(layers, width, e, res, B):
    """The explicit backward formulas the HIP training path will implement (BatchNorm backward from two column moments, 3x3 input
    gradient as a convolution with flipped / transposed weights, weight gradient as im2col^T . dz, average-pool broadcast, one-query
    attention pool) against autograd of the restatement -- which the fixture above ties to the reference module."""
    sd = RO.make_state_dict(layers, width, e, res, 13)
    g = torch.Generator().manual_seed(2)
    px = torch.randn(B, 3, res, res, generator=g)
    probe = torch.randn(B, e, generator=g)
    out, grads, _ = RO.train_step_grads(sd, layers, width, px, probe)
    out2, grads2 = RO.train_step_grads_by_steps(sd, layers, width, px, probe)
    assert float((out - out2).abs().max()) < 1e-5 * max(1.0, float(out.abs().max()))
    assert set(grads) == set(grads2)
    for k in grads:
        assert grads2[k].shape == grads[k].shape, k
        assert float((grads2[k] - grads[k]).norm()) <= 3e-4 * float(grads[k].norm()) + 1e-5, (k, float((grads2[k] - grads[k]).norm()), float(grads[k].norm()))


@pytest.mark.parametrize("B,I,O,H", [(2, 8, 16, 6), (3, 24, 8, 5)])
def test_training_path_layout_conventions(B, I, O, H):
    """The index conventions the training path's packing kernels will use, in the HIP tower's data layout (NHWC rows, channels padded
    to 64, packed weights [Opad][9 * Cp], the implicit 3x3 convolution of gemm.hip): forward, input gradient (the same implicit
    convolution with tap-flipped, in/out-transposed weights) and weight gradient (dz^T . im2col(x), packed K order) against
    torch's convolution and the step-by-step backward above."""
    g = torch.Generator().manual_seed(B * 100 + I)
    x = torch.randn(B, I, H, H, generator=g, dtype=torch.float64)
    w = torch.randn(O, I, 3, 3, generator=g, dtype=torch.float64)
    dz = torch.randn(B, O, H, H, generator=g, dtype=torch.float64)
    cp, opad = 64, 64
    xa = RO.to_nhwc(x, cp)
    z = RO.implicit_conv3x3_nhwc(xa, RO.pack_conv3x3(w, cp, opad), B, H, H)
    want = torch.nn.functional.conv2d(x, w, padding=1)
    assert float((RO.from_nhwc(z, B, O, H, H) - want).abs().max()) < 1e-10
    assert float(z[:, O:].abs().max()) == 0.0                                # padded output channels stay zero
    dx_want, dw_want = RO._conv_bwd(x, w, dz, 1, 1)
    da = RO.to_nhwc(dz, opad)
    dx = RO.implicit_conv3x3_nhwc(da, RO.pack_conv3x3_dgrad(w, cp, opad), B, H, H)
    assert float((RO.from_nhwc(dx, B, I, H, H) - dx_want).abs().max()) < 1e-10
    assert float(dx[:, I:].abs().max()) == 0.0
    dwp = da.t() @ RO.im2col3x3_nhwc(xa, B, H, H)                             # one TN product over the pixels
    assert float((RO.unpack_wgrad3x3(dwp, O, I, cp) - dw_want).abs().max()) < 1e-9
    # autograd agrees with all of it
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    (torch.nn.functional.conv2d(xr, wr, padding=1) * dz).sum().backward()
    assert float((xr.grad - dx_want).abs().max()) < 1e-10 and float((wr.grad - dw_want).abs().max()) < 1e-9


def test_w64_fixture_is_the_float64_reference_and_the_oracle_reproduces_it():
    """tests/golden/rn_w64_train_b32.npz (the bf16 GPU test's reference: the REFERENCE module evaluated in float64, width 64, 32 images):
    the float64 oracle reproduces features, every gradient (norm + 64 samples) and every moved statistic to float64 rounding."""
    import zlib
    z = np.load(os.path.join(HERE, "golden", "rn_w64_train_b32.npz"))
    c = json.loads(bytes(z["meta"]).decode())
    layers, width, e, res, B = tuple(c["layers"]), c["width"], c["output_dim"], c["resolution"], c["batch"]
    sd = {k: (v * c["bn3_gain"] if (k.endswith("bn3.weight") and ".layer" in k) else v).double()
          for k, v in RO.make_state_dict(layers, width, e, res, c["wseed"]).items()}
    rs = np.random.RandomState(c["iseed"])
    px = torch.from_numpy(rs.standard_normal((B, 3, res, res)).astype(np.float32)).double()
    probe = torch.from_numpy(rs.standard_normal((B, e)).astype(np.float32)).double()
    leaves = {k: (v.clone().requires_grad_(True) if not k.endswith(("running_mean", "running_var")) else v) for k, v in sd.items()}
    stats = {}
    raw = RO.modified_resnet_forward(leaves, layers, width, px, train=True, new_stats=stats)
    assert float((raw.detach() - torch.from_numpy(z["image_features"])).abs().max()) < 1e-9
    out = raw / raw.norm(dim=-1, keepdim=True)
    (out * probe).sum().backward()
    n = 0
    for k, v in leaves.items():
        if not v.requires_grad:
            ref = torch.from_numpy(z["stat:" + k])
            assert float((stats[k] - ref).abs().max()) <= 1e-10 * max(1.0, float(ref.abs().max())), k
            continue
        g = v.grad
        want = float(z["gnorm:" + k])
        assert abs(float(g.norm()) - want) <= 1e-7 * want + 1e-12, k      # (+ 1e-12: attnpool.k_proj.bias' exact gradient is 0, both sides hold noise)
        idx = np.random.RandomState(zlib.crc32(k.encode()) & 0x7fffffff).choice(g.numel(), size=min(64, g.numel()), replace=False).astype(np.int64)
        samp = torch.from_numpy(z["gsamp:" + k])
        assert float((g.reshape(-1)[torch.from_numpy(idx)] - samp).abs().max()) <= 1e-7 * float(samp.abs().max()) + 1e-9 * want + 1e-12, k
        n += 1
    assert n == len([k for k in z.files if k.startswith("gnorm:")])


def test_whole_model_train_fixture_matches_the_oracles():
    """tests/golden/clip_rn_tiny_train_b6_l24.npz -- the REFERENCE CHINESE_CLIP with the ModifiedResNet tower in train() mode: loss, embeddings,
    every parameter gradient, every moved statistic -- against resnet_oracle (image tower, training mode) + clip_oracle (text tower, loss)."""
    import zlib
    from oracle import clip_oracle as O
    z = np.load(os.path.join(HERE, "golden", "clip_rn_tiny_train_b6_l24.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    cfg, case = meta["cfg"], meta["case"]
    layers, width = tuple(cfg["vision_layers"]), cfg["vision_width"]
    vit_like = dict(cfg, vision_layers=1, vision_width=64)
    sd = {k: v for k, v in O.make_state_dict(vit_like, case["wseed"]).items() if not k.startswith("visual.")}
    sd.update(RO.make_state_dict(layers, width, cfg["embed_dim"], cfg["image_resolution"], case["rn_wseed"]))
    px, ids = O.make_inputs(cfg, case["batch"], case["seq_len"], case["iseed"])
    leaves = {k: (v.clone().requires_grad_(True) if not k.endswith(("running_mean", "running_var")) else v) for k, v in sd.items()}
    stats = {}
    img = O.l2_normalize(RO.modified_resnet_forward(leaves, layers, width, px, train=True, new_stats=stats))
    txt = O.encode_text(leaves, cfg, ids)
    lpt = (txt @ img.t()) * leaves["logit_scale"].exp()
    loss = O.clip_loss(lpt)
    loss.backward()
    assert float((img.detach() - torch.from_numpy(z["image_embeds"])).abs().max()) < 1e-5
    assert float((txt.detach() - torch.from_numpy(z["text_embeds"])).abs().max()) < 1e-5
    assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-5 * float(z["loss"])
    scale = max([float(np.linalg.norm(z[k].astype(np.float64))) for k in z.files if k.startswith("grad/")])
    seen = 0
    for k in z.files:
        if k.startswith("grad/"):
            n = k[len("grad/"):]
            ref = torch.from_numpy(z[k]).double()
            got = leaves[n].grad.double().reshape(ref.shape)
            assert float((got - ref).norm()) <= 1e-4 * float(ref.norm()) + 2e-6 * scale, (n, float((got - ref).norm()) / (float(ref.norm()) + 1e-30))
            seen += 1
        elif k.startswith("gnorm/"):
            n = k[len("gnorm/"):]
            got = leaves[n].grad.double().reshape(-1)
            want = float(z[k])
            assert abs(float(got.norm()) - want) <= 1e-4 * want + 2e-6 * scale, n
            idx = np.random.RandomState(zlib.crc32(n.encode()) & 0x7fffffff).choice(got.numel(), size=64, replace=False).astype(np.int64)
            samp = torch.from_numpy(z["gsamp/" + n]).double()
            assert float((got[torch.from_numpy(idx)] - samp).abs().max()) <= 1e-4 * float(samp.abs().max()) + 1e-4 * want / got.numel() ** 0.5, n
            seen += 1
        elif k.startswith("nograd/"):
            assert leaves[k[len("nograd/"):]].grad is None
        elif k.startswith("stat/") and not k.endswith("num_batches_tracked"):
            ref = torch.from_numpy(z[k])
            assert float((stats[k[len("stat/"):]] - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), k
        elif k.endswith("num_batches_tracked"):
            assert int(z[k]) == 1
    assert seen >= 110
