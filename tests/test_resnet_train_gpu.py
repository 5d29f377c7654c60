"""ModifiedResNet tower, TRAINING path (BatchNorm batch statistics + backward pass) at the engine level: ezclip_rn_encode_image_train /
ezclip_rn_backward against oracle/resnet_oracle.py in training mode -- which tests/test_resnet_oracle.py pins to the reference module
(features, 75 parameter gradients, 44 updated running statistics: tests/golden/rn_tiny_train_b4.npz).  The tower returns L2-normalised
features (CHINESE_CLIP.forward normalises, modeling_chineseclip.py:360), so the loss here is sum(normalise(features) * probe)."""
import json
import os

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from oracle import resnet_oracle as RO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None
HERE = os.path.dirname(os.path.abspath(__file__))


def _oracle(sd, layers, width, px, probe):
    leaves = {k: (v.detach().double().clone().requires_grad_(True) if not k.endswith(("running_mean", "running_var")) else v.detach().double().clone())
              for k, v in sd.items()}
    stats = {}
    raw = RO.modified_resnet_forward(leaves, layers, width, px.double(), train=True, new_stats=stats)
    out = raw / raw.norm(dim=-1, keepdim=True)
    (out * probe.double()).sum().backward()
    return out.detach(), {k: v.grad.detach() for k, v in leaves.items() if v.requires_grad}, stats


def _oracle_bf16_deviation(sd, layers, width, px, probe, want_g):
    """per-parameter rel-L2 deviation of a plain torch-CPU bfloat16 evaluation of the same algorithm from the float64 one: what bf16
    costs ANY implementation of this tower.  It is large: BatchNorm over a four-image batch after every convolution amplifies the rounding
    of the layer before it and the ReLUs behind turn it into different branch decisions (median ~0.5 on the tiny fixture)."""
    leaves = {k: (v.detach().bfloat16().clone().requires_grad_(True) if not k.endswith(("running_mean", "running_var")) else v.detach().bfloat16().clone())
              for k, v in sd.items()}
    raw = RO.modified_resnet_forward(leaves, layers, width, px.bfloat16(), train=True, new_stats={})
    out = raw / raw.norm(dim=-1, keepdim=True)
    (out * probe.bfloat16()).sum().backward()
    return {k: float((leaves[k].grad.double() - ref).norm()) / (float(ref.norm()) + 1e-30) for k, ref in want_g.items()}


def _mismatches(grads, want_g, rel, floor):
    bad = []
    for k, ref in want_g.items():
        err = float((grads[k].detach().cpu().double() - ref).norm())
        if err > rel * float(ref.norm()) + floor:
            bad.append((k, err, float(ref.norm())))
    return bad


def _explained_by_relu_decisions(sd, layers, width, px, probe, grads, rel, floor, delta=2e-5, most=16, max_flips=4):
    """A pre-activation within float32 rounding of zero is a coin toss for a float32 implementation, and ONE ReLU decision taken the
    other way moves every gradient upstream by 0.3-1 % (round 5: the first GPU runs of this path "failed" on exactly that -- layer4's
    conv2, channel 50, one element at +7e-7; tools/rn_train_where.py).  The float64 oracle lists the pre-activations below `delta`
    (oracle/resnet_oracle.py: train_step_grads_by_steps near_zero / flips); the device gradient must equal, at the SAME tolerance, the exact
    gradient of a decision pattern that differs from the oracle's in at most `max_flips` of the `most` smallest of them.  The search is
    greedy: flip whichever single candidate lowers the total error most, repeat.  Returns (matched, info)."""
    sd64 = {k: v.double() for k, v in sd.items()}
    px64, pr64 = px.double(), probe.double()
    with torch.no_grad():
        raw = RO.modified_resnet_forward(sd64, layers, width, px64, train=True, new_stats={})
    nrm = raw.norm(dim=-1, keepdim=True)
    out = raw / nrm
    d_raw = (pr64 - out * (out * pr64).sum(dim=-1, keepdim=True)) / nrm          # loss = sum(normalise(raw) * probe)
    near = []
    RO.train_step_grads_by_steps(sd64, layers, width, px64, d_raw, near_zero=near, delta=delta)
    near.sort(key=lambda t: abs(t[2]))
    near = near[:most]
    got = {k: v.detach().cpu().double() for k, v in grads.items()}

    def total_error(g):
        return sum(float((got[k] - ref).norm()) / (float(ref.norm()) + floor) for k, ref in g.items())

    flips, tried = set(), 0
    _, g = RO.train_step_grads_by_steps(sd64, layers, width, px64, d_raw)
    best = total_error(g)
    for _ in range(max_flips):
        pick = None
        for s_, i, v in near:
            if (s_, i) in flips:
                continue
            _, g = RO.train_step_grads_by_steps(sd64, layers, width, px64, d_raw, flips=flips | {(s_, i)})
            tried += 1
            err = total_error(g)
            if err < 0.9 * best and (pick is None or err < pick[0]):
                pick = (err, (s_, i), g)
        if pick is None:
            break
        best, g_best = pick[0], pick[2]
        flips.add(pick[1])
        if not _mismatches(grads, g_best, rel, floor):
            return True, {"flipped": sorted(flips), "near_zero": len(near), "patterns_tried": tried}
    return False, {"flipped": sorted(flips), "near_zero": [(s_, i, v) for s_, i, v in near], "patterns_tried": tried, "total_error": best}


def _run(layers, width, e, res, sd, px, probe, dtype):
    from easynlp_amd.appzoo.clip.rn_tower import RnEngine
    eng = RnEngine(layers, width, e, res, L.DTYPE_F32 if dtype == "fp32" else L.DTYPE_BF16)
    tensors = {n: sd[n].to(DEV).contiguous() for n in eng.names}
    eng.sync_train(tensors)
    out = eng.encode_image_train(px.to(DEV))
    grads = {n: torch.full(eng.shapes[n], 9.0, dtype=torch.float32, device=DEV) for n in eng.names if not eng.is_statistic(n)}
    eng.backward(out, probe.to(DEV), grads)
    torch.cuda.synchronize()
    return eng, tensors, out, grads


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_training_tower_against_the_reference_fixture(dtype):
    z = np.load(os.path.join(HERE, "golden", "rn_tiny_train_b4.npz"))
    c = json.loads(bytes(z["meta"]).decode())
    layers, width, e, res = tuple(c["layers"]), c["width"], c["output_dim"], c["resolution"]
    sd = RO.make_state_dict(layers, width, e, res, c["wseed"])
    px, probe = torch.from_numpy(z["pixels"]), torch.from_numpy(z["probe"])
    want_out, want_g, want_s = _oracle(sd, layers, width, px, probe)
    raw_ref = torch.from_numpy(z["image_features"]).double()
    assert float((want_out - raw_ref / raw_ref.norm(dim=-1, keepdim=True)).abs().max()) < 1e-5      # the oracle IS the fixture's function
    eng, tensors, out, grads = _run(layers, width, e, res, sd, px, probe, dtype)
    # fp32 is the parity gate (2e-5).  bf16: z and y of 14 convolutions are stored in bf16 and every BatchNorm divides by the standard
    # deviation of a FOUR-image batch, which amplifies the rounding of the layer before it: 3.2e-2 max-abs on these 24 unit-norm features
    # (measured, round 5) against 2e-2 for the eval-mode tower on running statistics; direction still within 0.995
    tol = 2e-5 if dtype == "fp32" else 5e-2
    assert float((out.cpu().double() - want_out).abs().max()) < tol
    assert float(torch.nn.functional.cosine_similarity(out.cpu().double(), want_out).min()) > (0.999999 if dtype == "fp32" else 0.995)
    # the running statistics the forward moved: the reference module's own (fixture), through the bound buffers
    for k in want_s:
        ref = torch.from_numpy(z["stat:" + k]).double()
        got = tensors[k].cpu().double()
        assert float((got - ref).abs().max()) <= (2e-5 if dtype == "fp32" else 2e-2) * max(1.0, float(ref.abs().max())), k
    scale = max(float(v.norm()) for v in want_g.values())
    if dtype == "fp32":
        bad = _mismatches(grads, want_g, 1e-3, 1e-6 * scale)
        if bad:
            ok, info = _explained_by_relu_decisions(sd, layers, width, px, probe, grads, 1e-3, 1e-6 * scale)
            assert ok, (bad[:5], info)
    else:
        # bf16: gradients of this tower at batch 4 are dominated by rounding in ANY bf16 implementation (torch's own CPU bfloat16
        # evaluation deviates by ~0.5 in the median, _oracle_bf16_deviation): the device must not be worse than that -- every parameter
        # within max(8e-2, 2.5 x torch-bf16's deviation for it), the median over the parameters within 1.5 x torch-bf16's median.
        # fp32 above is the parity gate of this path.  (Measured, round 5: bn3.bias 0.58 against torch-bf16's 0.36, conv1.weight 0.52 / 0.58.)
        dev16 = _oracle_bf16_deviation(sd, layers, width, px, probe, want_g)
        errs = {}
        for k, ref in want_g.items():
            err = float((grads[k].cpu().double() - ref).norm())
            errs[k] = err / (float(ref.norm()) + 1e-30)
            assert err <= max(8e-2, 2.5 * dev16[k]) * float(ref.norm()) + 8e-5 * scale, (k, errs[k], dev16[k])
        live = [k for k in want_g if float(want_g[k].norm()) > 1e-6 * scale]
        assert float(np.median([errs[k] for k in live])) <= 1.5 * float(np.median([dev16[k] for k in live])) + 8e-2


@pytest.mark.parametrize("layers,width,e,res,B", [((1, 1, 1, 1), 64, 128, 64, 4), ((2, 1, 2, 1), 32, 64, 96, 3), ((1, 1, 1, 1), 48, 128, 64, 4),
                                                  ((1, 1, 1, 1), 64, 128, 64, 3)])
def test_training_tower_wider_shapes_against_the_oracle(layers, width, e, res, B):
    """channel counts that need no padding (width 64: the RN50 family's) and a second depth / resolution; fp32"""
    sd = RO.make_state_dict(layers, width, e, res, 17)
    g = torch.Generator().manual_seed(6)
    px, probe = torch.randn(B, 3, res, res, generator=g), torch.randn(B, e, generator=g)
    want_out, want_g, want_s = _oracle(sd, layers, width, px, probe)
    eng, tensors, out, grads = _run(layers, width, e, res, sd, px, probe, "fp32")
    assert float((out.cpu().double() - want_out).abs().max()) < 3e-5
    for k, ref in want_s.items():
        assert float((tensors[k].cpu().double() - ref).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max())), k
    scale = max(float(v.norm()) for v in want_g.values())
    bad = _mismatches(grads, want_g, 1e-3, 2e-6 * scale)
    if bad:       # (4 x 10^5 pre-activations per pass: one within 1e-6 of zero is the rule, not the exception)
        ok, info = _explained_by_relu_decisions(sd, layers, width, px, probe, grads, 1e-3, 2e-6 * scale)
        assert ok, (bad[:5], info)
    # a second forward + backward on the same engine reproduces the gradients bit for bit given the same statistics
    tensors2 = {n: sd[n].to(DEV).contiguous() for n in eng.names}
    eng.sync_train(tensors2)
    out2 = eng.encode_image_train(px.to(DEV))
    grads2 = {n: torch.zeros(eng.shapes[n], dtype=torch.float32, device=DEV) for n in grads}
    eng.backward(out2, probe.to(DEV), grads2)
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    # eval-mode inference after training steps uses the MOVED statistics (the inference copies were marked dirty)
    eng.sync(tensors2)
    ev = eng.encode_image(px.to(DEV))
    with torch.no_grad():
        sd_moved = {k: (tensors2[k].cpu() if k in tensors2 else v) for k, v in sd.items()}
        want_ev = RO.modified_resnet_forward(sd_moved, layers, width, px)
        want_ev = want_ev / want_ev.norm(dim=-1, keepdim=True)
    assert float((ev.cpu() - want_ev).abs().max()) < 5e-5


def test_clipapp_trains_the_resnet_tower_when_asked():
    """in train() mode the image tower runs on batch statistics, moves its running statistics and hands autograd a gradient for every
    visual.* parameter (the default since round 5); eval() mode uses the moved statistics; ``rn_train=False`` is the frozen tower."""
    from easynlp_amd.appzoo.clip.model import CLIPApp
    from oracle import clip_oracle as O
    cfg = dict(O.CONFIGS["tiny"], vision_layers=[1, 2, 1, 1], vision_width=16, image_resolution=64)
    g = torch.Generator().manual_seed(3)
    px = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    _, ids = O.make_inputs(O.CONFIGS["tiny"], 4, 24, 2)
    ids = ids.to(DEV)
    app = CLIPApp.from_config(cfg, seed=5, device=DEV, compute_dtype="fp32")
    vis = {n: p for n, p in app.named_parameters() if ".visual." in n or n.startswith("chinese_clip.visual.")}
    assert vis and all(p.requires_grad for p in vis.values())
    stats0 = {n: b.clone() for n, b in app.named_buffers() if n.endswith("running_mean")}
    app.train()
    out = app({"pixel_values": px, "input_ids": ids})
    loss = app.compute_loss(out, [])["loss"]
    loss.backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in vis.values())
    assert any(float(p.grad.abs().max()) > 0 for p in vis.values())
    moved = {n: b for n, b in app.named_buffers() if n.endswith("running_mean")}
    assert any(not torch.equal(moved[n], stats0[n]) for n in stats0)
    app.eval()
    with torch.no_grad():
        e1 = app({"pixel_values": px, "input_ids": ids}, feat=True)["image_embeds"]
        e2 = app({"pixel_values": px, "input_ids": ids}, feat=True)["image_embeds"]
    assert torch.equal(e1, e2) and bool(torch.isfinite(e1).all())
    frozen = CLIPApp.from_config(cfg, seed=5, device=DEV, compute_dtype="fp32", rn_train=False)
    assert not any(p.requires_grad for n, p in frozen.named_parameters() if "visual." in n)
