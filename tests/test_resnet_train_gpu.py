"""ModifiedResNet tower, TRAINING path (BatchNorm batch statistics + backward pass): ezclip_rn_encode_image_train / ezclip_rn_backward and
CLIPApp with this tower against the reference's own numbers (tests/golden/rn_tiny_train_b4.npz, rn_w64_train_b32.npz,
clip_rn_tiny_train_b6_l24.npz: tools/make_golden_resnet.py) and against oracle/resnet_oracle.py in training mode, which
tests/test_resnet_oracle.py pins to the reference module.  The tower returns L2-normalised features (CHINESE_CLIP.forward normalises,
modeling_chineseclip.py:360), so the tower-level loss is sum(normalise(features) * probe).

Bars (round 6).  fp32: features and moved statistics 2e-5 / 3e-5, EVERY gradient 1e-4 rel-L2, flat.  ReLU decisions: a pre-activation within
float32 rounding of zero is a coin toss for any float32 implementation, and one decision moves every gradient upstream by 0.3-1 %
(profiles/r5_rn_train_where.log) -- so the INPUT SEED of every fp32 case is chosen such that the float64 oracle sees no pre-activation
within MARGIN = 1e-5 of zero (ten times the float32 noise; asserted here, searched by tools/make_golden_resnet.py --search; 1e-4 is out of
reach: ~0.8 N d of N pre-activations lie within d of zero, i.e. ~25 of these towers' 3 x 10^5 inside 1e-4 whatever the seed).  The round-5
search over decision patterns survives as a DIAGNOSTIC printed when a case fails; it cannot make one pass.  bf16: a fixture at which bf16
means something (width 64, 32 images: rn_w64_train_b32, the reference evaluated in float64) against the activation-rounding floor."""
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
ROOT = os.path.dirname(HERE)
MARGIN = 1e-5


def _oracle(sd, layers, width, px, probe, store=None):
    leaves = {k: (v.detach().double().clone().requires_grad_(True) if not k.endswith(("running_mean", "running_var")) else v.detach().double().clone())
              for k, v in sd.items()}
    stats = {}
    kw = {} if store is None else {"store": store}
    raw = RO.modified_resnet_forward(leaves, layers, width, px.double(), train=True, new_stats=stats, **kw)
    out = raw / raw.norm(dim=-1, keepdim=True)
    (out * probe.double()).sum().backward()
    return out.detach(), {k: v.grad.detach() for k, v in leaves.items() if v.requires_grad}, stats, raw.detach()


def _d_raw(sd64, layers, width, px64, pr64):
    with torch.no_grad():
        raw = RO.modified_resnet_forward(sd64, layers, width, px64, train=True, new_stats={})
    nrm = raw.norm(dim=-1, keepdim=True)
    out = raw / nrm
    return (pr64 - out * (out * pr64).sum(dim=-1, keepdim=True)) / nrm          # loss = sum(normalise(raw) * probe)


def _assert_relu_margin(sd, layers, width, px):
    """the case's seed keeps every ReLU pre-activation MARGIN away from zero (float64): float32 cannot take another branch"""
    sd64 = {k: v.double() for k, v in sd.items()}
    near = []
    RO.train_step_grads_by_steps(sd64, layers, width, px.double(), torch.zeros(px.shape[0], sd["visual.attnpool.c_proj.weight"].shape[0], dtype=torch.float64),
                                 near_zero=near, delta=MARGIN)
    assert not near, "this case's input seed leaves pre-activations within %g of zero: %r (tools/make_golden_resnet.py --search)" % (MARGIN, near[:4])


def _mismatches(grads, want_g, rel, floor):
    bad = []
    for k, ref in want_g.items():
        err = float((grads[k].detach().cpu().double() - ref).norm())
        if err > rel * float(ref.norm()) + floor:
            bad.append((k, err / (float(ref.norm()) + 1e-30)))
    return bad


def _relu_decision_diagnostic(sd, layers, width, px, probe, grads, floor, delta=1e-4, most=16, max_flips=4):
    """DIAGNOSTIC ONLY (printed with a failure, never a reason to pass): which ReLU decisions, inverted, would bring the float64 oracle's
    gradient closest to the device's -- greedy over the `most` smallest pre-activations below `delta`."""
    sd64 = {k: v.double() for k, v in sd.items()}
    px64 = px.double()
    d_raw = _d_raw(sd64, layers, width, px64, probe.double())
    near = []
    RO.train_step_grads_by_steps(sd64, layers, width, px64, d_raw, near_zero=near, delta=delta)
    near = sorted(near, key=lambda t: abs(t[2]))[:most]
    got = {k: v.detach().cpu().double() for k, v in grads.items()}

    def total_error(g):
        return sum(float((got[k] - ref).norm()) / (float(ref.norm()) + floor) for k, ref in g.items())

    flips = set()
    best = total_error(RO.train_step_grads_by_steps(sd64, layers, width, px64, d_raw)[1])
    start = best
    for _ in range(max_flips):
        pick = None
        for s_, i, v in near:
            if (s_, i) in flips:
                continue
            err = total_error(RO.train_step_grads_by_steps(sd64, layers, width, px64, d_raw, flips=flips | {(s_, i)})[1])
            if err < 0.9 * best and (pick is None or err < pick[0]):
                pick = (err, (s_, i))
        if pick is None:
            break
        best = pick[0]
        flips.add(pick[1])
    return {"smallest_pre_activations": near[:6], "flips_that_lower_the_error": sorted(flips), "summed_rel_error": (start, best)}


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


def _assert_fp32_gradients(sd, layers, width, px, probe, grads, want_g):
    scale = max(float(v.norm()) for v in want_g.values())
    bad = _mismatches(grads, want_g, 1e-4, 1e-6 * scale)
    if bad:
        info = _relu_decision_diagnostic(sd, layers, width, px, probe, grads, 1e-6 * scale)
        raise AssertionError("fp32 gradients off by more than 1e-4 rel-L2: %r\n(diagnostic) %r" % (sorted(bad, key=lambda t: -t[1])[:6], info))


def test_training_tower_against_the_reference_fixture():
    """fp32 against the REFERENCE module's train-mode pass (features, 75 parameter gradients, 44 moved running statistics)"""
    z = np.load(os.path.join(HERE, "golden", "rn_tiny_train_b4.npz"))
    c = json.loads(bytes(z["meta"]).decode())
    layers, width, e, res = tuple(c["layers"]), c["width"], c["output_dim"], c["resolution"]
    sd = RO.make_state_dict(layers, width, e, res, c["wseed"])
    px, probe = torch.from_numpy(z["pixels"]), torch.from_numpy(z["probe"])
    _assert_relu_margin(sd, layers, width, px)
    # the fixture's loss is sum(features * probe) on the RAW features; the tower's entry points return normalised features: the float64
    # oracle (pinned to this fixture's gradients by tests/test_resnet_oracle.py) provides the gradients of the normalised loss
    want_out, want_g, want_s, raw = _oracle(sd, layers, width, px, probe)
    raw_ref = torch.from_numpy(z["image_features"]).double()
    assert float((raw - raw_ref).abs().max()) < 1e-5 * max(1.0, float(raw_ref.abs().max()))      # the oracle IS the fixture's function
    eng, tensors, out, grads = _run(layers, width, e, res, sd, px, probe, "fp32")
    assert float((out.cpu().double() - want_out).abs().max()) < 2e-5
    assert float(torch.nn.functional.cosine_similarity(out.cpu().double(), want_out).min()) > 0.999999
    for k in want_s:               # the running statistics the forward moved: the reference module's own (fixture), through the bound buffers
        ref = torch.from_numpy(z["stat:" + k]).double()
        assert float((tensors[k].cpu().double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), k
    _assert_fp32_gradients(sd, layers, width, px, probe, grads, want_g)


# (layers, width, output_dim, resolution, batch, input seed): seeds from the margin search (weights: seed 17)
WIDE = [((1, 1, 1, 1), 64, 128, 32, 4, 16),      # the RN50 family's channel counts: no padding anywhere
        ((2, 1, 2, 1), 32, 64, 32, 4, 2),        # a second depth; 96-channel blocks padded to 128
        ((1, 1, 1, 1), 48, 128, 32, 4, 9),       # padded channels in every layer
        ((1, 1, 1, 1), 64, 128, 32, 6, 10)]      # a batch that is not a power of two


@pytest.mark.parametrize("layers,width,e,res,B,iseed", WIDE)
def test_training_tower_wider_shapes_against_the_oracle(layers, width, e, res, B, iseed):
    sd = RO.make_state_dict(layers, width, e, res, 17)
    g = torch.Generator().manual_seed(iseed)
    px, probe = torch.randn(B, 3, res, res, generator=g), torch.randn(B, e, generator=g)
    _assert_relu_margin(sd, layers, width, px)
    want_out, want_g, want_s, _ = _oracle(sd, layers, width, px, probe)
    eng, tensors, out, grads = _run(layers, width, e, res, sd, px, probe, "fp32")
    assert float((out.cpu().double() - want_out).abs().max()) < 3e-5
    for k, ref in want_s.items():
        assert float((tensors[k].cpu().double() - ref).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max())), k
    _assert_fp32_gradients(sd, layers, width, px, probe, grads, want_g)
    # a second forward + backward on the same engine reproduces the features bit for bit given the same statistics
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


class _RoundBf16(torch.autograd.Function):
    """x -> bf16(x) with a straight-through gradient: what STORING an activation in bf16 does to the forward pass, and nothing else"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def test_bf16_training_tower_against_the_float64_reference_fixture():
    """bf16 where bf16 means something: width 64 (no channel padding), 32 images of 64 x 64, (2, 2, 2, 2) blocks -- every BatchNorm
    averages over 128 ... 32768 values.  Reference: the REFERENCE module evaluated in float64 (rn_w64_train_b32.npz: gradient norms + 64
    samples per parameter, moved statistics); the float64 oracle is pinned to it at 1e-7 first, then provides the full gradients.
    Bound per parameter: max(2e-2, 2 x floor), floor = the same error for the float64 oracle when every tensor an implementation
    keeps between two kernels is rounded to bf16 (straight-through; oracle `store` hook) -- what storing activations in bf16 costs ANY
    implementation.  The table goes to gpurun_out/r6_bf16_grad_error_rn.md (committed as profiles/r6_bf16_grad_error_rn.md).
    What the floor turned out to be (CPU, before the first GPU run): 0.62 in the median with unit gains on the blocks' last BatchNorm, 0.24
    with those gains at 0.2 (this fixture), 0.18 for (1, 1, 1, 1) blocks -- a random-init BatchNorm-ReLU tower amplifies a perturbation from
    layer to layer (features: 0.1 % rounding per stored tensor -> 5-10 % at the output), and 0.3 % of the ReLU decisions of every layer
    differ between the rounded and the exact pass.  2 x floor is therefore a LOOSE bound on this tower whatever the fixture: a gradient that
    is wrong by a sign (error 2.0) or missing a term fails it, a 10 % defect does not -- those are caught by the fp32 tower tests above (the
    same orchestration code, templated on the type), by the operator tests of every bf16 kernel against float64 on identical inputs
    (tests/test_resnet_train_ops_gpu.py) and by the tower-level A/B of the weight-gradient routes below."""
    z = np.load(os.path.join(HERE, "golden", "rn_w64_train_b32.npz"))
    c = json.loads(bytes(z["meta"]).decode())
    layers, width, e, res, B = tuple(c["layers"]), c["width"], c["output_dim"], c["resolution"], c["batch"]
    sd = RO.make_state_dict(layers, width, e, res, c["wseed"])
    for k in sd:                      # (tools/make_golden_resnet.py w64_state_dict: small gains on every block's last BatchNorm)
        if k.endswith("bn3.weight") and ".layer" in k:
            sd[k] = sd[k] * c["bn3_gain"]
    rs = np.random.RandomState(c["iseed"])
    px = torch.from_numpy(rs.standard_normal((B, 3, res, res)).astype(np.float32))
    probe = torch.from_numpy(rs.standard_normal((B, e)).astype(np.float32))
    want_out, want_g, want_s, raw = _oracle(sd, layers, width, px, probe)
    assert float((raw - torch.from_numpy(z["image_features"])).abs().max()) < 1e-9
    import zlib
    for k, g in want_g.items():       # the oracle == the float64 reference, parameter by parameter
        want = float(z["gnorm:" + k])
        assert abs(float(g.norm()) - want) <= 1e-7 * want + 1e-12, k      # (+ 1e-12: attnpool.k_proj.bias' exact gradient is 0, both sides hold noise)
        idx = np.random.RandomState(zlib.crc32(k.encode()) & 0x7fffffff).choice(g.numel(), size=min(64, g.numel()), replace=False).astype(np.int64)
        samp = torch.from_numpy(z["gsamp:" + k])
        assert float((g.reshape(-1)[torch.from_numpy(idx)] - samp).abs().max()) <= 1e-7 * float(samp.abs().max()) + 1e-9 * want + 1e-12, k
    floor_out, floor_g, _, _ = _oracle(sd, layers, width, px, probe, store=_RoundBf16.apply)
    eng, tensors, out, grads = _run(layers, width, e, res, sd, px, probe, "bf16")
    feat_floor = float((floor_out - want_out).abs().max())
    feat_err = float((out.cpu().double() - want_out).abs().max())
    assert feat_err < max(1e-2, 2.0 * feat_floor), (feat_err, feat_floor)
    assert float(torch.nn.functional.cosine_similarity(out.cpu().double(), want_out).min()) > 0.998
    for k in want_s:
        ref = torch.from_numpy(z["stat:" + k]).double()
        assert float((tensors[k].cpu().double() - ref).abs().max()) <= 1e-2 * max(1.0, float(ref.abs().max())), k
    # attnpool.k_proj.bias: a constant added to every key of the one-query attention -- softmax is shift-invariant, its exact gradient is 0
    # (4e-16 in the float64 reference): not a gradient to take a relative error of.  Checked for smallness against v_proj.bias instead.
    zero_by_math = "visual.attnpool.k_proj.bias"
    assert float(grads[zero_by_math].double().norm()) <= 2e-2 * float(want_g["visual.attnpool.v_proj.bias"].norm())
    names = [k for k in want_g if k != zero_by_math]
    err = np.array([float((grads[k].cpu().double() - want_g[k]).norm()) / (float(want_g[k].norm()) + 1e-30) for k in names])
    flo = np.array([float((floor_g[k] - want_g[k]).norm()) / (float(want_g[k].norm()) + 1e-30) for k in names])
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    order = np.argsort(-err / np.maximum(flo, 1e-12))
    with open(os.path.join(out_dir, "r6_bf16_grad_error_rn.md"), "w") as f:
        f.write("# ModifiedResNet training tower, bf16 pipeline: per-parameter gradient error vs the float64 reference (`rn_w64_train_b32`)\n\n"
                "Written by tests/test_resnet_train_gpu.py::test_bf16_training_tower_against_the_float64_reference_fixture on the GPU box.  rel-L2 = "
                "|g_hip - g_ref| / |g_ref|; `floor` = the same for the float64 oracle with every stored activation rounded to bf16 (straight-through).  "
                "Bound: max(2e-2, 2 x floor) per parameter.\n\n")
        f.write("* features (unit norm, %d values of typical size %.3f): max-abs error %.4f, floor %.4f\n" % (want_out.numel(), float(want_out.abs().mean()), feat_err, feat_floor))
        f.write("* parameters: %d   error: median %.3e  90th percentile %.3e  max %.3e   floor: median %.3e  max %.3e   error / floor: median %.2f  max %.2f\n\n"
                % (len(names), float(np.median(err)), float(np.quantile(err, 0.9)), float(err.max()), float(np.median(flo)), float(flo.max()),
                   float(np.median(err / np.maximum(flo, 1e-12))), float((err / np.maximum(flo, 1e-12)).max())))
        f.write("| parameter | shape | |g_ref| | rel-L2 bf16 | floor | ratio |\n|---|---|---|---|---|---|\n")
        for i in order:
            k = names[i]
            f.write("| %s | %s | %.3e | %.3e | %.3e | %.2f |\n" % (k, "x".join(str(d) for d in want_g[k].shape), float(want_g[k].norm()), err[i], flo[i],
                                                                     err[i] / max(flo[i], 1e-12)))
    worst = [(names[i], float(err[i]), float(flo[i])) for i in range(len(names)) if err[i] > max(2e-2, 2.0 * flo[i])]
    assert not worst, worst[:8]
    assert float(np.median(err)) <= max(2e-2, 2.0 * float(np.median(flo))), (float(np.median(err)), float(np.median(flo)))


def test_weight_gradient_routes_agree_at_rn50_scale():
    """The operands-once weight-gradient kernels (rn_wgrad3x3_c64, rn_tn_skinny, the gathering TN kernel: resnet.hip rn_wgrad) against the
    explicit column matrix + generic / 8-phase TN products, at the TOWER level: RN50 (3, 4, 6, 3) width 64, 224 x 224, batch 8, bf16 --
    same operands, another summation order: every weight gradient agrees to 1e-3 rel-L2 and everything that is not a weight gradient bit
    for bit (EZCLIP_RN_EXPLICIT_IM2COL: 0 = the default routes, 1 = explicit im2col everywhere)."""
    layers, width, e, res, B = (3, 4, 6, 3), 64, 1024, 224, 8
    sd = RO.make_state_dict(layers, width, e, res, 5)
    g = torch.Generator().manual_seed(11)
    px, probe = torch.randn(B, 3, res, res, generator=g), torch.randn(B, e, generator=g)
    from easynlp_amd.appzoo.clip.rn_tower import RnEngine
    eng = RnEngine(layers, width, e, res, L.DTYPE_BF16)
    tensors = {n: sd[n].to(DEV).contiguous() for n in eng.names}
    eng.sync_train(tensors)
    out = eng.encode_image_train(px.to(DEV))
    results = {}
    old = os.environ.get("EZCLIP_RN_EXPLICIT_IM2COL")
    try:
        for mode in ("0", "1", "2", "3"):
            os.environ["EZCLIP_RN_EXPLICIT_IM2COL"] = mode
            grads = {n: torch.full(eng.shapes[n], 9.0, dtype=torch.float32, device=DEV) for n in eng.names if not eng.is_statistic(n)}
            eng.backward(out, probe.to(DEV), grads)
            torch.cuda.synchronize()
            results[mode] = grads
    finally:
        if old is None:
            os.environ.pop("EZCLIP_RN_EXPLICIT_IM2COL", None)
        else:
            os.environ["EZCLIP_RN_EXPLICIT_IM2COL"] = old
    conv_w = [n for n in results["0"] if n.endswith(".weight") and results["0"][n].dim() == 4]
    assert len(conv_w) == 3 + 16 * 3 + 4
    for mode in ("1", "2", "3"):
        worst = ("", 0.0)
        for n in results["0"]:
            a, b = results["0"][n].double(), results[mode][n].double()
            assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0, n
            rel = float((a - b).norm()) / (float(b.norm()) + 1e-30)
            if n in conv_w:
                worst = max(worst, (n, rel), key=lambda t: t[1])
                assert rel <= 1e-3, (mode, n, rel)
            else:     # BatchNorm, attention-pool gradients and the input-gradient chain do not depend on the weight-gradient route: equal up
                # to the order of the float atomics of the column sums behind the bias gradients (6e-7 measured on attnpool.k_proj.bias)
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, (mode, n, rel)
        print("EZCLIP_RN_EXPLICIT_IM2COL=0 vs %s: worst conv weight gradient rel-L2 %.2e (%s)" % (mode, worst[1], worst[0]))


def _clip_rn_fixture():
    z = np.load(os.path.join(HERE, "golden", "clip_rn_tiny_train_b6_l24.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta["cfg"], meta["case"]


def _clip_rn_app(tmp_path, cfg, case, dtype):
    from easynlp_amd.appzoo.clip.model import CLIPApp
    from oracle import clip_oracle as O
    from oracle import ref_harness as R
    vit_like = dict(cfg, vision_layers=1, vision_width=64)
    sd = {k: v for k, v in O.make_state_dict(vit_like, case["wseed"]).items() if not k.startswith("visual.")}
    sd.update(RO.make_state_dict(tuple(cfg["vision_layers"]), cfg["vision_width"], cfg["embed_dim"], cfg["image_resolution"], case["rn_wseed"]))
    R.write_checkpoint_dir(str(tmp_path), cfg, sd)
    app = CLIPApp(str(tmp_path), user_defined_parameters={"clip_compute_dtype": dtype}).cuda()
    px, ids = O.make_inputs(cfg, case["batch"], case["seq_len"], case["iseed"])
    return app, sd, px, ids


def test_clipapp_trains_the_resnet_tower_against_the_reference_fixture(tmp_path):
    """The WHOLE model with this tower in train() mode -- CHINESE_CLIP(vision_layers=(1, 2, 1, 1), ...) under core/trainer.py:658-661 --
    against the reference's own pass (tools/make_golden_resnet.py main_clip_train): embeddings, logits, loss, the gradient of EVERY
    parameter of both towers and of logit_scale at 1e-4, every moved running statistic, the num_batches_tracked counters; eval() then
    uses the moved statistics; ``clip_rn_train=0`` is the frozen tower."""
    import zlib
    z, cfg, case = _clip_rn_fixture()
    app, sd, px, ids = _clip_rn_app(tmp_path, cfg, case, "fp32")
    _assert_relu_margin({k: v for k, v in sd.items() if k.startswith("visual.")}, tuple(cfg["vision_layers"]), cfg["vision_width"], px)
    vis = {n: p for n, p in app.named_parameters() if ".visual." in n}
    assert vis and all(p.requires_grad for p in vis.values())
    app.train()
    out = app({"pixel_values": px.to(DEV), "input_ids": ids.to(DEV)})
    loss = app.compute_loss(out, [])["loss"]
    loss.backward()
    torch.cuda.synchronize()
    assert float((out["image_embeds"].detach().cpu() - torch.from_numpy(z["image_embeds"])).abs().max()) < 2e-5
    assert float((out["text_embeds"].detach().cpu() - torch.from_numpy(z["text_embeds"])).abs().max()) < 2e-5
    assert float((out["logits_per_text"].detach().cpu() - torch.from_numpy(z["logits_per_text"])).abs().max()) < 3e-4
    assert abs(float(loss.item()) - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    named = {n[len("chinese_clip."):]: p for n, p in app.named_parameters()}
    checked = 0
    full = [k[len("grad/"):] for k in z.files if k.startswith("grad/")]
    digests = [k[len("gnorm/"):] for k in z.files if k.startswith("gnorm/")]
    scale = max([float(np.linalg.norm(z["grad/" + n].astype(np.float64))) for n in full] + [float(z["gnorm/" + n]) for n in digests])
    for n in full:
        ref = torch.from_numpy(z["grad/" + n]).double()
        got = named[n].grad.detach().cpu().double().reshape(ref.shape)
        assert float((got - ref).norm()) <= 1e-4 * float(ref.norm()) + 1e-6 * scale, (n, float((got - ref).norm()) / (float(ref.norm()) + 1e-30))
        checked += 1
    for n in digests:
        want = float(z["gnorm/" + n])
        got = named[n].grad.detach().cpu().double().reshape(-1)
        assert abs(float(got.norm()) - want) <= 1e-4 * want + 1e-6 * scale, (n, float(got.norm()), want)
        idx = np.random.RandomState(zlib.crc32(n.encode()) & 0x7fffffff).choice(got.numel(), size=64, replace=False).astype(np.int64)
        samp = torch.from_numpy(z["gsamp/" + n]).double()
        assert float((got[torch.from_numpy(idx)] - samp).abs().max()) <= 1e-4 * float(samp.abs().max()) + 1e-4 * want / got.numel() ** 0.5, n
        checked += 1
    assert checked == len(full) + len(digests) >= 110
    for n in [k[len("nograd/"):] for k in z.files if k.startswith("nograd/")]:       # (the unused BertPooler: None in the reference, None here)
        assert named[n].grad is None, n
    state = app.state_dict()
    n_stats = 0
    for k in z.files:
        if not k.startswith("stat/"):
            continue
        ref = torch.from_numpy(z[k])
        got = state["chinese_clip." + k[len("stat/"):]].cpu()
        if k.endswith("num_batches_tracked"):
            assert int(got) == int(ref) == 1, k
        else:
            assert float((got.double() - ref.double()).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), k
        n_stats += 1
    assert n_stats == 66
    app.eval()
    with torch.no_grad():
        e1 = app({"pixel_values": px.to(DEV), "input_ids": ids.to(DEV)}, feat=True)["image_embeds"]
        sd_moved = {k[len("chinese_clip."):]: v.cpu() for k, v in state.items() if k.startswith("chinese_clip.visual.") and "num_batches" not in k}
        want_ev = RO.modified_resnet_forward(sd_moved, tuple(cfg["vision_layers"]), cfg["vision_width"], px)
        want_ev = want_ev / want_ev.norm(dim=-1, keepdim=True)
    assert float((e1.cpu() - want_ev).abs().max()) < 5e-5
    from easynlp_amd.appzoo.clip.model import CLIPApp
    frozen = CLIPApp(str(tmp_path), user_defined_parameters={"clip_compute_dtype": "fp32", "clip_rn_train": "0"}).cuda()
    assert not any(p.requires_grad for n, p in frozen.named_parameters() if "visual." in n)


def test_two_training_forwards_before_one_backward_keep_their_own_activations(tmp_path):
    """Two micro-batches whose losses are summed, and a train-mode feature call between a forward and its backward: every forward owns
    the workspace its activations are saved in (round 5's advisor finding: one slot per engine silently produced the first graph's
    gradients from the second forward's activations).  The gradient of loss(a) + loss(b) must equal grad loss(a) + grad loss(b) computed
    one at a time; a workspace no forward filled is refused by the library."""
    z, cfg, case = _clip_rn_fixture()
    app, sd, px, ids = _clip_rn_app(tmp_path, cfg, case, "fp32")
    app.train()
    px, ids = px.to(DEV), ids.to(DEV)
    stats0 = {n: b.clone() for n, b in app.named_buffers() if "running_" in n}

    def reset_stats():
        with torch.no_grad():
            for n, b in app.named_buffers():
                if n in stats0:
                    b.copy_(stats0[n])

    def one(sl):
        app.zero_grad(set_to_none=True)
        reset_stats()
        app.compute_loss(app({"pixel_values": px[sl], "input_ids": ids[sl].clone()}), [])["loss"].backward()
        return {n: p.grad.detach().clone() for n, p in app.named_parameters() if p.grad is not None}

    ga, gb = one(slice(0, 3)), one(slice(3, 6))
    app.zero_grad(set_to_none=True)
    reset_stats()
    la = app.compute_loss(app({"pixel_values": px[0:3], "input_ids": ids[0:3].clone()}), [])["loss"]
    reset_stats()                      # (the second forward must see the statistics `one(slice(3, 6))` saw; they do not enter the gradients)
    lb = app.compute_loss(app({"pixel_values": px[3:6], "input_ids": ids[3:6].clone()}), [])["loss"]
    with torch.no_grad():              # a no-grad train-mode feature call in between must not disturb either graph
        app({"pixel_values": px[1:5], "input_ids": ids[1:5].clone()}, feat=True)
    (la + lb).backward()
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in app.named_parameters():
        if n not in ga:
            continue
        want = ga[n].double() + gb[n].double()
        rel = float((p.grad.double() - want).norm()) / (float(want.norm()) + 1e-30)
        worst = max(worst, rel)
        assert rel <= 2e-6, (n, rel)
    # the library refuses a workspace no forward of the handle filled, and a batch that is not the forward's
    rn = app._rn
    bogus = L.alloc_bytes(rn.lib.ezclip_rn_train_saved_bytes(rn.handle, 3), DEV)
    grads = {n: torch.empty(rn.shapes[n], dtype=torch.float32, device=DEV) for n in rn.names if not rn.is_statistic(n)}
    feats = torch.zeros(3, rn.output_dim, device=DEV)
    with pytest.raises(L.EzclipError, match="no training forward of this handle filled"):
        rn.backward(feats, feats, grads, saved=bogus)
    with pytest.raises(L.EzclipError, match="holds a forward over"):
        rn.backward(torch.zeros(5, rn.output_dim, device=DEV), torch.zeros(5, rn.output_dim, device=DEV), grads)


def test_training_batch_of_one_and_ragged_output_dim():
    """A DataLoader tail batch of ONE pair trains where torch's BatchNorm2d does (more than one value per channel: batch * (R / 32)^2 > 1)
    and is refused, with torch's own reason, where it does not; an output_dim the training products cannot take is refused BEFORE the
    forward moves any running statistic."""
    from easynlp_amd.appzoo.clip.rn_tower import RnEngine
    layers, width, e, res = (1, 1, 1, 1), 16, 24, 64
    sd = RO.make_state_dict(layers, width, e, res, 3)
    px = torch.randn(1, 3, res, res, generator=torch.Generator().manual_seed(0))
    probe = torch.randn(1, e, generator=torch.Generator().manual_seed(2))
    _assert_relu_margin(sd, layers, width, px)
    want_out, want_g, want_s, _ = _oracle(sd, layers, width, px, probe)
    eng, tensors, out, grads = _run(layers, width, e, res, sd, px, probe, "fp32")
    assert float((out.cpu().double() - want_out).abs().max()) < 3e-5
    for k, ref in want_s.items():
        assert float((tensors[k].cpu().double() - ref).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max())), k
    _assert_fp32_gradients(sd, layers, width, px, probe, grads, want_g)
    small = RnEngine(layers, width, e, 32, L.DTYPE_F32)                 # 1 x 1 final map: one value per channel at batch 1
    t32 = {n: RO.make_state_dict(layers, width, e, 32, 3)[n].to(DEV).contiguous() for n in small.names}
    small.sync_train(t32)
    with pytest.raises(L.EzclipError, match="more than 1 value per channel"):
        small.encode_image_train(torch.zeros(1, 3, 32, 32, device=DEV))
    ragged = RnEngine(layers, width, 20, res, L.DTYPE_BF16)             # 20 * 2 bytes: not a multiple of 16
    sd22 = RO.make_state_dict(layers, width, 20, res, 3)
    t22 = {n: sd22[n].to(DEV).contiguous() for n in ragged.names}
    ragged.sync_train(t22)
    before = t22["visual.bn1.running_mean"].clone()
    with pytest.raises(L.EzclipError, match="output_dim"):
        ragged.encode_image_train(torch.randn(4, 3, res, res, device=DEV))
    assert torch.equal(t22["visual.bn1.running_mean"], before)
    ragged.sync(t22)
    assert bool(torch.isfinite(ragged.encode_image(torch.randn(4, 3, res, res, device=DEV))).all())      # inference accepts it
