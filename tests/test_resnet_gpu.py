"""ModifiedResNet image tower on the GPU (csrc/resnet.hip: implicit 3x3 convolutions on the MFMA GEMM, BatchNorm folded into the
packed weights, AttentionPool2d on the one-query attention kernel) against the fixture the REAL reference produced
(tests/golden/rn_tiny_b3.npz) and the CPU oracle (oracle/resnet_oracle.py, itself pinned to the reference) -- fp32 and bf16,
the RN50 shape included -- and the drop-in CLIPApp built from a `vision_layers` tuple."""
import json
import os

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.clip import CLIPApp
from easynlp_amd.appzoo.clip.rn_tower import RnEngine
from oracle import clip_oracle as O
from oracle import ref_harness as R
from oracle import resnet_oracle as RO

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_tower(layers, width, e, res, sd, px, dtype):
    eng = RnEngine(layers, width, e, res, L.dtype_code(dtype))
    dev = {n: sd[n].cuda().contiguous() for n in eng.names}
    eng.sync(dev)
    out = eng.encode_image(px.cuda())
    torch.cuda.synchronize()
    return out.cpu(), eng, dev


def norm(x):
    return x / x.norm(dim=-1, keepdim=True)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_tower_matches_the_reference_fixture(dtype):
    z = np.load(os.path.join(HERE, "golden", "rn_tiny_b3.npz"))
    c = json.loads(bytes(z["meta"]).decode())
    sd = RO.make_state_dict(c["layers"], c["width"], c["output_dim"], c["resolution"], c["wseed"])
    got, _, _ = run_tower(c["layers"], c["width"], c["output_dim"], c["resolution"], sd, torch.from_numpy(z["pixels"]), dtype)
    want = norm(torch.from_numpy(z["image_features"]))
    err = float((got - want).abs().max())
    assert err < (2e-5 if dtype == "fp32" else 2e-2), err
    assert float(torch.nn.functional.cosine_similarity(got, want).min()) > (0.999999 if dtype == "fp32" else 0.999)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("layers,width,e,res,B", [((3, 4, 6, 3), 64, 1024, 224, 3), ((2, 1, 2, 1), 8, 32, 96, 5),
                                                  ((1, 1, 1, 1), 32, 512, 160, 2)])
def test_tower_matches_the_oracle(layers, width, e, res, B, dtype):
    """(3, 4, 6, 3) x 64 at 224 is RN50: 256-column tiles, the 8-phase kernel on the 1x1 convolutions, 50 attention-pool tokens"""
    sd = RO.make_state_dict(layers, width, e, res, 11)
    g = torch.Generator().manual_seed(4)
    px = torch.randn(B, 3, res, res, generator=g)
    with torch.no_grad():
        want = norm(RO.modified_resnet_forward(sd, layers, width, px))
    got, eng, dev = run_tower(layers, width, e, res, sd, px, dtype)
    err = float((got - want).abs().max())
    assert err < (5e-5 if dtype == "fp32" else 2.5e-2), err
    assert float(torch.nn.functional.cosine_similarity(got, want).min()) > (0.999999 if dtype == "fp32" else 0.998)
    # the same again: deterministic; one image alone: the same row (chunking / batch independence)
    again = eng.encode_image(px.cuda()).cpu()
    assert torch.equal(again, got)
    one = eng.encode_image(px[1:2].cuda()).cpu()
    assert float((one - got[1:2]).abs().max()) < (1e-6 if dtype == "fp32" else 1e-2)
    # changed statistics reach the packed copies (BatchNorm is folded at refresh time)
    name = "visual.layer1.0.bn2.running_var"
    dev[name].mul_(1.5)
    eng.sync(dev)
    moved = eng.encode_image(px.cuda()).cpu()
    sd2 = dict(sd)
    sd2[name] = sd[name] * 1.5
    with torch.no_grad():
        want2 = norm(RO.modified_resnet_forward(sd2, layers, width, px))
    assert float((moved - got).abs().max()) > 1e-4
    assert float((moved - want2).abs().max()) < (5e-5 if dtype == "fp32" else 2.5e-2)


def test_a_batch_walked_in_chunks_equals_one_pass():
    """The tower bounds its activation buffers and walks large batches in chunks (1024 RN50 images: seven of them); with the
    bound lowered (ezclip_debug_set(10, MiB)) seven images of a small tower go through in chunks of two, three, ... -- the same
    rows, bit for bit, as in one pass."""
    layers, width, e, res, B = (1, 2, 1, 1), 16, 64, 128, 7
    sd = RO.make_state_dict(layers, width, e, res, 3)
    px = torch.randn(B, 3, res, res, generator=torch.Generator().manual_seed(2))
    lib = L.load()
    whole, _, _ = run_tower(layers, width, e, res, sd, px, "bf16")
    try:
        L.check(lib.ezclip_debug_set(10, 1))           # 1 MiB per buffer, 512 KiB per image of this tower: chunks of 2, 2, 2, 1
        a, eng, dev = run_tower(layers, width, e, res, sd, px, "bf16")
        per_image = lib.ezclip_rn_workspace_bytes(eng.handle, 1)
        assert lib.ezclip_rn_workspace_bytes(eng.handle, B) <= 2 * per_image + 4096
        assert torch.equal(a, whole)
    finally:
        L.check(lib.ezclip_debug_set(10, 256))
    with torch.no_grad():
        want = norm(RO.modified_resnet_forward(sd, layers, width, px))
    assert float((whole - want).abs().max()) < 2.5e-2


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_dropin_clipapp_with_a_resnet_tower(tmp_path, dtype):
    """config.json with a `vision_layers` tuple: forward() / compute_loss() as the reference's CHINESE_CLIP builds it
    (modeling_chineseclip.py:279-287,352-365).  eval(): running statistics, equal to the eval-mode oracle.  train(): the tower trains
    (round 5: batch statistics, gradients for visual.*; the numbers are tests/test_resnet_train_gpu.py's business); with
    clip_rn_train=0 it is a frozen tower and only the text side gets gradients."""
    cfg = dict(O.CONFIGS["tiny"], vision_layers=[1, 2, 1, 1], vision_width=16, image_resolution=64)
    sd = {k: v for k, v in O.make_state_dict(O.CONFIGS["tiny"], 5).items() if not k.startswith("visual.")}
    sd.update(RO.make_state_dict(cfg["vision_layers"], 16, cfg["embed_dim"], 64, 5))
    R.write_checkpoint_dir(str(tmp_path), cfg, sd)
    B, Lq = 6, 24
    _, ids = O.make_inputs(O.CONFIGS["tiny"], B, Lq, 2)
    px = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        img = norm(RO.modified_resnet_forward(sd, cfg["vision_layers"], 16, px))
        txt = O.encode_text(sd, O.CONFIGS["tiny"], ids)
        logits = float(torch.exp(sd["logit_scale"])) * txt @ img.t()
        ref_loss = float(O.clip_loss(logits))
    f32 = dtype == "fp32"
    for frozen in (True, False):
        udp = {"clip_compute_dtype": dtype}
        if frozen:
            udp["clip_rn_train"] = "0"
        app = CLIPApp(str(tmp_path), user_defined_parameters=udp).cuda()
        named = dict(app.named_parameters())
        assert all(p.requires_grad != frozen for n, p in named.items() if ".visual." in n)
        if not frozen:
            app.eval()                       # (a frozen tower ignores the mode: eval-mode statistics either way)
        out = app({"pixel_values": px, "input_ids": ids})
        loss = app.compute_loss(out, [])["loss"]
        assert float((out["image_embeds"].detach().cpu() - img).abs().max()) < (2e-5 if f32 else 2e-2)
        assert float((out["text_embeds"].detach().cpu() - txt).abs().max()) < (2e-5 if f32 else 1e-2)
        assert float((out["logits_per_text"].detach().cpu() - logits).abs().max()) < (4e-4 if f32 else 0.3)
        assert abs(loss.item() - ref_loss) < (1e-4 if f32 else 3e-2)
        if frozen:
            loss.backward()
            assert all(p.grad is None for n, p in named.items() if ".visual." in n)
            assert named["chinese_clip.text_projection"].grad is not None and float(named["chinese_clip.text_projection"].grad.abs().sum()) > 0
        else:
            app.train()
            stat0 = {n: b.clone() for n, b in app.named_buffers() if n.endswith("running_var")}
            loss = app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"]
            loss.backward()
            vis = [p for n, p in named.items() if ".visual." in n]
            assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in vis) and any(float(p.grad.abs().max()) > 0 for p in vis)
            assert any(not torch.equal(b, stat0[n]) for n, b in app.named_buffers() if n in stat0)      # the statistics moved
            app.eval()
        with torch.no_grad():                                       # single-modality calls
            only = app({"pixel_values": px}, feat=True)
            # (after a training forward the running statistics have moved: compare with the oracle only for the frozen model)
            assert only["text_embeds"] is None and bool(torch.isfinite(only["image_embeds"]).all())
            if frozen:
                assert float((only["image_embeds"].cpu() - img).abs().max()) < (2e-5 if f32 else 2e-2)


def test_from_config_initialises_the_resnet_tower():
    """CLIPApp.from_config with a `vision_layers` tuple (round 4): the ModifiedResNet parameters are initialised as
    CHINESE_CLIP.initialize_parameters does (modeling_chineseclip.py:323-334: attnpool projections at in_features^-0.5, bn3 gains
    zero) instead of staying all-zero -- the image embeddings are finite unit vectors that depend on the image -- and the tower's
    output equals the oracle's on those weights."""
    cfg = dict(O.CONFIGS["tiny"], vision_layers=[1, 2, 1, 1], vision_width=16, image_resolution=64)
    app = CLIPApp.from_config(cfg, seed=3, device="cuda", compute_dtype="fp32")
    app.eval()
    named = dict(app.chinese_clip.named_parameters())
    assert float(named["visual.layer2.1.bn3.weight"].abs().max()) == 0.0 and float(named["visual.layer2.1.bn1.weight"].min()) == 1.0
    assert float(named["visual.conv1.weight"].std()) > 0.05 and float(named["visual.attnpool.c_proj.weight"].std()) > 0.01
    px = torch.randn(5, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        out = app({"pixel_values": px}, feat=True)["image_embeds"].cpu()
    assert bool(torch.isfinite(out).all()) and float((out.norm(dim=-1) - 1).abs().max()) < 1e-5
    gram = out @ out.t()
    assert float((gram - torch.eye(5)).abs().max()) > 1e-4          # not one constant vector ...
    assert float(gram.min()) < 0.999999                              # ... and not all rows identical
    sd = {n: t.detach().cpu() for n, t in list(app.chinese_clip.named_parameters()) + list(app.chinese_clip.named_buffers())
          if n.startswith("visual.") and not n.endswith("num_batches_tracked")}
    with torch.no_grad():
        ref = norm(RO.modified_resnet_forward(sd, cfg["vision_layers"], 16, px))
    assert float((out - ref).abs().max()) < 2e-5
