"""open_clip branch of the drop-in CLIPApp on the GPU (reference: appzoo/clip/model.py:56-64,124-125; OPEN_CLIP,
modeling_openclip.py:255-385) against the fixtures produced by the REAL reference in that mode."""
import json
import os

import numpy as np
import pytest
import torch

from easynlp_amd.appzoo.clip import CLIPApp
from oracle import open_clip_oracle as OC

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, _, wseed, iseed = [str(x) for x in z["meta"][:5]]
    return z, OC.OPENCLIP_CONFIGS[cfg_name], int(B), int(wseed), int(iseed)


def make_app(tmp_path, cfg, seed, dtype):
    sd = OC.make_state_dict(cfg, seed)
    with open(os.path.join(str(tmp_path), "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"open_clip." + k: v for k, v in sd.items()}, os.path.join(str(tmp_path), "pytorch_model.bin"))
    app = CLIPApp(str(tmp_path), user_defined_parameters={"clip_compute_dtype": dtype}).cuda()
    assert app.model_type == "open_clip"
    return app, sd


@pytest.mark.parametrize("name", ["openclip_tiny_b6", "openclip_small_b5"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_openclip_forward_and_backward_match_reference_golden(tmp_path, name, dtype):
    z, cfg, B, wseed, iseed = load(name)
    app, sd = make_app(tmp_path, cfg, wseed, dtype)
    app.train()
    px, ids = OC.make_inputs(cfg, B, iseed)
    out = app({"pixel_values": px, "input_ids": ids})
    loss = app.compute_loss(out, [])["loss"]
    loss.backward()
    f32 = dtype == "fp32"
    for k in ("image_embeds", "text_embeds"):
        err = float((out[k].detach().cpu() - torch.from_numpy(z[k])).abs().max())
        assert err < (1e-5 if f32 else 1e-2), (k, err)
    assert abs(loss.item() - float(z["loss"])) < (1e-5 if f32 else 1.5e-2)
    params = {n.replace("open_clip.", "", 1): p for n, p in app.named_parameters()}
    scale = {}
    for key in z.files:
        if key.startswith(("grad/", "gnorm/")):
            n = key.split("/", 1)[1]
            gn = float(np.linalg.norm(z[key].astype(np.float64))) if key.startswith("grad/") else float(z[key])
            sk = ("visual" if n.startswith("visual") else "text", tuple(params[n].shape))
            scale[sk] = max(scale.get(sk, 0.0), gn)
    bad, seen = [], 0
    for key in z.files:
        if not key.startswith(("grad/", "gnorm/")):
            continue
        kind, n = key.split("/", 1)
        p = params[n]
        seen += 1
        floor = 0.0 if f32 else 2e-2 * scale[("visual" if n.startswith("visual") else "text", tuple(p.shape))]
        if kind == "grad":
            ref = torch.from_numpy(z[key]).double().reshape(p.shape)
            err = float((p.grad.detach().cpu().double() - ref).norm())
            if err > (2e-4 if f32 else 6e-2) * float(ref.norm()) + floor + 1e-7:
                bad.append((n, err, float(ref.norm())))
        else:
            ref, got = float(z[key]), float(p.grad.double().norm())
            if abs(got - ref) > (2e-4 if f32 else 6e-2) * ref + floor + 1e-7:
                bad.append((n, got, ref))
    assert seen == len(OC.param_shapes(cfg)) and not bad, bad[:10]


def test_openclip_inference_path_and_fast_path(tmp_path):
    """eval() takes the bf16 inference path (folded LayerNorms in the text blocks too); contrastive_step gives the same
    loss; tokens after the EOT do not matter (causal mask)."""
    cfg = OC.OPENCLIP_CONFIGS["oc_small"]
    app, sd = make_app(tmp_path, cfg, 3, "bf16")
    app.eval()
    px, ids = OC.make_inputs(cfg, 8, 1)
    with torch.no_grad():
        ref = OC.open_clip_forward(sd, cfg, px, ids)
        out = app({"pixel_values": px, "input_ids": ids})
        assert float((out["text_embeds"].cpu() - ref["text_embeds"]).abs().max()) < 1e-2
        assert float((out["image_embeds"].cpu() - ref["image_embeds"]).abs().max()) < 1e-2
        loss = app.compute_loss(out, [])["loss"]
        fused = app.contrastive_step(px.cuda(), ids.cuda(), process_group=False)
        assert abs(fused.item() - loss.item()) < 2e-3
        ids2 = ids.clone()
        eot = ids.argmax(dim=-1)
        for b in range(ids.shape[0]):
            ids2[b, eot[b] + 1:] = 7
        out2 = app({"input_ids": ids2}, feat=True)
        assert float((out2["text_embeds"] - out["text_embeds"]).abs().max()) < 1e-6
