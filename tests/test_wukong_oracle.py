"""Pin oracle/wukong_oracle.py (WukongCLIP: open_clip-style towers, LayerNorm eps 1e-7, feature of the token 102) against the
fixtures the REAL reference application produced (tools/make_golden.py: run_wukong_case) and, when the checkout is present,
against the live reference; plus the host-side name / config maps of the drop-in (no GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import open_clip_oracle as OC
from oracle import ref_harness as R
from oracle import wukong_oracle as WK

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, wseed, iseed = [str(x) for x in z["meta"][:4]]
    return z, WK.WUKONG_CONFIGS[cfg_name], int(B), int(wseed), int(iseed)


@pytest.mark.parametrize("name", ["wukong_tiny_b6", "wukong_small_b5"])
def test_wukong_oracle_matches_reference_golden(name):
    z, cfg, B, wseed, iseed = load(name)
    sd = WK.make_state_dict(cfg, wseed)
    px, ids = WK.make_inputs(cfg, B, iseed)
    assert bool(((ids == 102).sum(1) == 1).all())
    out, loss, grads = WK.forward_loss_backward(sd, cfg, px, ids)
    np.testing.assert_allclose(out["image_features"].numpy(), z["image_features"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["text_features"].numpy(), z["text_features"], atol=2e-6, rtol=0)
    assert abs(float(out["logit_scale"]) - float(z["logit_scale"])) < 1e-5
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    seen = 0
    for key in z.files:
        if key.startswith("grad/"):
            n = key[len("grad/"):]
            ref = torch.from_numpy(z[key]).reshape(grads[n].shape)
            assert float((grads[n] - ref).norm()) <= 1e-4 * float(ref.norm()) + 1e-7, n
            seen += 1
        elif key.startswith("gnorm/"):
            n = key[len("gnorm/"):]
            assert abs(float(grads[n].double().norm()) - float(z[key])) <= 1e-4 * float(z[key]) + 1e-7, n
            seen += 1
        elif key.startswith("nograd/"):
            raise AssertionError("every WukongModel parameter trains: " + key)
    assert seen == len(WK.param_shapes(cfg))


def test_fixture_discriminates_eps_and_pooling():
    """The golden would not match with the default eps (1e-5) or with argmax pooling: both differences are observable."""
    z, cfg, B, wseed, iseed = load("wukong_tiny_b6")
    sd = WK.make_state_dict(cfg, wseed)
    px, ids = WK.make_inputs(cfg, B, iseed)
    s = {WK.to_open_clip_name(n): v for n, v in sd.items()}
    oc = WK.open_clip_style_config(cfg)
    with torch.no_grad():
        good = OC.text_forward(s, oc, ids)
        eps5 = OC.text_forward(s, dict(oc, block_ln_eps=1e-5), ids)
        amax = OC.text_forward(s, {k: v for k, v in oc.items() if k != "eot_id"}, ids)
        from oracle import clip_oracle as O
        ref = torch.from_numpy(z["text_features"])
        assert float((O.l2_normalize(good) - ref).abs().max()) < 2e-6
        assert float((O.l2_normalize(eps5) - ref).abs().max()) > 1e-3
        assert float((O.l2_normalize(amax) - ref).abs().max()) > 1e-2
        vi = O.l2_normalize(O.vit_forward(s, dict(OC.chinese_style_config(oc), block_ln_eps=1e-5), px))
        assert float((vi - torch.from_numpy(z["image_features"])).abs().max()) > 1e-3


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_wukong_oracle_matches_live_reference_app(tmp_path):
    R.install_shims()
    from easynlp.appzoo.wukong_clip.model import WukongCLIP
    cfg = WK.WUKONG_CONFIGS["wk_small"]
    sd = WK.make_state_dict(cfg, 5)
    with open(os.path.join(str(tmp_path), "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save(sd, os.path.join(str(tmp_path), "pytorch_model.bin"))
    app = WukongCLIP(str(tmp_path)).eval()
    px, ids = WK.make_inputs(cfg, 3, 11)
    with torch.no_grad():
        ref, _ = app({"pixel_values": px, "input_ids": ids})
        mine = WK.wukong_forward(sd, cfg, px, ids)
    for k in ("text_features", "image_features"):
        assert float((ref[k] - mine[k]).abs().max()) < 2e-6, k


def test_dropin_name_and_config_maps_cover_the_reference_state_dict():
    """host side of the drop-in, no GPU: library names <-> WukongCLIP.state_dict() keys, config.json -> ezclip_config"""
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.wukong_clip import model as M
    cfg = WK.WUKONG_CONFIGS["wk_small"]
    lc = M.library_config(cfg)
    assert lc["text_hidden_size"] == 128 and lc["vision_width"] == 192 and lc["text_max_position_embeddings"] == 32
    assert lc["text_intermediate_size"] == 512 and lc["embed_dim"] == 128 and lc["text_num_attention_heads"] == 2
    oc_names = list(OC.param_shapes(WK.open_clip_style_config(cfg)))
    mapped = {"model." + M.reference_name(n) for n in oc_names}
    assert mapped == set(WK.param_shapes(cfg))
    for n in oc_names:
        assert WK.to_open_clip_name("model." + M.reference_name(n)) == n
    bad = json.loads(json.dumps(cfg))
    bad["model"]["text"]["heads"] = 4
    with pytest.raises(L.EzclipError):
        M.library_config(bad)
    with pytest.raises(L.EzclipError):
        M.library_config({"model_type": "chinese_clip"})
