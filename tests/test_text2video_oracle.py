"""Pin oracle/text2video_oracle.py (Text2VideoRetrieval: OPEN_CLIP per frame + masked mean pooling) against fixtures of the
REAL reference application (tools/make_golden.py: run_t2v_case) and, when present, the live reference; and run the drop-in's
host logic (frame reshape, mask pooling, output contract) on the CPU with the tower encodes supplied by the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import clip_oracle as O
from oracle import open_clip_oracle as OC
from oracle import ref_harness as R
from oracle import text2video_oracle as TV

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, T, wseed, iseed = [str(x) for x in z["meta"][:5]]
    return z, OC.OPENCLIP_CONFIGS[cfg_name], int(B), int(T), int(wseed), int(iseed)


@pytest.mark.parametrize("name", ["t2v_tiny_b4_t3", "t2v_small_b3_t5"])
def test_text2video_oracle_matches_reference_golden(name):
    z, cfg, B, T, wseed, iseed = load(name)
    sd = OC.make_state_dict(cfg, wseed)
    px, masks, ids = TV.make_inputs(cfg, B, T, iseed)
    assert int(masks[0].sum()) == T and int(masks[1].sum()) == 1
    out, loss, grads = TV.forward_loss_backward(sd, cfg, px, masks, ids)
    np.testing.assert_allclose(out["video_embeds"].numpy(), z["video_embeds"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["text_embeds"].numpy(), z["text_embeds"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["logits_per_text"].numpy(), z["logits_per_text"], atol=5e-5, rtol=0)
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    seen = 0
    for key in z.files:
        if key.startswith("gnorm/"):
            n = key[len("gnorm/"):]
            assert abs(float(grads[n].double().norm()) - float(z[key])) <= 1e-4 * float(z[key]) + 1e-7, n
            seen += 1
        assert not key.startswith("nograd/"), key
    assert seen == len(OC.param_shapes(cfg))


def test_mean_pooling_edge_cases_follow_the_reference():
    from easynlp_amd.appzoo.text2video_retrieval.model import mean_pooling_for_similarity_visual as mp
    x = torch.randn(3, 4, 8)
    m = torch.tensor([[1, 1, 1, 1], [1, 0, 0, 0], [0, 0, 0, 0]])
    got = mp(x, m)
    assert torch.allclose(got[0], x[0].mean(0), atol=1e-6) and torch.equal(got[1], x[1, 0])
    assert torch.equal(got[2], torch.zeros(8))                      # no valid frame: 0 / 1 (model.py:105)
    assert torch.equal(got, TV.mean_pooling(x, m))
    if R.reference_available():
        R.install_shims()
        from easynlp.appzoo.text2video_retrieval.model import Text2VideoRetrieval
        assert torch.equal(got, Text2VideoRetrieval._mean_pooling_for_similarity_visual(None, x, m))


def test_dropin_host_logic_with_oracle_encodes():
    """Text2VideoRetrieval.forward on the CPU with ``encode`` replaced by the oracle's towers: everything between the two
    library calls (5-d input -> B*T frames, device moves, pooling, renormalisation, output keys, in-place mutation of
    ``inputs`` as the reference does) against the golden of the real application."""
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.text2video_retrieval import Text2VideoRetrieval
    z, cfg, B, T, wseed, iseed = load("t2v_small_b3_t5")
    sd = OC.make_state_dict(cfg, wseed)
    px, masks, ids = TV.make_inputs(cfg, B, T, iseed)
    app = Text2VideoRetrieval(None)
    app._params = {"text_projection": torch.nn.Parameter(sd["text_projection"].clone()),
                   "logit_scale": torch.nn.Parameter(sd["logit_scale"].clone())}
    seen = {}

    def encode(pixel_values=None, input_ids=None):
        seen["px"] = None if pixel_values is None else tuple(pixel_values.shape)
        img = None if pixel_values is None else O.l2_normalize(O.vit_forward(sd, OC.chinese_style_config(cfg), pixel_values))
        txt = None if input_ids is None else O.l2_normalize(OC.text_forward(sd, cfg, input_ids))
        return img, txt
    app.encode = encode
    inputs = {"pixel_values": px.clone(), "video_masks": masks.clone(), "input_ids": ids.clone()}
    with torch.no_grad():
        out = app(inputs, feat=True)
    Rr = cfg["image_resolution"]
    assert seen["px"] == (B * T, 3, Rr, Rr) and tuple(inputs["pixel_values"].shape) == (B * T, 3, Rr, Rr)
    assert set(out) == {"video_embeds", "text_embeds"}
    np.testing.assert_allclose(out["video_embeds"].numpy(), z["video_embeds"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["text_embeds"].numpy(), z["text_embeds"], atol=2e-6, rtol=0)
    with torch.no_grad():
        only_text = app({"input_ids": ids.clone()}, feat=True)
    assert only_text["video_embeds"] is None and seen["px"] is None
    with pytest.raises(L.EzclipError):
        app({"pixel_values": px[:, 0], "video_masks": masks, "input_ids": ids}, feat=True)     # 4-d: not a clip batch
    with pytest.raises(L.EzclipError):
        app({"pixel_values": px.clone(), "input_ids": ids}, feat=True)                         # masks missing
    with pytest.raises(L.EzclipError):
        app({"pixel_values": px.clone(), "video_masks": masks[:, :2], "input_ids": ids}, feat=True)
    with pytest.raises(L.EzclipError):
        app({}, feat=True)


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_text2video_oracle_matches_live_reference_app(tmp_path):
    R.install_shims()
    from easynlp.appzoo.text2video_retrieval.model import Text2VideoRetrieval
    cfg = OC.OPENCLIP_CONFIGS["oc_tiny"]
    sd = OC.make_state_dict(cfg, 5)
    with open(os.path.join(str(tmp_path), "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"open_clip." + k: v for k, v in sd.items()}, os.path.join(str(tmp_path), "pytorch_model.bin"))
    app = Text2VideoRetrieval(str(tmp_path)).eval()
    px, masks, ids = TV.make_inputs(cfg, 3, 4, 11)
    with torch.no_grad():
        ref = app({"pixel_values": px.clone(), "video_masks": masks.clone(), "input_ids": ids.clone()})
        mine = TV.forward(sd, cfg, px, masks, ids)
    for k in ("text_embeds", "video_embeds", "logits_per_text"):
        assert float((ref[k] - mine[k]).abs().max()) < 5e-5, k
