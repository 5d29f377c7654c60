"""Pin oracle/hf_clip_oracle.py (huggingface_clip branch of the reference CLIPApp) against the committed fixtures the
REAL reference produced (tools/make_golden.py:run_hf_case) and, when /root/reference is present, against the live
reference CLIPApp loaded from a synthetic checkpoint directory."""
import os

import numpy as np
import pytest
import torch

from oracle import hf_clip_oracle as H
from oracle import ref_harness as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, L, wseed, iseed = [str(x) for x in z["meta"][:5]]
    return z, H.HF_CONFIGS[cfg_name], int(B), int(L), int(wseed), int(iseed)


@pytest.mark.parametrize("name", ["hf_tiny_b6_l24", "hf_small_b5_l40", "hf_large_text_b24_l40"])
def test_hf_oracle_matches_reference_golden(name):
    z, cfg, B, L, wseed, iseed = load(name)
    sd = H.make_state_dict(cfg, wseed)
    px, ids, tt, am = H.make_inputs(cfg, B, L, iseed)
    out, loss, grads = H.forward_loss_backward(sd, cfg, px, ids, tt, am)
    np.testing.assert_allclose(out["image_embeds"].numpy(), z["image_embeds"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["text_embeds"].numpy(), z["text_embeds"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["logits_per_text"].numpy(), z["logits_per_text"], atol=1e-4, rtol=0)
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    seen = 0
    for key in z.files:
        n = key.split("/", 1)[1] if "/" in key else None
        if key.startswith("nograd/"):
            # the vision tower is detached (model.py:140): no gradient below the projection
            assert n.startswith("vision_encoder.") and (grads[n] is None or float(grads[n].abs().max()) == 0.0), n
        elif key.startswith("grad/"):
            ref = torch.from_numpy(z[key]).reshape(grads[n].shape)
            assert float((grads[n] - ref).norm()) <= 1e-4 * float(ref.norm()) + 1e-7, n
            seen += 1
        elif key.startswith("gnorm/"):
            assert abs(float(grads[n].double().norm()) - float(z[key])) <= 1e-4 * float(z[key]) + 1e-7, n
            seen += 1
    assert seen > 30
    # the padding row of the word / position embeddings gets no gradient (nn.Embedding padding_idx = pad_token_id)
    pad = cfg["text_config"]["pad_token_id"]
    assert float(grads["text_encoder.embeddings.word_embeddings.weight"][pad].abs().max()) == 0.0
    assert float(grads["text_encoder.embeddings.position_embeddings.weight"][pad].abs().max()) == 0.0
    assert float(grads["text_encoder.embeddings.token_type_embeddings.weight"].abs().min()) > 0.0


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_hf_oracle_matches_live_reference_app(tmp_path):
    R.install_shims()
    from easynlp.appzoo.clip.model import CLIPApp
    cfg = H.HF_CONFIGS["hf_small"]
    sd = H.make_state_dict(cfg, 5)
    R.write_hf_checkpoint_dir(str(tmp_path), cfg, sd)
    app = CLIPApp(str(tmp_path))
    assert app.model_type == "huggingface_clip"
    app.eval()
    px, ids, tt, am = H.make_inputs(cfg, 3, 17, 11)
    with torch.no_grad():
        ref = app({"pixel_values": px, "input_ids": ids, "token_type_ids": tt, "attention_mask": am})
        mine = H.hf_clip_forward(sd, cfg, px, ids, tt, am)
    for k in ("text_embeds", "image_embeds"):
        assert float((ref[k] - mine[k]).abs().max()) < 2e-6, k
    assert set(app.state_dict()) - {"text_encoder.embeddings.position_ids", "vision_encoder.vision_model.embeddings.position_ids"} \
        == set(H.param_shapes(cfg))
