"""Pin oracle/open_clip_oracle.py (open_clip branch: OPEN_CLIP with the causal CLIP text transformer) against the
fixtures the REAL reference CLIPApp produced in that mode (tools/make_golden.py:run_openclip_case) and, when the checkout
is present, against the live reference."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import open_clip_oracle as OC
from oracle import ref_harness as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, _, wseed, iseed = [str(x) for x in z["meta"][:5]]
    return z, OC.OPENCLIP_CONFIGS[cfg_name], int(B), int(wseed), int(iseed)


@pytest.mark.parametrize("name", ["openclip_tiny_b6", "openclip_small_b5"])
def test_openclip_oracle_matches_reference_golden(name):
    z, cfg, B, wseed, iseed = load(name)
    sd = OC.make_state_dict(cfg, wseed)
    px, ids = OC.make_inputs(cfg, B, iseed)
    out, loss, grads = OC.forward_loss_backward(sd, cfg, px, ids)
    np.testing.assert_allclose(out["image_embeds"].numpy(), z["image_embeds"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["text_embeds"].numpy(), z["text_embeds"], atol=2e-6, rtol=0)
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
            raise AssertionError("every OPEN_CLIP parameter trains: " + key)
    assert seen == len(OC.param_shapes(cfg))
    # causal tower: tokens after the EOT cannot influence the feature -- changing the padding leaves it unchanged
    ids2 = ids.clone()
    eot = ids.argmax(dim=-1)
    for b in range(B):
        ids2[b, eot[b] + 1:] = 5
    with torch.no_grad():
        assert torch.equal(OC.text_forward(sd, cfg, ids2), OC.text_forward(sd, cfg, ids))


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_openclip_oracle_matches_live_reference_app(tmp_path):
    R.install_shims()
    from easynlp.appzoo.clip.model import CLIPApp
    cfg = OC.OPENCLIP_CONFIGS["oc_small"]
    sd = OC.make_state_dict(cfg, 5)
    with open(os.path.join(str(tmp_path), "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save({"open_clip." + k: v for k, v in sd.items()}, os.path.join(str(tmp_path), "pytorch_model.bin"))
    app = CLIPApp(str(tmp_path))
    assert app.model_type == "open_clip"
    app.eval()
    px, ids = OC.make_inputs(cfg, 3, 11)
    with torch.no_grad():
        ref = app({"pixel_values": px, "input_ids": ids})
        mine = OC.open_clip_forward(sd, cfg, px, ids)
    for k in ("text_embeds", "image_embeds"):
        assert float((ref[k] - mine[k]).abs().max()) < 2e-6, k
