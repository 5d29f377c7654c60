"""ModifiedResNet tower, host side (CPU): the library's parameter table, the drop-in CLIPApp's state dict for a `vision_layers`
tuple against the oracle's table and -- where the checkout is present -- against the REAL reference CLIPApp loaded from the same
checkpoint directory (keys, shapes, cross-loading).  The numerics run on the GPU: tests/test_resnet_gpu.py."""
import json
import os

import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.clip import CLIPApp
from easynlp_amd.appzoo.clip.rn_tower import RnEngine
from oracle import clip_oracle as O
from oracle import ref_harness as R
from oracle import resnet_oracle as RO

LAYERS, WIDTH, RES = (1, 2, 1, 1), 16, 64


def rn_cfg():
    return dict(O.CONFIGS["tiny"], vision_layers=list(LAYERS), vision_width=WIDTH, image_resolution=RES)


def rn_state_dict(cfg, seed=5):
    sd = {k: v for k, v in O.make_state_dict(O.CONFIGS["tiny"], seed).items() if not k.startswith("visual.")}
    sd.update(RO.make_state_dict(cfg["vision_layers"], cfg["vision_width"], cfg["embed_dim"], cfg["image_resolution"], seed))
    return sd


def test_library_parameter_table_is_the_reference_modules():
    for layers, width, e, res in ((LAYERS, WIDTH, 64, RES), ((3, 4, 6, 3), 64, 1024, 224), ((2, 1, 2, 1), 8, 32, 96)):
        eng = RnEngine(layers, width, e, res, L.DTYPE_BF16)
        assert eng.shapes == RO.param_shapes(layers, width, e, res)
    with pytest.raises(L.EzclipError):
        RnEngine((1, 1, 1), 16, 64, 64, L.DTYPE_F32)              # four stages
    with pytest.raises(L.EzclipError):
        RnEngine((1, 1, 1, 1), 16, 64, 70, L.DTYPE_F32)           # resolution % 32


def test_dropin_state_dict_round_trip_and_trainable_or_frozen_tower(tmp_path):
    cfg = rn_cfg()
    sd = rn_state_dict(cfg)
    R.write_checkpoint_dir(str(tmp_path), cfg, sd)
    app = CLIPApp(str(tmp_path))
    own = {k[len("chinese_clip."):]: v for k, v in app.state_dict().items()}
    for k, v in sd.items():
        assert torch.equal(own[k], v), k
    extra = set(own) - set(sd)
    assert all(k.endswith("num_batches_tracked") or k == "bert.embeddings.position_ids" for k in extra), sorted(extra)[:5]
    vis = {n: p for n, p in app.named_parameters() if ".visual." in n}
    assert vis and all(p.requires_grad for p in vis.values())                # round 5: the tower trains, as the reference's does ...
    frozen = CLIPApp(str(tmp_path), user_defined_parameters={"clip_rn_train": "0"})
    assert not any(p.requires_grad for n, p in frozen.named_parameters() if ".visual." in n)     # ... unless asked not to
    assert all(p.requires_grad for n, p in app.named_parameters() if ".bert.encoder." in n)
    assert "running_mean" not in "".join(vis)                                # statistics are buffers, as in nn.BatchNorm2d
    with pytest.raises(L.EzclipError):
        app.contrastive_step(torch.zeros(1, 3, RES, RES), torch.zeros(1, 8, dtype=torch.long))


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_keys_and_shapes_equal_the_reference_clipapp(tmp_path):
    cfg = rn_cfg()
    sd = rn_state_dict(cfg)
    R.write_checkpoint_dir(str(tmp_path), cfg, sd)
    ref = R.reference_clip_app(str(tmp_path))
    app = CLIPApp(str(tmp_path))
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in app.state_dict().items()}
    assert a == b
    assert {n for n, _ in ref.named_parameters()} == {n for n, _ in app.named_parameters()}
    # each loads what the other saved
    out = tmp_path / "saved"
    os.makedirs(out)
    torch.save(app.state_dict(), out / "pytorch_model.bin")
    json.dump(cfg, open(out / "config.json", "w"))
    ref2 = R.reference_clip_app(str(out))
    for k, v in ref.state_dict().items():
        assert torch.equal(ref2.state_dict()[k], v), k
