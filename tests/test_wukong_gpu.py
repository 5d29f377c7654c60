"""Drop-in WukongCLIP on the GPU (reference: appzoo/wukong_clip/model.py:8-73; WukongModel, modeling_wukong.py:238-433)
against the fixtures produced by the REAL reference application (tools/make_golden.py: run_wukong_case)."""
import json
import os

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.wukong_clip import WukongCLIP, WukongCLIPEvaluator
from oracle import wukong_oracle as WK

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, wseed, iseed = [str(x) for x in z["meta"][:4]]
    return z, WK.WUKONG_CONFIGS[cfg_name], int(B), int(wseed), int(iseed)


def make_app(tmp_path, cfg, seed, dtype):
    sd = WK.make_state_dict(cfg, seed)
    with open(os.path.join(str(tmp_path), "config.json"), "w") as f:
        json.dump(cfg, f)
    torch.save(sd, os.path.join(str(tmp_path), "pytorch_model.bin"))
    app = WukongCLIP(str(tmp_path), user_defined_parameters={"clip_compute_dtype": dtype}).cuda()
    assert sorted(app.state_dict()) == sorted(sd)                     # the reference's keys, nothing else
    return app, sd


@pytest.mark.parametrize("name", ["wukong_tiny_b6", "wukong_small_b5"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_wukong_forward_and_backward_match_reference_golden(tmp_path, name, dtype):
    z, cfg, B, wseed, iseed = load(name)
    app, sd = make_app(tmp_path, cfg, wseed, dtype)
    app.train()
    px, ids = WK.make_inputs(cfg, B, iseed)
    out, extra = app({"pixel_values": px, "input_ids": ids})
    assert extra == []
    loss = app.compute_loss(out, [])["loss"]
    loss.backward()
    f32 = dtype == "fp32"
    for k in ("image_features", "text_features"):
        err = float((out[k].detach().cpu() - torch.from_numpy(z[k])).abs().max())
        assert err < (1e-5 if f32 else 1.5e-2), (k, err)
    assert abs(float(out["logit_scale"].detach()) - float(z["logit_scale"])) < 1e-4
    assert abs(loss.item() - float(z["loss"])) < (1e-5 if f32 else 1.5e-2)
    params = dict(app.named_parameters())
    tower = lambda n: "visual" if ".visual_encoder." in n else "text"       # noqa: E731
    scale = {}
    for key in z.files:
        if key.startswith(("grad/", "gnorm/")):
            n = key.split("/", 1)[1]
            gn = float(np.linalg.norm(z[key].astype(np.float64))) if key.startswith("grad/") else float(z[key])
            sk = (tower(n), tuple(params[n].shape))
            scale[sk] = max(scale.get(sk, 0.0), gn)
    bad, seen = [], 0
    for key in z.files:
        if not key.startswith(("grad/", "gnorm/")):
            continue
        kind, n = key.split("/", 1)
        p = params[n]
        seen += 1
        floor = 0.0 if f32 else 2e-2 * scale[(tower(n), tuple(p.shape))]
        if kind == "grad":
            ref = torch.from_numpy(z[key]).double().reshape(p.shape)
            err = float((p.grad.detach().cpu().double() - ref).norm())
            if err > (2e-4 if f32 else 6e-2) * float(ref.norm()) + floor + 1e-7:
                bad.append((n, err, float(ref.norm())))
        else:
            ref, got = float(z[key]), float(p.grad.double().norm())
            if abs(got - ref) > (2e-4 if f32 else 6e-2) * ref + floor + 1e-7:
                bad.append((n, got, ref))
    assert seen == len(WK.param_shapes(cfg)) and not bad, bad[:10]


def test_wukong_inference_fast_path_evaluator_and_contract(tmp_path):
    cfg = WK.WUKONG_CONFIGS["wk_small"]
    app, sd = make_app(tmp_path, cfg, 3, "bf16")
    app.eval()
    px, ids = WK.make_inputs(cfg, 8, 1)
    with torch.no_grad():
        ref = WK.wukong_forward(sd, cfg, px, ids)
        out, _ = app({"pixel_values": px, "input_ids": ids})
        assert float((out["text_features"].cpu() - ref["text_features"]).abs().max()) < 1.5e-2
        assert float((out["image_features"].cpu() - ref["image_features"]).abs().max()) < 1.5e-2
        loss = app.compute_loss(out, [])["loss"]
        assert abs(loss.item() - WK.compute_loss(ref).item()) < 2e-2
        fused = app.contrastive_step(px.cuda(), ids.cuda(), process_group=False)
        assert abs(fused.item() - loss.item()) < 2e-3
        # causal tower: what follows the token 102 cannot influence its row
        ids2 = ids.clone()
        for b in range(ids.shape[0]):
            t = int((ids[b] == 102).nonzero()[0])
            ids2[b, t + 1:] = 7
        out2, _ = app({"input_ids": ids2})
        assert out2["image_features"] is None
        assert float((out2["text_features"] - out["text_features"]).abs().max()) < 1e-6
        # a row without / with two tail tokens: the reference would return a different number of rows; here it is an error
        bad = ids.clone()
        bad[0][bad[0] == 102] = 5
        with pytest.raises(L.EzclipError):
            app({"input_ids": bad})
        bad = ids.clone()
        bad[2, 0] = 102
        with pytest.raises(L.EzclipError):
            app({"input_ids": bad})

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 8

        def __getitem__(self, i):
            return {"pixel_values": px[i], "input_ids": ids[i]}

        @staticmethod
        def batch_fn(rows):
            return {k: torch.stack([r[k] for r in rows]) for k in rows[0]}

    ev = WukongCLIPEvaluator(DS(), user_defined_parameters={}, eval_batch_size=8)
    res = ev.evaluate(app)
    from oracle import clip_oracle as O
    want = O.recall_at_k(out["text_features"].cpu().float(), out["image_features"].cpu().float())
    assert res[0][0] == "mean_recall" and abs(res[0][1] - want[0]) < 1e-9
    assert WukongCLIPEvaluator(DS(), user_defined_parameters={"cosine_similarity": "True"}).evaluate(app) is None


def test_wukong_trains_through_autograd_and_fast_path_agree(tmp_path):
    """loss.backward() through forward / compute_loss and contrastive_step(backward=True) leave the same gradients"""
    cfg = WK.WUKONG_CONFIGS["wk_tiny"]
    app, sd = make_app(tmp_path, cfg, 11, "fp32")
    app.train()
    px, ids = WK.make_inputs(cfg, 6, 2)
    out, _ = app({"pixel_values": px, "input_ids": ids})
    app.compute_loss(out, [])["loss"].backward()
    g1 = {n: p.grad.clone() for n, p in app.named_parameters()}
    app.zero_grad(set_to_none=True)
    app.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=True)
    for n, p in app.named_parameters():
        assert float((p.grad - g1[n]).norm()) <= 1e-4 * float(g1[n].norm()) + 1e-7, n
