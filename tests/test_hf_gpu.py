"""huggingface_clip branch of the drop-in CLIPApp on the GPU (reference: appzoo/clip/model.py:73-104,128-150) against
the fixtures produced by the REAL reference in that mode (tools/make_golden.py:run_hf_case) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.clip import CLIPApp
from oracle import hf_clip_oracle as H
from oracle import ref_harness as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, Lq, wseed, iseed = [str(x) for x in z["meta"][:5]]
    return z, H.HF_CONFIGS[cfg_name], int(B), int(Lq), int(wseed), int(iseed)


def make_app(tmp_path, cfg, seed, dtype):
    sd = H.make_state_dict(cfg, seed)
    R.write_hf_checkpoint_dir(str(tmp_path), cfg, sd)
    app = CLIPApp(str(tmp_path), user_defined_parameters={"clip_compute_dtype": dtype}).cuda()
    assert app.model_type == "huggingface_clip"
    return app, sd


@pytest.mark.parametrize("name", ["hf_tiny_b6_l24", "hf_small_b5_l40", "hf_large_text_b24_l40"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_hf_forward_and_backward_match_reference_golden(tmp_path, name, dtype):
    z, cfg, B, Lq, wseed, iseed = load(name)
    app, sd = make_app(tmp_path, cfg, wseed, dtype)
    app.eval()      # (dropout probabilities of these fixtures are 0 anyway)
    px, ids, tt, am = H.make_inputs(cfg, B, Lq, iseed)
    out = app({"pixel_values": px, "input_ids": ids, "token_type_ids": tt, "attention_mask": am})
    loss = app.compute_loss(out, [])["loss"]
    loss.backward()
    f32 = dtype == "fp32"
    for k in ("image_embeds", "text_embeds"):
        err = float((out[k].detach().cpu() - torch.from_numpy(z[k])).abs().max())
        assert err < (1e-5 if f32 else 1e-2), (k, err)
    assert float((out["logits_per_text"].detach().cpu() - torch.from_numpy(z["logits_per_text"])).abs().max()) < (4e-4 if f32 else 0.15)
    assert abs(loss.item() - float(z["loss"])) < (1e-5 if f32 else 1.5e-2)
    params = dict(app.named_parameters())
    params["logit_scale"] = params.pop("logit_scale_param")
    # same-shape scale of the text tower for the bf16 bound
    scale = {}
    for key in z.files:
        if key.startswith(("grad/", "gnorm/")):
            n = key.split("/", 1)[1]
            gn = float(np.linalg.norm(z[key].astype(np.float64))) if key.startswith("grad/") else float(z[key])
            sk = (n.split(".")[0], tuple(params[n].shape))
            scale[sk] = max(scale.get(sk, 0.0), gn)
    bad, seen = [], 0
    for key in z.files:
        if "/" not in key:
            continue
        kind, n = key.split("/", 1)
        p = params[n]
        if kind == "nograd":
            assert n.startswith("vision_encoder.") and p.grad is None, n      # vision_outputs[1].detach()
            continue
        if kind not in ("grad", "gnorm"):
            continue
        seen += 1
        assert p.grad is not None, n
        floor = 0.0 if f32 else 2e-2 * scale[(n.split(".")[0], tuple(p.shape))]
        if kind == "grad":
            ref = torch.from_numpy(z[key]).double().reshape(p.shape)
            err = float((p.grad.detach().cpu().double() - ref).norm())
            if err > (2e-4 if f32 else 6e-2) * float(ref.norm()) + floor + 1e-7:
                bad.append((n, err, float(ref.norm())))
        else:
            ref, got = float(z[key]), float(p.grad.double().norm())
            if abs(got - ref) > (2e-4 if f32 else 6e-2) * ref + floor + 1e-7:
                bad.append((n, got, ref))
    assert seen > 30 and not bad, bad[:10]
    pad = cfg["text_config"]["pad_token_id"]
    assert float(params["text_encoder.embeddings.word_embeddings.weight"].grad[pad].abs().max()) == 0.0
    assert float(params["text_encoder.embeddings.position_embeddings.weight"].grad[pad].abs().max()) == 0.0


def test_hf_parameter_updates_reach_the_packed_copies(tmp_path):
    """in_proj = [q; k; v] and the transposed projections are derived copies: an in-place update of a source parameter
    (optimizer step, load_state_dict) must show up in the next forward."""
    cfg = H.HF_CONFIGS["hf_tiny"]
    app, sd = make_app(tmp_path, cfg, 7, "fp32")
    app.eval()
    px, ids, tt, am = H.make_inputs(cfg, 3, 10, 1)
    batch = lambda: {"pixel_values": px, "input_ids": ids, "token_type_ids": tt, "attention_mask": am}   # noqa: E731
    with torch.no_grad():
        a = app(batch(), feat=True)
        ref = H.hf_clip_forward(sd, cfg, px, ids, tt, am)
        assert float((a["image_embeds"].cpu() - ref["image_embeds"]).abs().max()) < 2e-5
        assert float((a["text_embeds"].cpu() - ref["text_embeds"]).abs().max()) < 2e-5
        g = torch.Generator().manual_seed(0)
        sd2 = dict(sd)
        for k in ("vision_encoder.vision_model.encoder.layers.0.self_attn.k_proj.weight", "vision_projection.weight",
                  "text_projection.weight", "text_encoder.pooler.dense.bias"):
            sd2[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=g)
            dict(app.named_parameters())[k].copy_(sd2[k].cuda())
        b = app(batch(), feat=True)
        ref2 = H.hf_clip_forward(sd2, cfg, px, ids, tt, am)
        assert float((ref2["image_embeds"] - ref["image_embeds"]).abs().max()) > 1e-3
        assert float((b["image_embeds"].cpu() - ref2["image_embeds"]).abs().max()) < 2e-5
        assert float((b["text_embeds"].cpu() - ref2["text_embeds"]).abs().max()) < 2e-5
    # single-modality calls, and the reference's KeyError when the tokenizer outputs are missing
    with torch.no_grad():
        only_img = app({"pixel_values": px}, feat=True)
        assert only_img["text_embeds"] is None
        with pytest.raises(KeyError):
            app({"input_ids": ids}, feat=True)
    sd_out = app.state_dict()
    assert set(sd_out) == set(H.param_shapes(cfg)) | {"text_encoder.embeddings.position_ids",
                                                      "vision_encoder.vision_model.embeddings.position_ids"}


def test_hf_predictor_and_evaluator_surface(tmp_path):
    from easynlp_amd.appzoo.clip import CLIPPredictor
    cfg = H.HF_CONFIGS["hf_tiny"]
    sd = H.make_state_dict(cfg, 3)
    R.write_hf_checkpoint_dir(str(tmp_path), cfg, sd)
    pred = CLIPPredictor(str(tmp_path), first_sequence="text", second_sequence="image", sequence_length=12)
    out = pred.run([{"text": "tok7 tok9 tok11"}, {"text": "tok8"}])
    assert len(out) == 2 and "text_feat" in out[0]
    v = np.array([float(x) for x in out[0]["text_feat"].split("\t")])
    assert abs(np.linalg.norm(v) - 1.0) < 1e-3


def test_hf_fast_path_contrastive_step_matches_golden(tmp_path):
    """contrastive_step (no autograd bookkeeping) in huggingface_clip mode: loss and gradients in the reference
    parameters' .grad, the projections through the transposed scratch."""
    z, cfg, B, Lq, wseed, iseed = load("hf_tiny_b6_l24")
    app, _ = make_app(tmp_path, cfg, wseed, "fp32")
    app.eval()
    px, ids, tt, am = H.make_inputs(cfg, B, Lq, iseed)
    loss = app.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=True, token_type_ids=tt.cuda(),
                                attention_mask=am.cuda())
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    params = dict(app.named_parameters())
    params["logit_scale"] = params.pop("logit_scale_param")
    n_checked = 0
    for key in z.files:
        if key.startswith("grad/"):
            n = key[len("grad/"):]
            ref = torch.from_numpy(z[key]).double().reshape(params[n].shape)
            err = float((params[n].grad.detach().cpu().double() - ref).norm())
            assert err <= 2e-4 * float(ref.norm()) + 1e-7, (n, err)
            n_checked += 1
        elif key.startswith("nograd/"):
            assert params[key[len("nograd/"):]].grad is None
    assert n_checked > 30


@pytest.mark.parametrize("path", ["fused", "autograd"])
@pytest.mark.parametrize("cfg_name", ["hf_tiny", "hf_small"])
def test_hf_packed_text_tower_with_dropout_equals_the_padded_one(tmp_path, path, cfg_name):
    """huggingface_clip flavour as the reference trains it (RoBERTa's train-mode dropout armed, appzoo/clip/model.py:128-144):
    the packed text tower -- explicit position / type / mask tensors read through the row map, pooler head on the gathered
    CLS rows -- regenerates exactly the padded run's dropout decisions: same embeddings, same loss, same gradients up to the
    summation order of the weight-gradient partials.  (Round 2 left this combination on padded rows because it had never run
    on hardware.)"""
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in H.HF_CONFIGS[cfg_name].items()}
    cfg["text_config"].update(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    app, sd = make_app(tmp_path, cfg, 6, "bf16")
    app.train()
    eng = app._engine
    assert eng.pack_hf_dropout
    B, Lq = 12, 40
    pad = cfg["text_config"]["pad_token_id"]
    px, ids, tt, am = H.make_inputs(cfg, B, Lq, 3)
    g = torch.Generator().manual_seed(8)
    lens = torch.randint(1, Lq, (B,), generator=g)
    lens[0], lens[1] = Lq - 3, 1
    keep = torch.arange(Lq)[None, :] < lens[:, None]
    ids = torch.where(keep, ids.clamp(min=2), torch.full_like(ids, pad))
    am = keep.long()
    tt = tt * am
    batch = lambda: {"pixel_values": px, "input_ids": ids.clone(), "token_type_ids": tt.clone(), "attention_mask": am.clone()}  # noqa: E731
    res, emb = {}, {}
    for pack in (False, True):
        eng.pack_text = pack
        torch.manual_seed(1234)
        with torch.no_grad():
            emb[pack] = app(dict(batch(), pixel_values=None), feat=True)["text_embeds"].float().cpu()
        assert (eng.last_text_rows[0] < eng.last_text_rows[1]) == pack, eng.last_text_rows
    assert float((emb[True] - emb[False]).abs().max()) < 1e-3, float((emb[True] - emb[False]).abs().max())
    for pack in (False, True):
        eng.pack_text = pack
        for p in app.parameters():
            p.grad = None
        torch.manual_seed(1234)
        if path == "fused":
            loss = app.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=True, zero_grad=True,
                                        token_type_ids=tt.cuda(), attention_mask=am.cuda())
        else:
            loss = app.compute_loss(app(batch()), [])["loss"]
            loss.backward()
        torch.cuda.synchronize()
        rows = eng.last_text_rows
        assert (rows[0] < rows[1]) == pack, rows
        res[pack] = (float(loss.item()), {n: p.grad.detach().float().cpu().clone() for n, p in app.named_parameters()
                                          if p.grad is not None})
    assert abs(res[True][0] - res[False][0]) < 1e-3, (res[True][0], res[False][0])
    assert set(res[True][1]) == set(res[False][1]) and len(res[True][1]) > 20
    floor = 1e-3 * max(float(v.norm()) for v in res[False][1].values())
    worst = max((float((res[True][1][n] - v).norm()) / (float(v.norm()) + floor), n) for n, v in res[False][1].items()
                if not n.endswith("k_proj.bias") and not n.endswith(".key.bias"))
    assert worst[0] < 2e-2, worst
    torch.manual_seed(99)           # the dropout really was on
    other = float(app.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=False,
                                       token_type_ids=tt.cuda(), attention_mask=am.cuda()).item())
    assert abs(other - res[True][0]) > 1e-4
    eng.pack_text = True


HF_VITL14_SHALLOW = dict(
    text_config=dict(vocab_size=21128, hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12,
                     max_position_embeddings=512, type_vocab_size=2, pad_token_id=1, layer_norm_eps=1e-12, hidden_act="gelu",
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
    vision_config=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=3, num_attention_heads=16, image_size=224,
                       patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5),
    projection_dim=768)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_hf_vitl14_width_towers_against_oracle(tmp_path, dtype):
    """The pai-clip-commercial-large SHAPES of BASELINE.json config 5 (CLIPVisionModel: width 1024, 16 heads, 257 tokens, patch 14,
    projection 768, detached; RobertaModel pooled output), three / two blocks deep: forward, loss and every gradient the
    reference's backward produces, against the oracle (appzoo/clip/model.py:128-150)."""
    cfg = HF_VITL14_SHALLOW
    app, sd = make_app(tmp_path, cfg, 17, dtype)
    app.eval()
    B, Lq = 16, 64
    px, ids, tt, am = H.make_inputs(cfg, B, Lq, 9)
    out = app({"pixel_values": px, "input_ids": ids, "token_type_ids": tt, "attention_mask": am})
    loss = app.compute_loss(out, [])["loss"]
    loss.backward()
    ref, ref_loss, grads = H.forward_loss_backward(sd, cfg, px, ids, tt, am)
    f32 = dtype == "fp32"
    for k in ("image_embeds", "text_embeds"):
        got = out[k].detach().cpu()
        assert float((got - ref[k]).abs().max()) < (2e-5 if f32 else 1e-2), k
        assert float(torch.nn.functional.cosine_similarity(got, ref[k]).min()) > (0.999999 if f32 else 0.9995), k
    assert abs(loss.item() - ref_loss.item()) < (2e-5 if f32 else 1.5e-2)
    params = dict(app.named_parameters())
    params["logit_scale"] = params.pop("logit_scale_param")
    worst, frozen = [], 0
    for n, want in grads.items():
        p = params[n]
        if want is None:
            frozen += 1
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        worst.append((float((p.grad.detach().cpu().double().reshape(want.shape) - want.double()).norm()), n,
                      float(want.double().norm()), (n.split(".")[0], tuple(want.shape))))
    by_shape = {}                                # (same bound as test_model_gpu.py::test_vitl14_width_towers_against_oracle)
    for _, _, wn, key in worst:
        by_shape[key] = max(by_shape.get(key, 0.0), wn)
    rel, floor = (3e-4, 1e-5) if f32 else (6e-2, 2e-2)
    worst = sorted(((err / (rel * wn + floor * by_shape[key] + 1e-12), n, err, wn) for err, n, wn, key in worst), reverse=True)
    print("worst gradient deviations (fraction of bound, name, |diff|, |ref|):", worst[:5])
    assert frozen > 40 and len(worst) > 30       # (the whole vision encoder is detached: model.py:140)
    assert worst[0][0] < 1.0, worst[:5]
