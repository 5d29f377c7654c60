"""Pin the CPU oracle (oracle/clip_oracle.py) against (a) the committed golden
fixtures produced by the real reference (tools/make_golden.py) and (b) the
live reference when /root/reference is present (build container only)."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_oracle as O
from oracle import ref_harness as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, L, wseed, iseed = [str(x) for x in z["meta"][:5]]
    return z, O.CONFIGS[cfg_name], int(B), int(L), int(wseed), int(iseed)


@pytest.mark.parametrize("name", ["tiny_b6_l24", "small_b5_l40", "p14_w256_b16_l32", "vitb16_bertbase_b4_l64", "large_text_b24_l40",
                                  "vitb16_bertbase_rg03_b4_l64"])
def test_oracle_forward_matches_reference_golden(name):
    z, cfg, B, L, wseed, iseed = load(name)
    sd = O.make_state_dict(cfg, wseed, float(z["residual_gain"]) if "residual_gain" in z.files else 1.0)
    px, ids = O.make_inputs(cfg, B, L, iseed)
    with torch.no_grad():
        out = O.clip_forward(sd, cfg, px, ids)
        loss = O.clip_loss(out["logits_per_text"])
    # fp32 CPU vs fp32 CPU, different summation order only
    np.testing.assert_allclose(out["image_embeds"].numpy(), z["image_embeds"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["text_embeds"].numpy(), z["text_embeds"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["logits_per_text"].numpy(), z["logits_per_text"], atol=1e-4, rtol=0)
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * max(1.0, abs(float(z["loss"])))


@pytest.mark.parametrize("name", ["tiny_b6_l24", "small_b5_l40"])
def test_oracle_backward_matches_reference_golden(name):
    z, cfg, B, L, wseed, iseed = load(name)
    sd = O.make_state_dict(cfg, wseed)
    px, ids = O.make_inputs(cfg, B, L, iseed)
    _, loss, grads = O.forward_loss_backward(sd, cfg, px, ids)
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    for key in z.files:
        if key.startswith("nograd/"):
            n = key[len("nograd/"):]
            assert grads[n] is None or float(grads[n].abs().max()) == 0.0, n
        if key.startswith("gnorm/"):
            n = key[len("gnorm/"):]
            gn = float(grads[n].double().norm())
            ref = float(z[key])
            # key biases have a mathematically zero gradient (softmax shift
            # invariance): both sides hold ~1e-9 rounding noise there
            assert abs(gn - ref) <= 1e-4 * ref + 1e-7, (n, gn, ref)
        if key.startswith("grad/"):
            n = key[len("grad/"):]
            ref = torch.from_numpy(z[key])
            err = float((grads[n] - ref).norm())
            assert err <= 1e-4 * float(ref.norm()) + 1e-7, (n, err)


def golden_dropout(z, cfg):
    """dropout argument of the oracle rebuilt from a fixture's packed keep masks."""
    masks = {}
    for key in z.files:
        if key.startswith("mask/"):
            n = key[len("mask/"):]
            shape = tuple(int(x) for x in z["mshape/" + n])
            bits = np.unpackbits(z[key])[:int(np.prod(shape))]
            masks[n] = torch.from_numpy(bits.reshape(shape).astype(np.bool_))
    return {"p_hidden": float(z["p_hidden"]), "p_attn": float(z["p_attn"]), "masks": masks}


def test_oracle_train_mode_dropout_matches_reference_golden():
    """The reference in train mode with BERT dropout 0.1 / 0.15 (masks recorded from its own F.dropout calls,
    tools/make_golden.py:run_dropout_case): replaying the masks, the oracle reproduces embeddings, loss and every
    gradient -- i.e. the four dropout sites sit where modeling_bert.py:128,238,266,344 put them."""
    z, cfg, B, L, wseed, iseed = load("tiny_dropout_b6_l24")
    sd = O.make_state_dict(cfg, wseed)
    px, ids = O.make_inputs(cfg, B, L, iseed)
    drop = golden_dropout(z, cfg)
    assert len(drop["masks"]) == 1 + 3 * cfg["text_num_hidden_layers"]
    keep = np.mean([float(m.float().mean()) for n, m in drop["masks"].items() if not n.endswith("attn")])
    assert abs(keep - 0.9) < 0.02
    out, loss, grads = O.forward_loss_backward(sd, cfg, px, ids, dropout=drop)
    np.testing.assert_allclose(out["text_embeds"].numpy(), z["text_embeds"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out["image_embeds"].numpy(), z["image_embeds"], atol=2e-6, rtol=0)
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    # and the eval-mode output differs (the masks matter)
    z0 = np.load(os.path.join(GOLD, "tiny_b6_l24.npz"))
    assert np.abs(z["text_embeds"] - z0["text_embeds"]).max() > 1e-3
    for key in z.files:
        if key.startswith("grad/"):
            n = key[len("grad/"):]
            ref = torch.from_numpy(z[key])
            err = float((grads[n] - ref).norm())
            assert err <= 1e-4 * float(ref.norm()) + 1e-7, (n, err)


def test_oracle_recall_matches_bruteforce():
    g = torch.Generator().manual_seed(5)
    t = torch.nn.functional.normalize(torch.randn(40, 16, generator=g), dim=-1)
    i = torch.nn.functional.normalize(t + 0.7 * torch.randn(40, 16, generator=g), dim=-1)
    mean, r1, r5, r10 = O.recall_at_k(t, i)
    sim = t @ i.t()
    rank = (sim > sim.diagonal()[:, None]).sum(-1)
    assert r1 == int((rank < 1).sum()) / 40
    assert r5 == int((rank < 5).sum()) / 40
    assert r10 == int((rank < 10).sum()) / 40
    assert abs(mean - (r1 + r5 + r10) / 3) < 1e-12


def test_oracle_recall_ranks_both_directions_against_counting():
    """stable-sort position == number of strictly larger scores + equal scores at a smaller index, for rows (text -> image) and
    columns (image -> text); duplicated gallery / query rows make the ties real"""
    g = torch.Generator().manual_seed(6)
    t = torch.nn.functional.normalize(torch.randn(37, 16, generator=g), dim=-1)
    v = torch.nn.functional.normalize(t + 0.7 * torch.randn(37, 16, generator=g), dim=-1)
    v[3] = v[20]; v[21] = v[20]; t[8] = t[30]
    t2i, i2t = O.recall_ranks(t, v)
    sim = t @ v.t()
    idx = torch.arange(37)
    d = sim.diagonal()
    want_r = ((sim > d[:, None]) | ((sim == d[:, None]) & (idx[None, :] < idx[:, None]))).sum(1)
    want_c = ((sim > d[None, :]) | ((sim == d[None, :]) & (idx[:, None] < idx[None, :]))).sum(0)
    assert torch.equal(t2i, want_r) and torch.equal(i2t, want_c)
    mean, r1, r5, r10 = O.recall_at_k(t, v)
    assert abs(r1 - float((t2i < 1).sum()) / 37) < 1e-12 and abs(r10 - float((t2i < 10).sum()) / 37) < 1e-12


def test_global_loss_shards_sum_to_full_loss():
    g = torch.Generator().manual_seed(1)
    t = torch.nn.functional.normalize(torch.randn(12, 8, generator=g), dim=-1)
    i = torch.nn.functional.normalize(torch.randn(12, 8, generator=g), dim=-1)
    ls = torch.tensor(2.3)
    full = O.clip_loss((t @ i.t()) * ls.exp())
    parts = sum(O.global_clip_loss_rank(t, i, ls, r, 4) for r in range(3)) / 3
    assert abs(full.item() - parts.item()) < 1e-6


@pytest.mark.skipif(not R.reference_available(), reason="reference checkout not present")
def test_oracle_matches_live_reference_and_appzoo_contract(tmp_path):
    cfg = O.CONFIGS["tiny"]
    sd = O.make_state_dict(cfg, 7)
    R.write_checkpoint_dir(str(tmp_path), cfg, sd)
    app = R.reference_clip_app(str(tmp_path))
    px, ids = O.make_inputs(cfg, 8, 16, 11)
    with torch.no_grad():
        ref = app({"pixel_values": px, "input_ids": ids})
        ref_loss = app.compute_loss(ref, [])["loss"]
        out = O.clip_forward(sd, cfg, px, ids)
        loss = O.clip_loss(out["logits_per_text"])
    assert set(ref) == set(out)
    for k in out:
        assert float((ref[k] - out[k]).abs().max()) < 1e-4, k
    assert abs(ref_loss.item() - loss.item()) < 1e-5
    # shapes of the oracle's parameter table == the reference module's
    ref_sd = app.chinese_clip.state_dict()
    shapes = O.param_shapes(cfg)
    assert set(shapes) == {k for k in ref_sd if "position_ids" not in k}
    for k, s in shapes.items():
        assert tuple(ref_sd[k].shape) == s
