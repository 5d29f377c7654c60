"""What packed text batches rest on, shown on the oracle -- i.e. on the reference's own arithmetic: a padded
position has no path to the text feature or to any gradient, with or without train-mode dropout.  Neither its keep decisions nor its
token id change the embeddings, the loss or the gradients (CPU, float64 where exactness matters)."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as O


def _batch(cfg, B, Lq, seed):
    px, ids = O.make_inputs(cfg, B, Lq, seed)
    g = torch.Generator().manual_seed(seed + 1)
    lens = torch.randint(2, Lq + 1, (B,), generator=g)
    lens[0] = Lq
    ids = ids.clamp(min=1) * (torch.arange(Lq)[None, :] < lens[:, None])
    return px, ids, lens


def _masks(cfg, B, Lq, p, seed):
    g = torch.Generator().manual_seed(seed)
    H, nh, nl = cfg["text_hidden_size"], cfg["text_num_attention_heads"], cfg["text_num_hidden_layers"]
    m = {"emb": torch.rand(B, Lq, H, generator=g) >= p}
    for i in range(nl):
        m["%d.attn" % i] = torch.rand(B, nh, Lq, Lq, generator=g) >= p
        m["%d.self_out" % i] = torch.rand(B, Lq, H, generator=g) >= p
        m["%d.out" % i] = torch.rand(B, Lq, H, generator=g) >= p
    return m


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_padded_positions_do_not_reach_the_text_feature(p):
    cfg = O.CONFIGS["tiny"]
    sd = O.make_state_dict(cfg, 3)
    B, Lq = 5, 12
    _, ids, lens = _batch(cfg, B, Lq, 11)
    pad = torch.arange(Lq)[None, :] >= lens[:, None]          # [B, Lq]
    masks = _masks(cfg, B, Lq, p, 5) if p > 0 else None
    drop = {"p_hidden": p, "p_attn": p, "masks": masks} if p > 0 else None
    with torch.no_grad():
        base = O.encode_text(sd, cfg, ids, dropout=drop)
    # (1) other keep decisions at every padded position: hidden sites at padded rows, attention at padded queries AND padded keys
    if p > 0:
        other = {k: v.clone() for k, v in masks.items()}
        flip = _masks(cfg, B, Lq, 0.5, 99)
        for k in other:
            if k.endswith(".attn"):
                sel = pad[:, None, :, None] | pad[:, None, None, :]
                other[k] = torch.where(sel.expand_as(other[k]), flip[k], other[k])
            else:
                other[k] = torch.where(pad[:, :, None].expand_as(other[k]), flip[k], other[k])
        with torch.no_grad():
            again = O.encode_text(sd, cfg, ids, dropout={"p_hidden": p, "p_attn": p, "masks": other})
        assert float((again - base).abs().max()) < 1e-6
    # (2) the rows of padded positions themselves carry nothing: give the padded slots arbitrary embeddings by moving the word
    #     table's pad row -- input_ids stay 0 there, so the mask (ids != 0) is unchanged
    sd2 = dict(sd)
    w = sd["bert.embeddings.word_embeddings.weight"].clone()
    w[0] = torch.randn_like(w[0]) * 3
    sd2["bert.embeddings.word_embeddings.weight"] = w
    with torch.no_grad():
        moved = O.encode_text(sd2, cfg, ids, dropout=drop)
    assert float((moved - base).abs().max()) < 1e-5


def test_padded_positions_get_no_gradient_and_give_none():
    cfg = O.CONFIGS["tiny"]
    sd = O.make_state_dict(cfg, 4)
    B, Lq = 4, 10
    px, ids, lens = _batch(cfg, B, Lq, 21)
    _, loss, grads = O.forward_loss_backward(sd, cfg, px, ids)
    # position rows beyond the longest sentence are never looked up by an unmasked token ... and the looked-up rows of padded
    # slots get exactly nothing either: the position-embedding gradient of row t only sums over sentences longer than t
    gp = grads["bert.embeddings.position_embeddings.weight"]
    longest = int(lens.max())
    assert float(gp[longest:].abs().max()) == 0.0 if longest < gp.shape[0] else True
    # per-position check through an input perturbation: changing the pad row of the word table leaves loss and gradients alone
    sd2 = dict(sd)
    w = sd["bert.embeddings.word_embeddings.weight"].clone()
    w[0] = torch.randn_like(w[0]) * 3
    sd2["bert.embeddings.word_embeddings.weight"] = w
    _, loss2, grads2 = O.forward_loss_backward(sd2, cfg, px, ids)
    assert abs(float(loss2) - float(loss)) < 1e-5
    for n in ("text_projection", "bert.encoder.layer.0.attention.self.query.weight", "bert.embeddings.position_embeddings.weight"):
        assert float((grads2[n] - grads[n]).abs().max()) < 1e-5 * max(1.0, float(grads[n].abs().max())), n
    # the pad row of the word table: whatever the oracle's plain indexing accumulates there comes from padded slots only, i.e. from
    # rows without a path to the loss -- it is zero (torch's padding_idx would also force it to zero)
    assert float(grads["bert.embeddings.word_embeddings.weight"][0].abs().max()) < 1e-7
