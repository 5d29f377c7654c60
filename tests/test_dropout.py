"""BERT train-mode dropout on the HIP path (modeling_bert.py:128,238,266,344).

torch's own mask stream cannot be reproduced (and is not part of any contract), so parity is checked by REPLAY:
the library's counter-based masks (Philox-4x32-10, include/ezclip.h: ezclip_op_dropout_mask) are exported and fed to
the oracle, whose mask-replay formulation is pinned to the real reference in train mode by
tests/test_oracle.py::test_oracle_train_mode_dropout_matches_reference_golden.  The generator itself is pinned to the
published Philox known-answer vectors.
"""
import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from oracle import clip_oracle as O

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
U32 = 0xFFFFFFFF


def philox4x32_10(c, k):
    """numpy restatement (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11);
    c: four uint64 arrays holding 32-bit values, k: two python ints."""
    c0, c1, c2, c3 = [np.asarray(x, dtype=np.uint64) for x in c]
    k0, k1 = k
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(U32)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(U32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & U32, (k1 + W1) & U32
    return c0, c1, c2, c3


def site_words(seed, site, rows, cols):
    """the 32-bit word deciding every element of a [rows, cols] dropout site (csrc/dropout.h)"""
    r, c = np.meshgrid(np.arange(rows, dtype=np.uint64), np.arange(cols, dtype=np.uint64), indexing="ij")
    w = philox4x32_10((c >> np.uint64(2), r, np.full_like(r, site), np.zeros_like(r)), (seed & U32, (seed >> 32) & U32))
    sel = (c & np.uint64(3)).astype(np.int64)
    return np.choose(sel, w)


def test_philox_restatement_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((U32, U32, U32, U32), (U32, U32), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for c, k, want in kat:
        got = philox4x32_10([np.array([x]) for x in c], k)
        assert tuple(int(g[0]) for g in got) == want


# ------------------------------------------------------------------------------------------------ GPU

def library_text_masks(p_hidden, p_attn, seed, B, Lq, cfg, device="cuda"):
    H, heads, nl = cfg["text_hidden_size"], cfg["text_num_attention_heads"], cfg["text_num_hidden_layers"]
    masks = {}
    if p_hidden > 0:
        masks["emb"] = L.op_dropout_mask(p_hidden, seed, 0, B * Lq, H, device).cpu().bool().reshape(B, Lq, H)
    for i in range(nl):
        if p_attn > 0:
            masks[f"{i}.attn"] = L.op_dropout_mask(p_attn, seed, 1 + 3 * i, B * heads * Lq, Lq, device).cpu().bool() \
                .reshape(B, heads, Lq, Lq)
        if p_hidden > 0:
            masks[f"{i}.self_out"] = L.op_dropout_mask(p_hidden, seed, 2 + 3 * i, B * Lq, H, device).cpu().bool().reshape(B, Lq, H)
            masks[f"{i}.out"] = L.op_dropout_mask(p_hidden, seed, 3 + 3 * i, B * Lq, H, device).cpu().bool().reshape(B, Lq, H)
    return {"p_hidden": p_hidden, "p_attn": p_attn, "masks": masks}


@pytest.mark.gpu
def test_mask_generator_is_philox_and_threshold_is_exact():
    seed, site, p = 0x1234_5678_9ABC_DEF1, 7, 0.1
    keep, words = L.op_dropout_mask(p, seed, site, 37, 70, "cuda", want_words=True)
    want = site_words(seed, site, 37, 70)
    assert np.array_equal(words.cpu().numpy().astype(np.uint64), want)
    thr = int(p * 2 ** 32 + 0.5)
    assert np.array_equal(keep.cpu().numpy().astype(bool), want >= thr)
    # statistics and independence between sites / seeds on a big site
    k0 = L.op_dropout_mask(p, seed, 0, 2048, 768, "cuda").float()
    k1 = L.op_dropout_mask(p, seed, 1, 2048, 768, "cuda").float()
    k2 = L.op_dropout_mask(p, seed + 1, 0, 2048, 768, "cuda").float()
    n = k0.numel()
    for k in (k0, k1, k2):
        assert abs(float(k.mean()) - 0.9) < 4 * (0.09 / n) ** 0.5 + 1e-4
    for a, b in ((k0, k1), (k0, k2)):
        corr = float(((a - 0.9) * (b - 0.9)).mean()) / 0.09
        assert abs(corr) < 5 / n ** 0.5
    # neighbouring elements of a row are independent too
    corr = float(((k0[:, 1:] - 0.9) * (k0[:, :-1] - 0.9)).mean()) / 0.09
    assert abs(corr) < 5 / n ** 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_rows_op(dtype):
    g = torch.Generator().manual_seed(3)
    rows, d, p, seed, site = 100, 192, 0.1, 99, 5
    x = torch.randn(rows, d, generator=g).to(dtype)
    r = torch.randn(rows, d, generator=g).to(dtype)
    keep = L.op_dropout_mask(p, seed, site, rows, d, "cuda").cpu().bool()
    y = L.op_dropout(x.cuda(), p, seed, site).cpu()
    want = (x.float() * keep / (1 - p))
    # (x * (1/(1-p)) vs x / (1-p): one ulp)
    assert float((y.float() - want.to(dtype).float()).abs().max()) <= (1e-6 if dtype == torch.float32 else 4e-2)
    assert torch.equal(y == 0, ~keep | (x == 0))
    y2 = L.op_dropout(x.cuda(), p, seed, site, residual=r.cuda()).cpu()
    want2 = want.to(dtype).float() + r.float()
    assert float((y2.float() - want2).abs().max()) <= (1e-6 if dtype == torch.float32 else 4e-2)


def _attn_reference(qkv, B, Lq, heads, key_bias, keep, p):
    D = heads * 64
    q, k, v = [t.reshape(B, Lq, heads, 64).transpose(1, 2) for t in qkv.split(D, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * 0.125
    if key_bias is not None:
        s = s + key_bias.reshape(B, 1, 1, Lq)
    pr = torch.softmax(s, dim=-1)
    if keep is not None:
        pr = pr * keep.to(pr.dtype) / (1 - p)
    return (pr @ v).transpose(1, 2).reshape(B * Lq, D)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Lq", [24, 33, 64, 100])
def test_attention_with_dropout_forward_and_backward(dtype, Lq):
    """softmax -> dropout -> P V (modeling_bert.py:234-244) and its gradients, masks replayed in torch (fp64)."""
    B, heads, p, seed, site = 3, 2, 0.2, 4242, 4
    g = torch.Generator().manual_seed(Lq)
    D = heads * 64
    qkv = (torch.randn(B * Lq, 3 * D, generator=g) * 1.5).to(dtype)
    dctx = torch.randn(B * Lq, D, generator=g).to(dtype)
    kb = torch.zeros(B, Lq)
    kb[1, Lq - 5:] = -10000.0
    keep = L.op_dropout_mask(p, seed, site, B * heads * Lq, Lq, "cuda").cpu().bool().reshape(B, heads, Lq, Lq)
    ctx, lse = L.op_attention(qkv.cuda(), B, Lq, heads, key_bias=kb.reshape(-1).cuda(), want_lse=True, dropout=(p, seed, site))
    dqkv = L.op_attention_bwd(qkv.cuda(), ctx, dctx.cuda(), lse, B, Lq, heads, key_bias=kb.reshape(-1).cuda(), dropout=(p, seed, site))
    dq, dk, dv = dqkv.split(D, dim=-1)
    ref_in = qkv.double().requires_grad_(True)
    ref = _attn_reference(ref_in, B, Lq, heads, kb.double(), keep, p)
    ref.backward(dctx.double())
    gq, gk, gv = ref_in.grad.split(D, dim=-1)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert float((ctx.cpu().double() - ref.detach()).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    for got, want in ((dq, gq), (dk, gk), (dv, gv)):
        err = float((got.cpu().double() - want).norm()) / float(want.norm())
        assert err < (1e-5 if dtype == torch.float32 else 2e-2), err
    # and with dropout off the same call is the plain attention
    ctx0 = L.op_attention(qkv.cuda(), B, Lq, heads, key_bias=kb.reshape(-1).cuda())
    ref0 = _attn_reference(qkv.double(), B, Lq, heads, kb.double(), None, 0.0)
    assert float((ctx0.cpu().double() - ref0).abs().max()) < tol * max(1.0, float(ref0.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("cfg_name,B,Lq", [("tiny", 6, 24), ("small", 5, 40)])
def test_text_tower_train_mode_dropout_matches_oracle_replay(tmp_path, dtype, cfg_name, B, Lq):
    """CLIPApp in train mode with the reference's dropout probabilities: embeddings, loss and every parameter
    gradient equal the oracle's when the oracle replays the library's masks; eval mode ignores dropout."""
    from test_model_gpu import make_app
    cfg = dict(O.CONFIGS[cfg_name], text_hidden_dropout_prob=0.1, text_attention_probs_dropout_prob=0.15)
    app, sd = make_app(tmp_path, cfg, 1234, dtype)
    px, ids = O.make_inputs(cfg, B, Lq, 3)
    seed = 0x5EED_0000_0001
    app.train()
    app.dropout_seed = seed
    out = app({"pixel_values": px, "input_ids": ids})
    loss = app.compute_loss(out, [])["loss"]
    loss.backward()
    drop = library_text_masks(0.1, 0.15, seed, B, Lq, cfg)
    ref_out, ref_loss, ref_g = O.forward_loss_backward(sd, cfg, px, ids, dropout=drop)
    f32 = dtype == "fp32"
    assert float((out["text_embeds"].detach().cpu() - ref_out["text_embeds"]).abs().max()) < (2e-5 if f32 else 1e-2)
    assert float((out["image_embeds"].detach().cpu() - ref_out["image_embeds"]).abs().max()) < (2e-5 if f32 else 1e-2)
    assert abs(loss.item() - ref_loss.item()) < (2e-5 if f32 else 1.5e-2)
    # gradients: fp32 at 1e-4 (2e-4 for the near-cancelling ones); bf16 relative to the same-shape scale of the tower
    scale = {}
    for n, gr in ref_g.items():
        if gr is not None:
            key = (n.split(".")[0], tuple(gr.shape))
            scale[key] = max(scale.get(key, 0.0), float(gr.norm()))
    bad = []
    for n, gr in ref_g.items():
        p = app._params[n]
        if gr is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        err = float((p.grad.detach().cpu().double() - gr.double()).norm())
        bound = (2e-4 * float(gr.norm()) + 1e-7) if f32 else (6e-2 * float(gr.norm()) + 2e-2 * scale[(n.split(".")[0], tuple(gr.shape))] + 1e-6)
        if err > bound:
            bad.append((n, err, float(gr.norm())))
    assert not bad, bad[:10]
    # same seed -> same result; another seed -> another mask; eval mode -> no dropout at all
    with torch.no_grad():
        again = app({"input_ids": ids}, feat=True)["text_embeds"]
        assert torch.equal(again, out["text_embeds"].detach())
        app.dropout_seed = seed + 1
        other = app({"input_ids": ids}, feat=True)["text_embeds"]
        assert float((other - again).abs().max()) > 1e-3
        app.eval()
        ev = app({"input_ids": ids}, feat=True)["text_embeds"].cpu()
        assert float((ev - O.encode_text(sd, cfg, ids)).abs().max()) < (2e-5 if f32 else 1e-2)


@pytest.mark.gpu
def test_dropout_seed_follows_torch_manual_seed(tmp_path):
    """Without a pinned seed every train-mode pass draws a fresh seed from torch's CPU generator."""
    from test_model_gpu import make_app
    cfg = dict(O.CONFIGS["tiny"], text_hidden_dropout_prob=0.1, text_attention_probs_dropout_prob=0.1)
    app, _ = make_app(tmp_path, cfg, 1, "fp32")
    _, ids = O.make_inputs(cfg, 4, 16, 0)
    app.train()
    with torch.no_grad():
        torch.manual_seed(7)
        a1 = app({"input_ids": ids}, feat=True)["text_embeds"].clone()
        a2 = app({"input_ids": ids}, feat=True)["text_embeds"].clone()
        torch.manual_seed(7)
        b1 = app({"input_ids": ids}, feat=True)["text_embeds"].clone()
    assert torch.equal(a1, b1) and not torch.equal(a1, a2)
