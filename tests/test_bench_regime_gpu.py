"""Parity in the regime bench.py runs in (GPU).

The headline workloads launch the persistent 8-phase GEMM with thousands of 256 x 256 tiles on 256 CUs: the
multi-tile loop (next tile's LDS-DMA issued under the epilogue, the relaxed first-K-tile wait, the XCD-aware raster
over rounds -- gemm8p.hip) is the only path they execute, and it only exists when a launch has more tiles than CUs.
These tests assert numerics THERE:

* op level -- every epilogue the towers use (bias / activation / residual, row-stat partials, second pre-activation
  output, act'(U) with fused column sums, folded LayerNorm) at M ~ 70 000 ragged rows (274 row tiles x 3..12 column
  tiles = 822..3288 tiles): bit-equal to the 128 x 128 kernel where that kernel has the epilogue, bit-equal to the same
  kernel launched one workgroup per tile (no persistent loop) everywhere, and within bf16 rounding of an fp64
  evaluation on sampled rows;  the weight-gradient kernel at full contraction length;
* model level -- ViT-B/16 + BERT-base at 64 and 256 pairs: the bf16 pipeline against the fp32 pipeline (which the
  small-batch tests pin to the real reference at 1e-5) and the CPU oracle on a 16-pair sample; loss and the gradient of
  every parameter.

Reference semantics: modeling_chineseclip.py:184-253 (ViT blocks), modeling_bert.py:349-526 (BERT layers).
"""
import math

import pytest
import torch

from easynlp_amd import lib as L
from oracle import clip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
M_BIG = 70011            # 274 row tiles, the last one ragged (123 rows)


def _inputs(M, N, K, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    a = (torch.randn(M, K, generator=g, device=DEV) * 0.5).bfloat16()
    b = (torch.randn(N, K, generator=g, device=DEV) * 0.2).bfloat16()
    bias = torch.randn(N, generator=g, device=DEV)
    res = torch.randn(M, N, generator=g, device=DEV).bfloat16()
    u = torch.randn(M, N, generator=g, device=DEV).bfloat16()
    return a, b, bias, res, u


def _rows(M, n=48, seed=0):
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, M, (n,), generator=g)
    idx[0], idx[1], idx[2] = 0, M - 1, (M // 256) * 256          # first row, last (ragged tile), first row of the ragged tile
    return idx.to(DEV)


def _act64(z, act):
    if act == L.ACT_QUICKGELU:
        return O.quick_gelu(z)
    if act == L.ACT_GELU_ERF:
        return O.gelu_erf(z)
    return z


def _act_grad64(u, act):
    u = u.clone().requires_grad_(True)
    _act64(u, act).sum().backward()
    return u.grad


def _tiles(M, N):
    return ((M + 255) // 256) * (N // 256)


@pytest.mark.parametrize("N,K", [(768, 768), (2304, 768), (768, 3072)])
@pytest.mark.parametrize("epi", ["bias", "bias_res_ps", "bias_qgelu", "bias_gelu_c2", "u_qgelu_colsum", "u_gelu_colsum"])
def test_gemm_nt_persistent_regime(N, K, epi):
    M = M_BIG
    assert _tiles(M, N) > 3 * 256          # at least three rounds of tiles on a 256-CU part
    a, b, bias, res, u = _inputs(M, N, K, seed=N + K)
    act = L.ACT_QUICKGELU if "qgelu" in epi else (L.ACT_GELU_ERF if "gelu" in epi else L.ACT_NONE)
    kw = dict(act=act)
    if epi.startswith("u_"):
        kw["u"] = u                       # dX = (dY . W) * act'(U); colsum = the bias gradient
    else:
        kw["bias"] = bias
    if "res" in epi:
        kw["residual"] = res
    outs = {}
    for fk in (0, 2, 24):
        extra = {}
        if "ps" in epi and fk != 0:
            extra["rowstat_part"] = torch.full((M, N // 64, 2), float("nan"), device=DEV)
        if "c2" in epi:
            extra["c2"] = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        if "colsum" in epi:
            extra["colsum"] = torch.zeros(N, device=DEV)
        c = L.op_gemm_nt_ex(a, b, force_kernel=fk, **kw, **extra)
        torch.cuda.synchronize()
        outs[fk] = (c, extra)
    c0, c2, c24 = outs[0][0], outs[2][0], outs[24][0]
    assert torch.equal(c2, c24), "persistent multi-tile loop differs from one-workgroup-per-tile launches"
    assert torch.equal(c2, c0), "8-phase kernel differs from the 128x128 kernel"
    if "c2" in epi:
        assert torch.equal(outs[2][1]["c2"], outs[0][1]["c2"]) and torch.equal(outs[2][1]["c2"], outs[24][1]["c2"])
    # fp64 on sampled rows
    idx = _rows(M)
    z = a[idx].double() @ b.double().t()
    if epi.startswith("u_"):
        ref = z * _act_grad64(u[idx].double(), act)
    else:
        ref = _act64(z + bias.double(), act)
        if "c2" in epi:
            pre = z + bias.double()
            assert float((outs[2][1]["c2"][idx].double() - pre).abs().max()) < 0.06 * max(1.0, math.sqrt(K) / 8)
    if "res" in epi:
        ref = ref + res[idx].double()
    tol = 0.06 * max(1.0, math.sqrt(K) / 8)
    assert float((c2[idx].double() - ref).abs().max()) < tol
    if "colsum" in epi:
        # fused (atomics per wave tile) vs the separate column-sum pass of the other kernel vs fp64 of the rounded output
        exact = c2.double().sum(0)
        for fk in (0, 2, 24):
            got = outs[fk][1]["colsum"].double()
            assert float((got - exact).abs().max()) < 2e-3 * float(exact.abs().max() + c2.double().abs().sum(0).max() * 1e-3)
    if "ps" in epi:
        part = outs[2][1]["rowstat_part"]
        assert torch.equal(part, outs[24][1]["rowstat_part"])
        cs = c2.float().reshape(M, N // 64, 64)
        s1, s2 = cs.double().sum(-1), (cs.double() ** 2).sum(-1)
        assert float((part[..., 0].double() - s1).abs().max()) < 1e-3 * max(1.0, float(s1.abs().max()))
        assert float((part[..., 1].double() - s2).abs().max()) < 1e-4 * float(s2.abs().max())


@pytest.mark.parametrize("N,act", [(2304, L.ACT_NONE), (3072, L.ACT_QUICKGELU)])
def test_gemm_nt_folded_layernorm_persistent_regime(N, act):
    """in_proj / c_fc of the bf16 inference path: LayerNorm folded into the product (GemmArgs::ln_stats)."""
    M, K, eps = M_BIG, 768, 1e-5
    g = torch.Generator(device=DEV).manual_seed(N)
    x = (torch.randn(M, K, generator=g, device=DEV) * 1.5 + 0.3).bfloat16()
    w = torch.randn(N, K, generator=g, device=DEV) * 0.05
    gain = 1 + 0.2 * torch.randn(K, generator=g, device=DEV)
    shift = 0.1 * torch.randn(K, generator=g, device=DEV)
    bias = torch.randn(N, generator=g, device=DEV)
    wf = (w * gain).bfloat16()                                      # W o g, rounded as the library packs it
    c1 = wf.float().sum(1).contiguous()
    c2v = (w.double() @ shift.double() + bias.double()).float().contiguous()
    stats = L.op_layernorm_stats(x, eps)
    outs = {}
    for fk in (2, 24):
        outs[fk] = L.op_gemm_nt_ex(x, wf, ln_stats=stats, ln_c1=c1, ln_c2=c2v, act=act, force_kernel=fk)
        torch.cuda.synchronize()
    assert torch.equal(outs[2], outs[24]), "persistent multi-tile loop differs from one-workgroup-per-tile launches"
    idx = _rows(M)
    xr = x[idx].double()
    # the row statistics themselves
    mu, var = xr.mean(-1), xr.var(-1, unbiased=False)
    rstd = torch.rsqrt(var + eps)
    assert float((stats[idx, 0].double() - rstd).abs().max() / rstd.abs().max()) < 1e-5
    assert float((stats[idx, 1].double() + mu * rstd).abs().max()) < 1e-4
    # LN(x) W^T + b evaluated directly in fp64 (gain folded as the library rounds it)
    ln = (xr - mu[:, None]) * rstd[:, None]
    ref = _act64(ln @ wf.double().t() + c2v.double(), act)
    assert float((outs[2][idx].double() - ref).abs().max()) < 0.06


@pytest.mark.parametrize("N,K", [(3072, 768), (768, 3072), (768, 768)])
def test_gemm_tn_full_contraction(N, K):
    """Weight gradients dW = dY^T X over ~70 000 rows: split partials + fixed-order reduction (bit-reproducible), ragged M."""
    M = M_BIG
    g = torch.Generator(device=DEV).manual_seed(N * 3 + K)
    dy = torch.randn(M, N, generator=g, device=DEV).bfloat16()
    x = torch.randn(M, K, generator=g, device=DEV).bfloat16()
    lib = L.load()
    outs = []
    for rep in range(2):
        c = torch.full((N, K), 1.0, device=DEV)
        L.check(lib.ezclip_op_gemm_tn(dy.data_ptr(), N, x.data_ptr(), K, c.data_ptr(), K, M, N, K, 1, L.DTYPE_BF16, L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(c)
    assert torch.equal(outs[0], outs[1]), "weight gradient is not bit-reproducible"
    # the 128x128 atomics kernel (another accumulation order) and fp64 on sampled output rows
    L.check(lib.ezclip_debug_set(0, 0))
    try:
        c0 = torch.full((N, K), 1.0, device=DEV)
        L.check(lib.ezclip_op_gemm_tn(dy.data_ptr(), N, x.data_ptr(), K, c0.data_ptr(), K, M, N, K, 1, L.DTYPE_BF16, L.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        L.check(lib.ezclip_debug_set(0, -1))
    rows = torch.randint(0, N, (24,), generator=torch.Generator().manual_seed(1)).to(DEV)
    ref = 1.0 + dy[:, rows].double().t() @ x.double()
    scale = math.sqrt(M)
    assert float((outs[0][rows].double() - ref).abs().max()) < 2e-5 * scale
    assert float((c0[rows].double() - ref).abs().max()) < 2e-4 * scale
    assert float((outs[0] - c0).abs().max()) < 2e-4 * scale


# ------------------------------------------------------------------------------------------------------ model level

VITB16 = dict(
    model_type="chinese_clip", embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
    vision_patch_size=16, vocab_size=21128, text_attention_probs_dropout_prob=0.0, text_hidden_act="gelu",
    text_hidden_dropout_prob=0.0, text_hidden_size=768, text_initializer_range=0.02, text_intermediate_size=3072,
    text_max_position_embeddings=512, text_num_attention_heads=12, text_num_hidden_layers=12, text_type_vocab_size=2)


def _synth(batch, seq, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    px = torch.randn((batch, 3, 224, 224), generator=g, device=DEV)
    ids = torch.randint(1, VITB16["vocab_size"], (batch, seq), generator=g, device=DEV)
    lens = torch.randint(8, seq + 1, (batch,), generator=g, device=DEV)
    return px, ids * (torch.arange(seq, device=DEV)[None, :] < lens[:, None])


def _check_ranks(rank, txt, img, eps=2e-6):
    """The evaluator's rank of the matching image in each text row's descending score order (evaluator.py:47-67) against the
    float64 scores of the same embeddings.  On random-init towers the embeddings nearly coincide and neighbouring scores differ by
    less than a float32 ulp of the device's f32 product, so an exact comparison with a host sort is a coin toss per near-tie (it
    held in round 3 and flipped with the GELU polynomial of round 4): the rank must lie between the number of scores that beat the
    diagonal by more than eps and the number within eps of beating it -- and equal the host sort wherever no score is that close."""
    sim = txt.double().cpu() @ img.double().cpu().t()
    d = sim.diagonal()[:, None]
    lo, hi = (sim > d + eps).sum(1), (sim > d - eps).sum(1) - 1          # (the diagonal itself is within eps of itself)
    assert bool(((rank >= lo) & (rank <= hi)).all()), (rank[:8], lo[:8], hi[:8])
    clear = lo == hi
    assert int(clear.sum()) > 0.5 * len(rank)                             # the bracket is tight for most rows
    order = torch.sort(sim.float(), dim=1, descending=True, stable=True).indices
    want = (order == torch.arange(len(rank))[:, None]).long().argmax(dim=1)
    assert torch.equal(rank[clear], want[clear])


def _train_step(app, px, ids):
    for p in app.parameters():
        p.grad = None
    loss = app.contrastive_step(px, ids, process_group=False, backward=True, zero_grad=True)
    torch.cuda.synchronize()
    return float(loss.item()), {n: p.grad.detach().clone() for n, p in app._params.items() if p.grad is not None}


@pytest.mark.parametrize("B", [64, 256])
def test_vitb16_bertbase_bf16_against_fp32_and_oracle(B):
    """B = 64: 49 x {3, 9, 12} = 147..588 tiles per ViT GEMM (one to three rounds); B = 256: 197 x ... = 591..2364 tiles
    (up to ten rounds) -- the persistent regime of the headline run."""
    from easynlp_amd.appzoo.clip import CLIPApp
    S = 64
    px, ids = _synth(B, S, seed=1000)
    res = {}
    for dtype in ("fp32", "bf16"):
        app = CLIPApp.from_config(VITB16, seed=1234, device=DEV, compute_dtype=dtype)
        app.eval()
        with torch.no_grad():
            out = app({"pixel_values": px, "input_ids": ids})
            loss_ag = float(app.compute_loss(out, [])["loss"].item())
            loss_fused = float(app.contrastive_step(px, ids, process_group=False).item())
        loss_bwd, grads = _train_step(app, px, ids)
        res[dtype] = dict(img=out["image_embeds"].float().cpu(), txt=out["text_embeds"].float().cpu(), loss_ag=loss_ag,
                          loss_fused=loss_fused, loss_bwd=loss_bwd, grads={n: g.cpu() for n, g in grads.items()})
        if dtype == "fp32":
            sd = {k: v.detach().cpu() for k, v in app._params.items()}
        del app, out, grads
        torch.cuda.empty_cache()
    f, h = res["fp32"], res["bf16"]
    # (1) the fp32 pipeline against the CPU oracle on a 16-pair sample (the towers are per-sample: rows are independent)
    sample = torch.arange(0, B, B // 16)[:16]
    cfg = O.CONFIGS["vitb16_bertbase"]
    with torch.no_grad():
        ref = O.clip_forward(sd, cfg, px[sample].cpu(), ids[sample].cpu())
    assert float((f["img"][sample] - ref["image_embeds"]).abs().max()) < 2e-5
    assert float((f["txt"][sample] - ref["text_embeds"]).abs().max()) < 2e-5
    # ... and its loss against the oracle's formula on the full embedding sets (fp64)
    scale = math.exp(float(sd["logit_scale"]))
    logits = scale * f["txt"].double() @ f["img"].double().t()
    ref_loss = float(O.clip_loss(logits))
    assert abs(f["loss_ag"] - ref_loss) < 1e-4 and abs(f["loss_fused"] - ref_loss) < 1e-4 and abs(f["loss_bwd"] - ref_loss) < 1e-4
    # (2) bf16 against fp32: embeddings, loss
    assert float((h["img"] - f["img"]).abs().max()) < 1e-2 and float((h["txt"] - f["txt"]).abs().max()) < 1e-2
    assert float(torch.nn.functional.cosine_similarity(h["img"], f["img"]).min()) > 0.9995
    assert float(torch.nn.functional.cosine_similarity(h["txt"], f["txt"]).min()) > 0.9995
    for k in ("loss_ag", "loss_fused", "loss_bwd"):
        assert abs(h[k] - ref_loss) < 5e-3, (k, h[k], ref_loss)
    # (3) every parameter gradient: same set, and bf16 within the bf16 bound of the small-batch golden tests
    #     (6 % of its own norm + 1 % of the largest gradient norm among same-shape parameters of the tower)
    assert set(h["grads"]) == set(f["grads"])
    assert not any(n.startswith("bert.pooler") for n in f["grads"])
    by_shape = {}
    for n, g in f["grads"].items():
        key = (n.split(".")[0], tuple(g.shape))
        by_shape[key] = max(by_shape.get(key, 0.0), float(g.double().norm()))
    worst = []
    for n, g in f["grads"].items():
        gn = float(g.double().norm())
        assert math.isfinite(gn) and gn > 0, n
        err = float((h["grads"][n].double() - g.double()).norm())
        bound = 6e-2 * gn + 1e-2 * by_shape[(n.split(".")[0], tuple(g.shape))]
        worst.append((err / bound, n, err, gn))
    worst.sort(reverse=True)
    print("worst bf16 gradient deviations (fraction of bound, name, |diff|, |ref|):", worst[:6])
    assert worst[0][0] < 1.0, worst[:6]


def test_headline_batch_of_1024_pairs_size_independent_properties():
    """The bench's own configuration (ViT-B/16 + BERT-base, bf16, 1 024 pairs, 64 tokens with ragged lengths; 2 364..9 456 tiles per ViT
    product) through properties that do not need an oracle run of that size:
      * a pair's embeddings do not depend on the batch it is computed in (the towers are per-sample, modeling_chineseclip.py:343-365):
        pairs 512..767 encoded alone give the same BITS (other GEMM tile counts, another packing of the text rows, the same
        arithmetic per row), 16 scattered pairs encoded alone the same rows within the bf16 bound (another kernel selection);
      * the loss of the fused step equals the reference's formula (appzoo/clip/model.py:154-164) evaluated in float64 on the
        embeddings the step produced;
      * the bf16 rows stay within the bf16 bound of the float32 pipeline's rows on the sample;
      * the evaluator's ranks (evaluator.py:47-67: descending sort of each text row's scores) on the 1 024 embeddings equal a sort
        on the host."""
    from easynlp_amd.appzoo.clip import CLIPApp
    from easynlp_amd.appzoo.clip.evaluator import recall_ranks
    B, S = 1024, 64
    px, ids = _synth(B, S, seed=1000)
    app = CLIPApp.from_config(VITB16, seed=1234, device=DEV, compute_dtype="bf16")
    app.eval()
    sample = torch.arange(5, B, B // 16)[:16].to(DEV)
    with torch.no_grad():
        out = app({"pixel_values": px, "input_ids": ids})
        img, txt = out["image_embeds"].float(), out["text_embeds"].float()
        loss_fused = float(app.contrastive_step(px, ids, process_group=False).item())
        loss_ag = float(app.compute_loss(out, [])["loss"].item())
        sub = app({"pixel_values": px[sample].contiguous(), "input_ids": ids[sample].contiguous()})
        quarter = app({"pixel_values": px[512:768].contiguous(), "input_ids": ids[512:768].contiguous()})
        again = app({"pixel_values": px, "input_ids": ids})
    assert torch.equal(again["image_embeds"], out["image_embeds"]) and torch.equal(again["text_embeds"], out["text_embeds"])
    assert float((img.norm(dim=-1) - 1).abs().max()) < 1e-3 and float((txt.norm(dim=-1) - 1).abs().max()) < 1e-3
    d_img = float((sub["image_embeds"].float() - img[sample]).abs().max())
    d_txt = float((sub["text_embeds"].float() - txt[sample]).abs().max())
    q_img = float((quarter["image_embeds"].float() - img[512:768]).abs().max())
    q_txt = float((quarter["text_embeds"].float() - txt[512:768]).abs().max())
    print("1024-pair batch vs pairs 512..767 alone: max |d image_embeds| %.3e  |d text_embeds| %.3e;  vs 16 pairs alone: %.3e  %.3e"
          % (q_img, q_txt, d_img, d_txt))
    # 256 pairs take the same kernels as 1 024 (persistent GEMM, LayerNorm folded into it): the rows are the same BITS.  16 pairs fall
    # under the row count at which LayerNorm is folded into the following product (model.hip can_fold_ln) and see the materialised
    # LayerNorm's extra bf16 rounding: same rows within the bf16 bound.
    assert q_img == 0.0 and q_txt == 0.0
    assert d_img < 2e-3 and d_txt < 2e-3
    scale = math.exp(float(app._params["logit_scale"].detach()))
    logits = scale * txt.double().cpu() @ img.double().cpu().t()
    ref_loss = float(O.clip_loss(logits))
    assert abs(loss_fused - ref_loss) < 5e-3 and abs(loss_ag - ref_loss) < 5e-3, (loss_fused, loss_ag, ref_loss)
    # ranks: the f32 rounding of the f64 scores, stable descending sort (ties between DIFFERENT scores after rounding are possible
    # in principle; none at this size -- checked by the equality itself)
    _check_ranks(recall_ranks(txt, img).cpu().long(), txt, img)
    # (round 4) the bf16 rows of the 1 024-pair batch against the CPU ORACLE itself on the 16 sampled pairs --
    # not only against the float32 HIP pipeline below (which the 64 / 256-pair test ties to the oracle)
    sd = {k: v.detach().cpu() for k, v in app._params.items()}
    with torch.no_grad():
        oref = O.clip_forward(sd, O.CONFIGS["vitb16_bertbase"], px[sample].cpu(), ids[sample].cpu())
    for got, want_e in ((img[sample].cpu(), oref["image_embeds"]), (txt[sample].cpu(), oref["text_embeds"])):
        assert float((got - want_e).abs().max()) < 1e-2
        assert float(torch.nn.functional.cosine_similarity(got, want_e).min()) > 0.9995
    del app, out, again, sub, quarter
    torch.cuda.empty_cache()
    f32 = CLIPApp.from_config(VITB16, seed=1234, device=DEV, compute_dtype="fp32")
    f32.eval()
    with torch.no_grad():
        ref = f32({"pixel_values": px[sample].contiguous(), "input_ids": ids[sample].contiguous()})
    for got, want_e in ((img[sample], ref["image_embeds"]), (txt[sample], ref["text_embeds"])):
        assert float((got - want_e).abs().max()) < 1e-2
        assert float(torch.nn.functional.cosine_similarity(got, want_e).min()) > 0.9995


@pytest.mark.parametrize("flavour", ["chinese_clip_vitl14", "huggingface_clip_vitl14_large_text"])
def test_config5_batch_of_512_pairs_size_independent_properties(flavour):
    """BASELINE.json config 5 at the bench's own depth and batch (round 4): ViT-L/14 with all 24 blocks
    (257 tokens, width 1024, patch 14) at 512 pairs in bf16 -- as a chinese_clip model over the BERT-base-arch text tower
    (`bf16_vitl14_b512_*`) and as the huggingface_clip flavour over the LARGE RoBERTa text tower (hidden 1024, 24 layers, 16 heads,
    FFN 4096: configuration_clip.py:90-95; `bf16_hf_vitl14_large_b512_train`).  No oracle run of that size is needed:
      * a second evaluation gives the same bits; pairs 128..383 encoded alone give the same BITS as inside the 512-pair batch (the towers
        are per-sample: modeling_chineseclip.py:343-365 / model.py:128-144) -- other tile counts, another packing of the text rows
        (256 pairs, not fewer: below 256 rows the CLS-only last block materialises its LayerNorm instead of folding it into the
        product -- one more bf16 rounding, 2e-4 on the embeddings; the 1 024-pair test pins that regime with its 16-pair sample);
      * 8 scattered pairs against the CPU ORACLE evaluated on exactly those pairs with the same weights (bf16 bound of the golden tests);
      * the loss of the fused step equals the reference's formula (model.py:154-164) in float64 on the embeddings the step produced;
      * the evaluator's ranks equal a host sort of the same scores."""
    import bench
    from easynlp_amd.appzoo.clip import CLIPApp
    from easynlp_amd.appzoo.clip.evaluator import recall_ranks
    from oracle import hf_clip_oracle as H
    B, S = 512, 64
    px, ids = _synth(B, S, seed=2000)
    hf = flavour.startswith("huggingface")
    if hf:
        cfg = bench.HF_VITL14_ROBERTA_LARGE
        app = CLIPApp.from_hf_config(cfg, seed=4321, device=DEV, compute_dtype="bf16")
        extra = lambda i: {"token_type_ids": torch.zeros_like(i), "attention_mask": i.ne(0).long()}     # noqa: E731
    else:
        cfg = O.CONFIGS["vitl14_robertabase"]
        assert cfg == bench.VITL14_ROBERTA
        app = CLIPApp.from_config(cfg, seed=4321, device=DEV, compute_dtype="bf16")
        extra = lambda i: {}                                                                            # noqa: E731
    app.eval()
    sample = torch.arange(3, B, B // 8)[:8].to(DEV)

    def run(p, i):
        return app(dict({"pixel_values": p.contiguous(), "input_ids": i.contiguous()}, **extra(i.contiguous())))

    with torch.no_grad():
        out = run(px, ids)
        img, txt = out["image_embeds"].float(), out["text_embeds"].float()
        loss_ag = float(app.compute_loss(out, [])["loss"].item())
        e = extra(ids)
        loss_fused = float(app.contrastive_step(px, ids, process_group=False, **e).item())
        again = run(px, ids)
        quarter = run(px[128:384], ids[128:384])
    assert torch.equal(again["image_embeds"], out["image_embeds"]) and torch.equal(again["text_embeds"], out["text_embeds"])
    assert float((img.norm(dim=-1) - 1).abs().max()) < 1e-3 and float((txt.norm(dim=-1) - 1).abs().max()) < 1e-3
    q_img = float((quarter["image_embeds"].float() - img[128:384]).abs().max())
    q_txt = float((quarter["text_embeds"].float() - txt[128:384]).abs().max())
    print("%s: 512-pair batch vs pairs 128..383 alone: max |d image_embeds| %.3e  |d text_embeds| %.3e" % (flavour, q_img, q_txt))
    assert q_img == 0.0 and q_txt == 0.0
    scale = math.exp(float((app._hf_params if hf else app._params)["logit_scale"].detach()))
    logits = scale * txt.double().cpu() @ img.double().cpu().t()
    ref_loss = float(O.clip_loss(logits))
    assert abs(loss_fused - ref_loss) < 5e-3 and abs(loss_ag - ref_loss) < 5e-3, (loss_fused, loss_ag, ref_loss)
    _check_ranks(recall_ranks(txt, img).cpu().long(), txt, img)
    # the oracle on the 8 sampled pairs, same weights (24 + 12 / 24 + 24 blocks in float32 on the host: seconds)
    spx, sids = px[sample].cpu(), ids[sample].cpu()
    with torch.no_grad():
        if hf:
            sd = {n: p.detach().cpu() for n, p in app._hf_params.items()}
            ref = H.hf_clip_forward(sd, cfg, spx, sids, torch.zeros_like(sids), sids.ne(0).long())
        else:
            sd = {k: v.detach().cpu() for k, v in app._params.items()}
            ref = O.clip_forward(sd, cfg, spx, sids)
    for got, want in ((img[sample].cpu(), ref["image_embeds"]), (txt[sample].cpu(), ref["text_embeds"])):
        d = float((got - want).abs().max())
        c = float(torch.nn.functional.cosine_similarity(got, want).min())
        print("  vs oracle on 8 pairs: max |d| %.3e  min cosine %.6f" % (d, c))
        assert d < 1e-2 and c > 0.9995
    del app, out, again, quarter
    torch.cuda.empty_cache()


_DD_CASES = {
    # the headline towers at the headline batch
    "vitb16_b1024": (VITB16, 1024, ["text_projection", "visual.proj", "visual.transformer.resblocks.5.mlp.c_fc.weight", "visual.positional_embedding",
                                   "bert.encoder.layer.6.attention.self.query.weight", "bert.encoder.layer.11.output.dense.weight", "logit_scale"]),
    # BASELINE config 5's towers at full depth (24 ViT-L/14 blocks, 257 tokens; round 4): 256 pairs -- the float32
    # pipeline keeps 16 M D floats per block for the backward pass, 104 GB at this size (207 GB at 512 pairs)
    "vitl14_b256": (dict(O.CONFIGS["vitl14_robertabase"]), 256,
                    ["visual.proj", "visual.transformer.resblocks.12.mlp.c_fc.weight", "visual.transformer.resblocks.23.attn.in_proj_weight",
                     "visual.conv1.weight", "bert.encoder.layer.6.attention.self.query.weight", "text_projection"]),
}


@pytest.mark.parametrize("case", ["vitb16_b1024", "vitl14_b256"])
def test_directional_derivatives_at_1024_pairs_fp32(case):
    """Gradient check at the benchmark's sizes, float32 pipeline (the bf16 pipeline is tied to it at 64 / 256 pairs above): for
    parameters spread over both towers and the heads, the directional derivative <dL/dp, d> of ONE fused training step
    (autograd of core/trainer.py:658-661 through the kernels' backward) against the central difference (L(p + h d) - L(p - h d)) / 2h of
    two forward steps.  A size-independent property: no oracle run of this size is needed, and every kernel of the backward pass sees
    the row counts of the benchmark (201 728 ViT rows at 1 024 pairs of ViT-B/16; 65 792 rows x 24 blocks of ViT-L/14)."""
    from easynlp_amd.appzoo.clip import CLIPApp
    cfg, B, names = _DD_CASES[case]
    S = 64
    px, ids = _synth(B, S, seed=1000)
    app = CLIPApp.from_config(cfg, seed=1234, device=DEV, compute_dtype="fp32")
    app.train()                                   # (dropout probabilities are 0 in these configs)
    loss0, grads = _train_step(app, px, ids)
    g = torch.Generator(device=DEV).manual_seed(77)
    report = []
    for n in names:
        p = app._params[n]
        d = torch.randn(p.shape, generator=g, device=DEV) if p.dim() > 0 else torch.ones((), device=DEV)
        d = d * (float(p.detach().norm()) / max(float(d.norm()), 1e-30)) if p.dim() > 0 else d
        analytic = float((grads[n].double() * d.double()).sum())
        h = 2e-2 if p.dim() > 0 else 5e-2           # relative to |p| (d has p's norm); logit_scale: absolute
        vals = []
        for sgn in (+1.0, -1.0):
            with torch.no_grad():
                p.add_(d, alpha=sgn * h)
                app._engine.mark_weights_dirty()
                vals.append(float(app.contrastive_step(px, ids, process_group=False).item()))
                p.add_(d, alpha=-sgn * h)
        app._engine.mark_weights_dirty()
        fd = (vals[0] - vals[1]) / (2 * h)
        report.append((n, analytic, fd))
    with torch.no_grad():
        back = float(app.contrastive_step(px, ids, process_group=False).item())
    print("loss %.6f (again %.6f); directional derivatives (name, analytic, central difference):" % (loss0, back), report)
    assert abs(back - loss0) < 2e-6 * max(1.0, abs(loss0))      # the parameters are back where they were
    for n, analytic, fd in report:
        assert math.isfinite(analytic) and abs(analytic) > 1e-6, (n, analytic)
        # central difference: O(h^2) truncation + the loss's float32 rounding (~1e-6) / 2h
        assert abs(fd - analytic) < 3e-2 * abs(analytic) + 1e-4, (n, analytic, fd)
