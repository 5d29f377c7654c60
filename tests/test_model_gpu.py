"""Model-level parity (GPU): the HIP dual encoder behind the CLIPApp API against
(a) the committed golden fixtures produced by the real reference and (b) the
CPU oracle run on the same seeded inputs.

Tolerances: f32 path -- embeddings max-abs <= 1e-5 (unit vectors),
logits <= 2e-4 * scale, loss <= 1e-5 rel; bf16 path -- embedding max-abs <= 1e-2
/ cosine >= 0.9995, logits <= 0.15, loss <= 1.5e-2 abs (5e-3 holds on the full-size model).
"""
import os

import numpy as np
import pytest
import torch

from easynlp_amd import lib as L
from easynlp_amd.appzoo.clip import CLIPApp, CLIPEvaluator, CLIPPredictor
from oracle import clip_oracle as O
from oracle import ref_harness as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_gold(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg_name, B, Lq, wseed, iseed = [str(x) for x in z["meta"][:5]]
    return z, O.CONFIGS[cfg_name], int(B), int(Lq), int(wseed), int(iseed)


def make_app(tmp_path, cfg, seed, dtype):
    sd = O.make_state_dict(cfg, seed)
    R.write_checkpoint_dir(str(tmp_path), cfg, sd)
    app = CLIPApp(str(tmp_path), user_defined_parameters={"clip_compute_dtype": dtype}).cuda()
    return app, sd


@pytest.mark.parametrize("name", ["tiny_b6_l24", "small_b5_l40", "p14_w256_b16_l32", "vitb16_bertbase_b4_l64", "large_text_b24_l40"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_forward_matches_reference_golden(tmp_path, name, dtype):
    z, cfg, B, Lq, wseed, iseed = load_gold(name)
    app, _ = make_app(tmp_path, cfg, wseed, dtype)
    app.eval()
    px, ids = O.make_inputs(cfg, B, Lq, iseed)
    with torch.no_grad():
        out = app({"pixel_values": px, "input_ids": ids})
        loss = app.compute_loss(out, [])["loss"]
    assert set(out) == {"logits_per_text", "logits_per_image", "image_embeds", "text_embeds"}
    img, txt = out["image_embeds"].cpu(), out["text_embeds"].cpu()
    gi, gt = torch.from_numpy(z["image_embeds"]), torch.from_numpy(z["text_embeds"])
    lpt = out["logits_per_text"].cpu()
    assert torch.equal(out["logits_per_image"].cpu(), lpt.t())
    scale = float(np.exp(O.make_state_dict(cfg, wseed)["logit_scale"]))
    if dtype == "fp32":
        assert float((img - gi).abs().max()) < 1e-5
        assert float((txt - gt).abs().max()) < 1e-5
        assert float((lpt - torch.from_numpy(z["logits_per_text"])).abs().max()) < 2e-4 * scale / 14.3 * 2
        assert abs(loss.item() - float(z["loss"])) < 1e-5 * max(1.0, abs(float(z["loss"])))
    else:
        assert float((img - gi).abs().max()) < 1e-2
        assert float((txt - gt).abs().max()) < 1e-2
        assert float(torch.nn.functional.cosine_similarity(img, gi).min()) > 0.9995
        assert float(torch.nn.functional.cosine_similarity(txt, gt).min()) > 0.9995
        assert float((lpt - torch.from_numpy(z["logits_per_text"])).abs().max()) < 0.15
        # loss error is bounded by the logit error (<= 0.15 abs): 1/10 of that on these small batches
        assert abs(loss.item() - float(z["loss"])) < 1.5e-2


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_ragged_batches_and_single_modality(tmp_path, dtype):
    cfg = O.CONFIGS["small"]
    app, sd = make_app(tmp_path, cfg, 5, dtype)
    app.eval()
    tol = 2e-5 if dtype == "fp32" else 1e-2
    for B, Lq in [(1, 1), (3, 7), (7, 33), (2, 128)]:
        px, ids = O.make_inputs(cfg, B, Lq, B + Lq)
        with torch.no_grad():
            ref = O.clip_forward(sd, cfg, px, ids)
            oi = app({"pixel_values": px}, feat=True)
            ot = app({"input_ids": ids}, feat=True)
        assert oi["text_embeds"] is None and ot["image_embeds"] is None
        assert float((oi["image_embeds"].cpu() - ref["image_embeds"]).abs().max()) < tol
        assert float((ot["text_embeds"].cpu() - ref["text_embeds"]).abs().max()) < tol


def test_all_padding_rows_and_pad_id_inside_sequence(tmp_path):
    """mask = ids != 0 (modeling_chineseclip.py:347): zeros inside a sentence are masked keys too,
    and a fully padded row (all keys at -10000) must still give the reference's finite output."""
    cfg = O.CONFIGS["tiny"]
    app, sd = make_app(tmp_path, cfg, 9, "fp32")
    app.eval()
    _, ids = O.make_inputs(cfg, 4, 16, 2)
    ids[1, 3] = 0
    ids[1, 7] = 0
    ids[2, :] = 0
    with torch.no_grad():
        ref = O.encode_text(sd, cfg, ids)
        got = app({"input_ids": ids}, feat=True)["text_embeds"].cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) < 2e-5


def test_folded_layernorm_inference_path(tmp_path):
    """bf16 inference folds ln_1 / ln_2 of every ViT block into the in_proj / c_fc products
    (LN(x) W^T + b = rstd (x (W o g)^T) - rstd mean c1 + c2, GemmArgs::ln_stats), with the row statistics either left
    behind by the epilogue of the GEMM that writes the residual stream (mode 1, default) or computed by a separate pass
    (mode 2): same embeddings as the path with separate LayerNorm kernels (mode 0) to bf16 accuracy, and at least as
    close to the fp32 reference."""
    from easynlp_amd import lib as L
    z, cfg, B, Lq, wseed, iseed = load_gold("vitb16_bertbase_b4_l64")
    app, _ = make_app(tmp_path, cfg, wseed, "bf16")
    app.eval()
    B = 8     # (M = 8 * 197 rows: ragged against the 256-row tiles)
    px, ids = O.make_inputs(cfg, B, Lq, iseed)
    lib = L.load()
    outs = {}
    for mode in (1, 2, 0):
        L.check(lib.ezclip_debug_set(2, mode))
        try:
            with torch.no_grad():
                outs[mode] = app({"pixel_values": px, "input_ids": None}, feat=True)["image_embeds"].cpu().clone()
        finally:
            L.check(lib.ezclip_debug_set(2, 1))
    with torch.no_grad():
        gi = O.encode_image(O.make_state_dict(cfg, wseed), cfg, px)
    assert float((outs[1][:4] - torch.from_numpy(z["image_embeds"])).abs().max()) < 1e-2     # (same first 4 inputs)
    e = {k: float((v - gi).abs().max()) for k, v in outs.items()}
    assert float((outs[1] - outs[0]).abs().max()) < 1e-2 and float((outs[2] - outs[0]).abs().max()) < 1e-2
    # the fused statistics see exactly the values the separate pass reads: the two folded modes agree far below bf16 noise
    assert float((outs[1] - outs[2]).abs().max()) < 2e-3
    assert max(e.values()) < 1e-2
    assert e[1] < 1.5 * e[0] + 1e-3 and e[2] < 1.5 * e[0] + 1e-3, e
    # deterministic (fixed-order partial sums, no atomics)
    with torch.no_grad():
        again = app({"pixel_values": px, "input_ids": None}, feat=True)["image_embeds"].cpu()
    assert torch.equal(again, outs[1])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_text_tower_at_512_tokens(tmp_path, dtype):
    """BERT's full position range (max_position_embeddings 512, modeling_bert.py:88): in f32 the attention forward walks
    the keys in blocks above 288 tokens (attn_fwd_chunked_kernel), forward and backward against the oracle."""
    cfg = dict(O.CONFIGS["small"], text_max_position_embeddings=512)
    app, sd = make_app(tmp_path, cfg, 9, dtype)
    app.eval()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1, cfg["vocab_size"], (3, 512), generator=g)
    ids[1, 300:] = 0
    ids[2, 17:] = 0
    out = app({"input_ids": ids}, feat=True)["text_embeds"]
    w = torch.randn(out.shape, generator=g).to(out.device)
    (out * w).sum().backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.encode_text(sdr, cfg, ids)
    (ref * w.cpu()).sum().backward()
    tol = 2e-5 if dtype == "fp32" else 1e-2
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < tol
    for n in ("text_projection", "bert.encoder.layer.0.attention.self.value.weight", "bert.encoder.layer.1.output.dense.weight",
              "bert.embeddings.position_embeddings.weight"):
        got, want = app._params[n].grad.detach().cpu().double(), sdr[n].grad.double()
        assert float((got - want).norm()) <= (2e-4 if dtype == "fp32" else 8e-2) * float(want.norm()), n


@pytest.mark.parametrize("cfg_name,B,Lq", [("small", 9, 40), ("vitb16_bertbase", 6, 64)])
def test_packed_text_tower_equals_the_padded_one(tmp_path, cfg_name, B, Lq):
    """bf16 inference runs the BERT tower on the unmasked tokens only (ezclip_encode_text_packed): padded positions carry
    the -10000 key bias and only x[:, 0] is read (modeling_chineseclip.py:347-350), so nothing of them reaches the feature.
    Suffix padding: bit-identical embeddings (same rows through the same tiles, masked keys add exact zeros).  Pad ids
    inside a sentence, a masked CLS token, a sentence without any unmasked key (kept whole: its softmax is uniform over
    ALL positions in the reference): the oracle within the bf16 bound, the padded path within summation-order noise."""
    cfg = O.CONFIGS[cfg_name]
    app, sd = make_app(tmp_path, cfg, 4, "bf16")
    app.eval()
    eng = app._engine
    g = torch.Generator().manual_seed(B + Lq)
    ids = torch.randint(1, cfg["vocab_size"], (B, Lq), generator=g)
    lens = torch.randint(1, Lq + 1, (B,), generator=g)
    lens[0], lens[1] = Lq, 1
    ids = ids * (torch.arange(Lq)[None, :] < lens[:, None])
    res = {}
    for pack in (False, True):
        eng.pack_text = pack
        with torch.no_grad():
            res[pack] = app({"input_ids": ids.clone()}, feat=True)["text_embeds"].cpu()      # host ids: metadata from the host copy
            rows = eng.last_text_rows
            dev = app({"input_ids": ids.cuda()}, feat=True)["text_embeds"].cpu()             # device ids: metadata on the device
            fused = app.contrastive_step(torch.zeros(B, 3, cfg["image_resolution"], cfg["image_resolution"], device="cuda"),
                                         ids.cuda(), process_group=False)
        assert torch.equal(dev, res[pack]) and torch.isfinite(fused)
        assert rows == ((int(lens.sum()), B * Lq) if pack else (B * Lq, B * Lq)), rows
    assert torch.equal(res[True], res[False]), float((res[True] - res[False]).abs().max())
    with torch.no_grad():
        ref = O.encode_text(sd, cfg, ids)
    assert float((res[True] - ref).abs().max()) < 1e-2
    # the awkward rows
    ids2 = ids.clone()
    ids2[2, :] = 0                      # no unmasked key at all
    ids2[3, 0] = 0                      # masked CLS token (still the query whose output is used)
    ids2[4, 1] = 0
    if Lq > 5:
        ids2[4, 5] = 0                  # pad ids inside a sentence
    out2 = {}
    for pack in (False, True):
        eng.pack_text = pack
        with torch.no_grad():
            out2[pack] = app({"input_ids": ids2.clone()}, feat=True)["text_embeds"].cpu()
    eng.pack_text = True
    with torch.no_grad():
        ref2 = O.encode_text(sd, cfg, ids2)
    assert torch.isfinite(out2[True]).all()
    assert float((out2[True] - ref2).abs().max()) < 1e-2 and float((out2[False] - ref2).abs().max()) < 1e-2
    assert float((out2[True] - out2[False]).abs().max()) < 4e-3
    untouched = [i for i in range(B) if i not in (2, 3, 4)]
    assert torch.equal(out2[True][untouched], res[True][untouched])          # samples are independent


@pytest.mark.parametrize("path", ["fused", "autograd"])
def test_packed_text_tower_training_step_equals_the_padded_one(tmp_path, path):
    """Forward with save_for_backward + backward on the packed rows: same loss, and every parameter gradient equals the padded
    run's up to the summation order of the weight-gradient partials (a dropped token has exactly zero gradient in the padded run:
    its key is masked in every layer and its own outputs feed nothing)."""
    cfg = O.CONFIGS["small"]
    app, sd = make_app(tmp_path, cfg, 6, "bf16")
    app.eval()
    eng = app._engine
    B, Lq = 10, 40
    px, ids = O.make_inputs(cfg, B, Lq, 3)
    g = torch.Generator().manual_seed(8)
    lens = torch.randint(1, Lq + 1, (B,), generator=g)
    lens[0], lens[1] = Lq, 1
    ids = ids.clamp(min=1) * (torch.arange(Lq)[None, :] < lens[:, None])
    ids[2, :] = 0                       # no unmasked key at all: kept whole
    ids[3, 0] = 0                       # masked CLS token
    ids[4, 2] = 0                       # a pad id inside the sentence
    res = {}
    for pack in (False, True):
        eng.pack_text = pack
        for p in app.parameters():
            p.grad = None
        if path == "fused":
            loss = app.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=True, zero_grad=True)
        else:
            loss = app.compute_loss(app({"pixel_values": px, "input_ids": ids.clone()}), [])["loss"]
            loss.backward()
        torch.cuda.synchronize()
        rows = eng.last_text_rows
        assert (rows[0] < rows[1]) == pack, rows
        res[pack] = (float(loss.item()), {n: p.grad.detach().float().cpu().clone() for n, p in app._params.items() if p.grad is not None})
    eng.pack_text = True
    assert abs(res[True][0] - res[False][0]) < 2e-3
    assert set(res[True][1]) == set(res[False][1])
    # (key-bias gradients are mathematically zero -- softmax does not see a per-query constant -- i.e. pure rounding noise)
    floor = 1e-3 * max(float(v.norm()) for v in res[False][1].values())
    worst = max((float((res[True][1][n] - v).norm()) / (float(v.norm()) + floor), n) for n, v in res[False][1].items()
                if not n.endswith(".key.bias"))
    assert worst[0] < 2e-2, worst
    # ... and against the oracle (bf16 bound of the golden tests on the big matrices)
    _, ref_loss, ref_g = O.forward_loss_backward(sd, cfg, px, ids)
    assert abs(res[True][0] - float(ref_loss)) < 1.5e-2
    for n in ("text_projection", "bert.encoder.layer.0.intermediate.dense.weight", "bert.embeddings.word_embeddings.weight",
              "bert.embeddings.position_embeddings.weight", "bert.embeddings.token_type_embeddings.weight"):
        r, got = ref_g[n].double(), res[True][1][n].double()
        if n.endswith("word_embeddings.weight"):
            # nn.Embedding(padding_idx=0) (modeling_bert.py:77): torch gives row 0 no gradient; the oracle's plain indexing does
            # (only visible here, where pad tokens are queries that reach the loss)
            assert float(got[0].abs().max()) == 0.0
            r, got = r[1:], got[1:]
        assert float((got - r).norm()) < 8e-2 * float(r.norm()), n


@pytest.mark.parametrize("path", ["fused", "autograd"])
def test_packed_text_tower_with_dropout_equals_the_padded_one(tmp_path, path):
    """Train mode with the reference's dropout (hidden and attention probabilities): the masks are numbered by padded rows and
    positions, a packed batch whose sentences keep a prefix of their tokens regenerates exactly the padded run's decisions -- same
    loss, same gradients up to summation order.  A batch with a hole in a sentence is not packed under dropout."""
    cfg = dict(O.CONFIGS["small"], text_hidden_dropout_prob=0.1, text_attention_probs_dropout_prob=0.1)
    app, sd = make_app(tmp_path, cfg, 6, "bf16")
    app.train()
    eng = app._engine
    B, Lq = 12, 40
    px, ids = O.make_inputs(cfg, B, Lq, 3)
    g = torch.Generator().manual_seed(8)
    lens = torch.randint(1, Lq + 1, (B,), generator=g)
    lens[0], lens[1] = Lq - 3, 1        # (the longest sentence is SHORTER than the padded length: mask rows count padded positions)
    assert int(lens.max()) < Lq
    ids = ids.clamp(min=1) * (torch.arange(Lq)[None, :] < lens[:, None])
    ids[2, :] = 0                       # no unmasked key at all: kept whole (still a prefix)
    res, emb = {}, {}
    for pack in (False, True):
        eng.pack_text = pack
        torch.manual_seed(1234)
        with torch.no_grad():           # forward only, train mode: the same decisions -> the same embeddings
            emb[pack] = app({"pixel_values": None, "input_ids": ids.clone()}, feat=True)["text_embeds"].float().cpu()
        assert (eng.last_text_rows[0] < eng.last_text_rows[1]) == pack
    assert float((emb[True] - emb[False]).abs().max()) < 1e-3, float((emb[True] - emb[False]).abs().max())
    for pack in (False, True):
        eng.pack_text = pack
        for p in app.parameters():
            p.grad = None
        torch.manual_seed(1234)          # the dropout seed of the step is drawn from torch's CPU generator
        if path == "fused":
            loss = app.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=True, zero_grad=True)
        else:
            loss = app.compute_loss(app({"pixel_values": px, "input_ids": ids.clone()}), [])["loss"]
            loss.backward()
        torch.cuda.synchronize()
        rows = eng.last_text_rows
        assert (rows[0] < rows[1]) == pack, rows
        res[pack] = (float(loss.item()), {n: p.grad.detach().float().cpu().clone() for n, p in app._params.items() if p.grad is not None})
    assert abs(res[True][0] - res[False][0]) < 1e-3, (res[True][0], res[False][0])
    assert set(res[True][1]) == set(res[False][1])
    floor = 1e-3 * max(float(v.norm()) for v in res[False][1].values())
    worst = max((float((res[True][1][n] - v).norm()) / (float(v.norm()) + floor), n) for n, v in res[False][1].items()
                if not n.endswith(".key.bias"))
    assert worst[0] < 2e-2, worst
    # the dropout really was on: another seed gives another loss
    torch.manual_seed(99)
    other = float(app.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=False).item())
    assert abs(other - res[True][0]) > 1e-4
    # a hole inside a sentence: no packing while dropout is armed (a packed position would not be the padded one)
    ids2 = ids.clone()
    ids2[4, 2] = 0
    app.contrastive_step(px.cuda(), ids2.cuda(), process_group=False, backward=True, zero_grad=True)
    assert eng.last_text_rows[0] == eng.last_text_rows[1]
    app.eval()
    with torch.no_grad():
        app.contrastive_step(px.cuda(), ids2.cuda(), process_group=False, backward=False)
    assert eng.last_text_rows[0] < eng.last_text_rows[1]      # (eval mode: packed as before)
    eng.pack_text = True


class _DS(torch.utils.data.Dataset):
    def __init__(self, px, ids):
        self.px, self.ids = px, ids

    def __len__(self):
        return self.px.shape[0]

    def __getitem__(self, i):
        return {"pixel_values": self.px[i:i + 1], "text": {"input_ids": self.ids[i:i + 1]}}

    @staticmethod
    def batch_fn(features):   # shape of CLIPDataset.batch_fn (appzoo/clip/data.py:275-295)
        return {"pixel_values": torch.cat([f["pixel_values"] for f in features]),
                "input_ids": torch.cat([f["text"]["input_ids"] for f in features]), "label_ids": []}


def test_evaluator_and_predictor_contract(tmp_path):
    cfg = O.CONFIGS["tiny"]
    app, sd = make_app(tmp_path, cfg, 21, "fp32")
    px, ids = O.make_inputs(cfg, 24, 12, 4)
    ev = CLIPEvaluator(_DS(px, ids), eval_batch_size=7)
    res = ev.evaluate(app)
    assert res[0][0] == "mean_recall"
    with torch.no_grad():
        ref = O.clip_forward(sd, cfg, px, ids)
    want = O.recall_at_k(ref["text_embeds"], ref["image_embeds"])
    assert abs(res[0][1] - want[0]) <= 1e-3
    pred = CLIPPredictor(str(tmp_path), CLIPApp, user_defined_parameters={"clip_compute_dtype": "fp32"})
    rows = pred.run([{"input_ids": ids[i:i + 1]} for i in range(3)])
    assert len(rows) == 3 and set(rows[0]) == {"text_feat"}
    vals = np.array([float(x) for x in rows[0]["text_feat"].split("\t")])
    assert np.abs(vals - ref["text_embeds"][0].numpy()).max() < 2e-5
    rows = pred.run([{"pixel_values": px[i:i + 1]} for i in range(2)])
    assert set(rows[0]) == {"image_feat"}


def test_weights_refresh_after_inplace_update(tmp_path):
    cfg = O.CONFIGS["tiny"]
    app, sd = make_app(tmp_path, cfg, 2, "bf16")
    app.eval()
    px, ids = O.make_inputs(cfg, 3, 8, 1)
    with torch.no_grad():
        a = app({"pixel_values": px, "input_ids": ids}, feat=True)["image_embeds"].clone()
        app.chinese_clip.visual.proj.mul_(-1.0)    # optimizer-style in-place update
        b = app({"pixel_values": px, "input_ids": ids}, feat=True)["image_embeds"]
    assert float((a + b).abs().max()) < 1e-6


# ----------------------------------------------------------------------------- backward

def _grad_check(app, z, tol_rel, tol_norm, skip_tiny=1e-7, noise_factor=0.0, scale_floor=0.0):
    """Every parameter gradient against the reference's (tools/make_golden.py).

    fp32 pipeline: plain relative tolerances (1e-4).  bf16 pipeline: on these random-init fixtures the towers
    rank-collapse (all tokens of a sentence nearly equal after a few layers), so some gradients are
    near-cancellations: BERT query/key weights come out ~1e-3 of the value weight's gradient in the same layer,
    attention key biases are mathematically 0, logit_scale and the last layer's biases are small differences of
    large terms.  A bf16 implementation's error is bounded relative to the SCALE of the gradients of that kind,
    not relative to a near-cancelled result, so a parameter passes when its error is within
        max(tol, noise_factor * bf16dev[p]) * |ref_p|  +  scale_floor * S(p)
    where bf16dev[p] (stored in the fixture) is the deviation of a plain torch-CPU bfloat16 evaluation of the
    same algorithm from the fp32 reference, and S(p) is the largest reference gradient norm among the
    parameters of the same shape in the same tower (q/k/v/dense weights of all layers; all [D] vectors; ...).
    The fp32 pipeline checks the very same tensors at 1e-4 with no floor."""
    def group(n):
        tower = "visual" if n.startswith("visual") else ("bert" if n.startswith("bert") else n)
        return tower, tuple(app._params[n].shape)
    scale = {}
    for key in z.files:
        if key.startswith("gnorm/"):
            n = key[len("gnorm/"):]
            gk = group(n)
            scale[gk] = max(scale.get(gk, 0.0), float(z[key]))
    bad = []
    for key in z.files:
        if key.startswith("nograd/"):
            n = key[len("nograd/"):]
            p = app._params[n]
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        if not (key.startswith("gnorm/") or key.startswith("grad/")):
            continue
        n = key.split("/", 1)[1]
        noise = noise_factor * float(z["bf16dev/" + n]) if noise_factor > 0 else 0.0
        floor = skip_tiny + scale_floor * scale.get(group(n), 0.0)
        if key.startswith("gnorm/"):
            ref = float(z[key])
            got = float(app._params[n].grad.double().norm())
            if abs(got - ref) > max(tol_norm, noise) * ref + floor:
                bad.append((n, "norm", got, ref, noise, floor))
        else:
            ref = torch.from_numpy(z[key]).double()
            got = app._params[n].grad.detach().cpu().double()
            err = float((got - ref).norm())
            if err > max(tol_rel, noise) * float(ref.norm()) + floor:
                bad.append((n, "rel", err, float(ref.norm()), noise, floor))
    assert not bad, bad[:12]


@pytest.mark.parametrize("name", ["tiny_b6_l24", "small_b5_l40", "p14_w256_b16_l32", "vitb16_bertbase_b4_l64", "large_text_b24_l40"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("path", ["autograd", "fused"])
def test_backward_matches_reference_golden(tmp_path, name, dtype, path):
    """Gradients of every parameter vs the real reference's autograd (tools/make_golden.py).
    f32: rel-L2 <= 1e-4; bf16: max(6e-2, 1.5 x the measured deviation of a torch-CPU bf16 evaluation
    of the same algorithm) per parameter plus 1 % of the gradient scale of its kind -- see _grad_check."""
    z, cfg, B, Lq, wseed, iseed = load_gold(name)
    app, _ = make_app(tmp_path, cfg, wseed, dtype)
    app.train()
    px, ids = O.make_inputs(cfg, B, Lq, iseed)
    if path == "autograd":
        out = app({"pixel_values": px, "input_ids": ids})
        loss = app.compute_loss(out, [])["loss"]
        loss.backward()
    else:
        loss = app.contrastive_step(px.cuda(), ids.cuda(), process_group=False, backward=True)
    torch.cuda.synchronize()
    ltol = 1e-5 if dtype == "fp32" else 1.5e-2
    assert abs(loss.item() - float(z["loss"])) < ltol * max(1.0, abs(float(z["loss"])))
    if dtype == "fp32":
        _grad_check(app, z, tol_rel=1e-4, tol_norm=1e-4)
    else:
        _grad_check(app, z, tol_rel=6e-2, tol_norm=6e-2, skip_tiny=1e-5, noise_factor=1.5, scale_floor=1e-2)


def test_gradient_accumulation_and_optimizer_step(tmp_path):
    """Trainer-style loop (easynlp/core/trainer.py:626-661): loss.backward() twice accumulates,
    AdamW + clip_grad_norm_ run on the real nn.Parameters, and the next forward sees the update."""
    cfg = O.CONFIGS["tiny"]
    app, sd = make_app(tmp_path, cfg, 4, "fp32")
    app.train()
    px, ids = O.make_inputs(cfg, 6, 12, 3)
    named = dict(app.named_parameters())
    assert "chinese_clip.visual.transformer.resblocks.0.attn.in_proj_weight" in named
    opt = torch.optim.AdamW(app.parameters(), lr=1e-3)
    l0 = app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"]
    l0.backward()
    g1 = app._params["text_projection"].grad.clone()
    app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"].backward()
    assert float((app._params["text_projection"].grad - 2 * g1).abs().max()) < 1e-6 * float(g1.abs().max()) + 1e-9
    torch.nn.utils.clip_grad_norm_(app.parameters(), 1.0)
    opt.step()
    opt.zero_grad()
    l1 = app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"]
    assert l1.item() < l0.item()
    # the oracle with the updated weights agrees: the packed weight copies were refreshed
    sd2 = {k: app._params[k].detach().cpu() for k in sd}
    with torch.no_grad():
        ref = O.clip_loss(O.clip_forward(sd2, cfg, px, ids)["logits_per_text"])
    assert abs(ref.item() - l1.item()) < 1e-4


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_overfit_fixed_pairs_recall_matches_reference_math(tmp_path, dropout):
    """BASELINE north star: recall@1 on fixed synthetic pairs within 1e-3 of the reference.  Train the bf16 HIP path
    (train mode, reference dropout probabilities) on 16 fixed pairs until they are separable, then evaluate in eval mode:
    R@1/5/10 from CLIPEvaluator equal those of the fp32 CPU oracle run on the TRAINED weights."""
    cfg = dict(O.CONFIGS["tiny"], text_hidden_dropout_prob=dropout, text_attention_probs_dropout_prob=dropout)
    app, sd = make_app(tmp_path, cfg, 8, "bf16")
    px, ids = O.make_inputs(cfg, 16, 12, 5)
    torch.manual_seed(0)
    opt = torch.optim.AdamW(app.parameters(), lr=2e-3, weight_decay=0.0)
    app.train()
    losses = []
    for _ in range(60):
        opt.zero_grad()
        loss = app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(app.parameters(), 1.0)
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    ev = CLIPEvaluator(_DS(px, ids), eval_batch_size=8)
    res = ev.evaluate(app)                                     # eval(): dropout off
    sd2 = {k: app._params[k].detach().cpu() for k in sd}
    with torch.no_grad():
        ref = O.clip_forward(sd2, cfg, px, ids)
    want = O.recall_at_k(ref["text_embeds"], ref["image_embeds"])
    assert abs(res[0][1] - want[0]) <= 1e-3, (res, want)
    r1 = O.recall_at_k(ref["text_embeds"], ref["image_embeds"], ks=(1,))
    assert r1[-1] >= 0.9, r1                                    # the pairs really became separable


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("cfg_name,B,Lq", [("small", 5, 40), ("vitb16_bertbase", 4, 64), ("p14_w256", 16, 32)])
def test_last_block_on_cls_rows_only_equals_full_evaluation(tmp_path, dtype, cfg_name, B, Lq):
    """Inference path: the last block of each tower is evaluated for the CLS rows only (nothing else is read by
    ln_post(x[:, 0]) @ proj / bert(...)[0][:, 0] @ text_projection).  Same embeddings as the full evaluation
    (ezclip_debug_set(3, 0)) up to the rounding of the one-query attention kernel, and as the oracle."""
    cfg = O.CONFIGS[cfg_name]
    app, sd = make_app(tmp_path, cfg, 11, dtype)
    app.eval()
    px, ids = O.make_inputs(cfg, B, Lq, 2)
    ids[1, 5] = 0                         # a masked key inside a sentence
    lib = L.load()
    outs = {}
    for mode in (1, 0):
        L.check(lib.ezclip_debug_set(3, mode))
        try:
            with torch.no_grad():
                o = app({"pixel_values": px, "input_ids": ids}, feat=True)
                outs[mode] = (o["image_embeds"].cpu().clone(), o["text_embeds"].cpu().clone())
        finally:
            L.check(lib.ezclip_debug_set(3, 1))
    with torch.no_grad():
        ref = O.clip_forward(sd, cfg, px, ids)
    tol = 1e-5 if dtype == "fp32" else 1e-2
    for k, name in ((0, "image_embeds"), (1, "text_embeds")):
        assert float((outs[1][k] - outs[0][k]).abs().max()) < (2e-6 if dtype == "fp32" else 6e-3), name
        assert float((outs[1][k] - ref[name]).abs().max()) < tol, name


def test_training_path_last_block_on_cls_rows_equals_full_backward(tmp_path):
    """Training path: with the last block of each tower evaluated on the CLS rows only (default) the loss and every
    parameter gradient equal those of the all-token evaluation (ezclip_debug_set(4, 0)) -- fp32, so any slip shows."""
    cfg = O.CONFIGS["small"]
    px, ids = O.make_inputs(cfg, 5, 40, 9)
    lib = L.load()
    res = {}
    for mode in (1, 0):
        L.check(lib.ezclip_debug_set(4, mode))
        try:
            app, _ = make_app(tmp_path, cfg, 17, "fp32")
            app.train()
            loss = app.compute_loss(app({"pixel_values": px, "input_ids": ids}), [])["loss"]
            loss.backward()
            res[mode] = (loss.item(), {n: p.grad.detach().cpu().clone() for n, p in app._params.items() if p.grad is not None})
        finally:
            L.check(lib.ezclip_debug_set(4, 1))
    assert abs(res[1][0] - res[0][0]) < 1e-6
    assert set(res[1][1]) == set(res[0][1])
    for n, g0 in res[0][1].items():
        err = float((res[1][1][n] - g0).norm())
        assert err <= 2e-5 * float(g0.norm()) + 1e-8, (n, err, float(g0.norm()))


VITL14_SHALLOW = dict(O.CONFIGS["vitb16_bertbase"], embed_dim=768, vision_layers=3, vision_width=1024, vision_patch_size=14,
                      text_num_hidden_layers=2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_vitl14_width_towers_against_oracle(tmp_path, dtype):
    """BASELINE.json config 5's tower SHAPES (ViT-L/14: width 1024, 16 heads, 16 x 16 + 1 = 257 tokens, patch 14 -> K = 588 padded to
    640, MLP 4096, embed_dim 768; BERT-base-arch text tower), three / two blocks deep so that the CPU oracle stays quick: forward,
    loss and parameter gradients against the oracle.  257 tokens is the nine-wave case of the short attention kernels and the widest
    LayerNorm rows of the path; at 16 pairs the ViT products (4 112 rows) run on the persistent 8-phase GEMM."""
    cfg = VITL14_SHALLOW
    app, sd = make_app(tmp_path, cfg, 21, dtype)
    app.train()                                   # (dropout probabilities are 0 in this config)
    B, Lq = 16, 64
    px, ids = O.make_inputs(cfg, B, Lq, 5)
    out = app({"pixel_values": px, "input_ids": ids})
    loss = app.compute_loss(out, [])["loss"]
    loss.backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.clip_forward(sdr, cfg, px, ids)
    ref_loss = O.clip_loss(ref["logits_per_text"])
    ref_loss.backward()
    img, txt = out["image_embeds"].detach().cpu(), out["text_embeds"].detach().cpu()
    ri, rt = ref["image_embeds"].detach(), ref["text_embeds"].detach()
    if dtype == "fp32":
        assert float((img - ri).abs().max()) < 2e-5 and float((txt - rt).abs().max()) < 2e-5
        assert abs(loss.item() - ref_loss.item()) < 2e-5 * max(1.0, abs(ref_loss.item()))
    else:
        assert float((img - ri).abs().max()) < 1e-2 and float((txt - rt).abs().max()) < 1e-2
        assert float(torch.nn.functional.cosine_similarity(img, ri).min()) > 0.9995
        assert float(torch.nn.functional.cosine_similarity(txt, rt).min()) > 0.9995
        assert abs(loss.item() - ref_loss.item()) < 1.5e-2
    # every parameter the reference's backward reaches, against the oracle's autograd
    worst = []
    for n, p in app._params.items():
        want = sdr[n].grad
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        got = p.grad.detach().cpu().double()
        worst.append((float((got - want.double()).norm()), n, float(want.double().norm()), (n.split(".")[0], tuple(want.shape))))
    # bound: a fraction of the gradient's own norm + a fraction of the largest norm among same-shape parameters of the tower
    # (the key biases' gradients are mathematically zero -- softmax is shift-invariant -- and 1e-9 in the reference)
    by_shape = {}
    for _, _, wn, key in worst:
        by_shape[key] = max(by_shape.get(key, 0.0), wn)
    rel, floor = (3e-4, 1e-5) if dtype == "fp32" else (6e-2, 2e-2)
    worst = sorted(((err / (rel * wn + floor * by_shape[key] + 1e-12), n, err, wn) for err, n, wn, key in worst), reverse=True)
    print("worst gradient deviations (fraction of bound, name, |diff|, |ref|):", worst[:5])
    assert len(worst) > 60
    assert worst[0][0] < 1.0, worst[:5]
