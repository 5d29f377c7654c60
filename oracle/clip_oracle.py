"""CPU oracle for the EasyNLP CLIP text-image retrieval hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``easynlp_amd`` (the product) may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

It is a plain restatement (torch CPU tensor arithmetic, explicit matmul /
softmax / layer-norm formulas -- no ``nn.MultiheadAttention``, no ``nn.Module``
from the reference) of what the reference computes on this path.  Every
function cites the reference file:line (relative to the EasyNLP checkout) it
follows.  The whole thing is dtype-parametric: run it in float32 for the parity
bar, in float64 as a tie-breaker.

Parity pin: the reference's own tests hold no numeric vectors for this path
(``tests/test_clip.py`` asserts nothing numeric and needs network fixtures),
so the oracle is pinned against outputs of the *reference code itself* run in
the build container -- ``tools/make_golden.py`` imports the reference
``CHINESE_CLIP``/``CLIPApp`` from ``/root/reference``, loads the deterministic
weights of :func:`make_state_dict` and commits the outputs under
``tests/golden/``.  ``tests/test_oracle.py`` checks this file against those
fixtures (and, when ``/root/reference`` is present, against the live
reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

# --------------------------------------------------------------------------
# configurations (schema = CHINESE_CLIP ctor kwargs,
# easynlp/modelzoo/models/clip/modeling_chineseclip.py:256-276)
# --------------------------------------------------------------------------

CONFIGS: Dict[str, dict] = {
    # head_dim is 64 in both towers (vision_heads = width // 64, :289)
    "tiny": dict(
        model_type="chinese_clip", embed_dim=64, image_resolution=64,
        vision_layers=2, vision_width=128, vision_patch_size=16,
        vocab_size=211, text_attention_probs_dropout_prob=0.0,
        text_hidden_act="gelu", text_hidden_dropout_prob=0.0,
        text_hidden_size=128, text_initializer_range=0.02,
        text_intermediate_size=512, text_max_position_embeddings=64,
        text_num_attention_heads=2, text_num_hidden_layers=2,
        text_type_vocab_size=2),
    # an odd-shaped small config: 5x5+1 = 26 tokens (ragged vs. 32-row tiles)
    "small": dict(
        model_type="chinese_clip", embed_dim=128, image_resolution=80,
        vision_layers=3, vision_width=192, vision_patch_size=16,
        vocab_size=523, text_attention_probs_dropout_prob=0.0,
        text_hidden_act="gelu", text_hidden_dropout_prob=0.0,
        text_hidden_size=192, text_initializer_range=0.02,
        text_intermediate_size=768, text_max_position_embeddings=128,
        text_num_attention_heads=3, text_num_hidden_layers=3,
        text_type_vocab_size=2),
    # a scaled-down ViT-L/14-style tower: patch 14 (3*14*14 = 588 is not a tile multiple: K is padded to 640 and the
    # conv gradient un-padded), widths that are multiples of 256 so that batch 16 reaches the 8-phase GEMM kernels
    "p14_w256": dict(
        model_type="chinese_clip", embed_dim=128, image_resolution=56,
        vision_layers=2, vision_width=256, vision_patch_size=14,
        vocab_size=523, text_attention_probs_dropout_prob=0.0,
        text_hidden_act="gelu", text_hidden_dropout_prob=0.0,
        text_hidden_size=256, text_initializer_range=0.02,
        text_intermediate_size=1024, text_max_position_embeddings=64,
        text_num_attention_heads=4, text_num_hidden_layers=2,
        text_type_vocab_size=2),
    # BASELINE.json configs 1-4: ViT-B/16 + BERT-base
    "vitb16_bertbase": dict(
        model_type="chinese_clip", embed_dim=512, image_resolution=224,
        vision_layers=12, vision_width=768, vision_patch_size=16,
        vocab_size=21128, text_attention_probs_dropout_prob=0.0,
        text_hidden_act="gelu", text_hidden_dropout_prob=0.0,
        text_hidden_size=768, text_initializer_range=0.02,
        text_intermediate_size=3072, text_max_position_embeddings=512,
        text_num_attention_heads=12, text_num_hidden_layers=12,
        text_type_vocab_size=2),
    # BASELINE.json config 5: ViT-L/14 + chinese-roberta-wwm-ext (BERT arch, base)
    "vitl14_robertabase": dict(
        model_type="chinese_clip", embed_dim=768, image_resolution=224,
        vision_layers=24, vision_width=1024, vision_patch_size=14,
        vocab_size=21128, text_attention_probs_dropout_prob=0.0,
        text_hidden_act="gelu", text_hidden_dropout_prob=0.0,
        text_hidden_size=768, text_initializer_range=0.02,
        text_intermediate_size=3072, text_max_position_embeddings=512,
        text_num_attention_heads=12, text_num_hidden_layers=12,
        text_type_vocab_size=2),
    # the LARGE text tower (chinese-roberta-wwm-ext-large / CLIPTextConfig's defaults, configuration_clip.py:90-95: hidden 1024,
    # 16 heads, FFN 4096), two layers deep over a tiny image tower: post-LN LayerNorm at D = 1024, 16-head attention, K = 1024 /
    # 4096 BERT products; 24 rows x 40 tokens reach the 8-phase GEMM kernels
    "large_text": dict(
        model_type="chinese_clip", embed_dim=256, image_resolution=64,
        vision_layers=2, vision_width=128, vision_patch_size=16,
        vocab_size=523, text_attention_probs_dropout_prob=0.0,
        text_hidden_act="gelu", text_hidden_dropout_prob=0.0,
        text_hidden_size=1024, text_initializer_range=0.02,
        text_intermediate_size=4096, text_max_position_embeddings=64,
        text_num_attention_heads=16, text_num_hidden_layers=2,
        text_type_vocab_size=2),
}

VIT_LN_EPS = 1e-5    # nn.LayerNorm default, modeling_chineseclip.py:170
BERT_LN_EPS = 1e-12  # modeling_chineseclip.py:311


def param_shapes(cfg: dict) -> Dict[str, tuple]:
    """Names/shapes of ``CHINESE_CLIP.state_dict()`` (float parameters only;
    verified against the instantiated reference in tests/test_oracle.py)."""
    W, E = cfg["vision_width"], cfg["embed_dim"]
    P, R = cfg["vision_patch_size"], cfg["image_resolution"]
    Lv = (R // P) ** 2 + 1
    H, F = cfg["text_hidden_size"], cfg["text_intermediate_size"]
    s: Dict[str, tuple] = {}
    s["visual.class_embedding"] = (W,)
    s["visual.positional_embedding"] = (Lv, W)
    s["visual.proj"] = (W, E)
    s["visual.conv1.weight"] = (W, 3, P, P)
    s["visual.ln_pre.weight"] = (W,)
    s["visual.ln_pre.bias"] = (W,)
    for i in range(cfg["vision_layers"]):
        p = f"visual.transformer.resblocks.{i}."
        s[p + "attn.in_proj_weight"] = (3 * W, W)
        s[p + "attn.in_proj_bias"] = (3 * W,)
        s[p + "attn.out_proj.weight"] = (W, W)
        s[p + "attn.out_proj.bias"] = (W,)
        s[p + "ln_1.weight"] = (W,)
        s[p + "ln_1.bias"] = (W,)
        s[p + "mlp.c_fc.weight"] = (4 * W, W)
        s[p + "mlp.c_fc.bias"] = (4 * W,)
        s[p + "mlp.c_proj.weight"] = (W, 4 * W)
        s[p + "mlp.c_proj.bias"] = (W,)
        s[p + "ln_2.weight"] = (W,)
        s[p + "ln_2.bias"] = (W,)
    s["visual.ln_post.weight"] = (W,)
    s["visual.ln_post.bias"] = (W,)
    s["bert.embeddings.word_embeddings.weight"] = (cfg["vocab_size"], H)
    s["bert.embeddings.position_embeddings.weight"] = (cfg["text_max_position_embeddings"], H)
    s["bert.embeddings.token_type_embeddings.weight"] = (cfg["text_type_vocab_size"], H)
    s["bert.embeddings.LayerNorm.weight"] = (H,)
    s["bert.embeddings.LayerNorm.bias"] = (H,)
    for i in range(cfg["text_num_hidden_layers"]):
        p = f"bert.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (H, H)
            s[p + f"attention.self.{n}.bias"] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (F, H)
        s[p + "intermediate.dense.bias"] = (F,)
        s[p + "output.dense.weight"] = (H, F)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    s["bert.pooler.dense.weight"] = (H, H)
    s["bert.pooler.dense.bias"] = (H,)
    s["text_projection"] = (H, E)
    s["logit_scale"] = ()
    return s


RESIDUAL_OUT = ("attn.out_proj.weight", "mlp.c_proj.weight", "attention.output.dense.weight", "output.dense.weight")


def make_state_dict(cfg: dict, seed: int = 1234, residual_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights, independent of torch's initialisers
    (``numpy.random.RandomState`` streams are frozen across numpy versions).

    ``residual_gain`` multiplies the weight of every residual branch's LAST linear layer (ViT ``attn.out_proj`` / ``mlp.c_proj``, BERT
    ``attention.output.dense`` / ``output.dense``).  At 1 a 12 + 12-layer random-init model is rank-collapsed (uniform attention averages the
    tokens): its query / key gradients are differences of nearly equal terms and carry no information in ANY bf16 pipeline (activation-rounding
    floor up to 0.6 of their norm).  At 0.3 every block is a moderate update of its residual stream, as in a trained model: the floor is
    2.6e-2 in the median and 4e-2 at worst (round 6; sharpening the attention instead makes the model chaotic: floor > 1).

    Scales mimic the reference init (``VisualTransformer.__init__``
    modeling_chineseclip.py:226-234, ``BertPreTrainedModel._init_weights``
    bert/modeling_bert.py:624-638, ``text_projection`` :337) but LayerNorm
    gains/biases and Linear biases are *randomised* (reference: 1/0/0) so that a
    kernel that drops a bias or a gain cannot pass parity.
    """
    rs = np.random.RandomState(seed)
    W = cfg["vision_width"]
    H = cfg["text_hidden_size"]
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name == "logit_scale":
            v = np.asarray(math.log(1.0 / 0.07), dtype=np.float32)
        elif name.endswith("LayerNorm.weight") or ".ln_" in name and name.endswith(".weight"):
            v = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith("LayerNorm.bias") or ".ln_" in name and name.endswith(".bias"):
            v = 0.05 * rs.standard_normal(shape)
        elif name.endswith(".bias") or name.endswith("in_proj_bias"):
            v = 0.02 * rs.standard_normal(shape)
        elif name in ("visual.class_embedding", "visual.positional_embedding", "visual.proj"):
            v = (W ** -0.5) * rs.standard_normal(shape)
        elif name == "text_projection":
            v = (H ** -0.5) * rs.standard_normal(shape)
        elif name == "visual.conv1.weight":
            fan_in = shape[1] * shape[2] * shape[3]
            v = (fan_in ** -0.5) * rs.standard_normal(shape)
        elif name.startswith("visual."):
            # nn.Linear / MultiheadAttention style: ~ in_features ** -0.5
            v = (shape[-1] ** -0.5) * rs.standard_normal(shape)
        else:  # bert.*  (initializer_range 0.02) -- use a livelier 0.04
            v = 0.04 * rs.standard_normal(shape)
        if residual_gain != 1.0 and name.endswith(RESIDUAL_OUT):
            v = v * residual_gain
        sd[name] = torch.from_numpy(np.asarray(v, dtype=np.float32).copy())
    return sd


def make_inputs(cfg: dict, batch: int, seq_len: int, seed: int = 0):
    """Synthetic batch: pixels ~ N(0,1) fp32 NCHW, ids uniform in
    [1, vocab) with a random-length zero-padded tail (exercises ``text.ne(0)``,
    modeling_chineseclip.py:347-348).  Sample 0 is always full length."""
    rs = np.random.RandomState(seed)
    R = cfg["image_resolution"]
    px = rs.standard_normal((batch, 3, R, R)).astype(np.float32)
    ids = rs.randint(1, cfg["vocab_size"], size=(batch, seq_len)).astype(np.int64)
    lo = min(8, seq_len)
    lens = rs.randint(lo, seq_len + 1, size=(batch,))
    lens[0] = seq_len
    for b in range(batch):
        ids[b, lens[b]:] = 0
    return torch.from_numpy(px), torch.from_numpy(ids)


# --------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------

def layer_norm(x, w, b, eps):
    """torch.nn.LayerNorm over the last dim (biased variance).
    ViT: modeling_chineseclip.py:170-176 (fp32 upcast, eps 1e-5);
    BERT: bert/modeling_bert.py:83,261,339 (eps = config.layer_norm_eps)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def quick_gelu(x):
    """modeling_chineseclip.py:179-181."""
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x):
    """F.gelu (erf form) -- easynlp/modelzoo/activations.py:45-48,98."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x, w, b=None):
    """nn.Linear: y = x W^T + b, W is [out, in]."""
    y = x @ w.t()
    return y if b is None else y + b


# --------------------------------------------------------------------------
# vision tower
# --------------------------------------------------------------------------

def patchify(pixels, patch):
    """im2col of a stride==kernel conv: [B,3,R,R] -> [B, G*G, 3*P*P] with the
    inner index ordered (c, ky, kx) -- the flattening of conv1.weight
    [W,3,P,P] (modeling_chineseclip.py:224,237-239)."""
    B, C, R, _ = pixels.shape
    G = R // patch
    x = pixels[:, :, :G * patch, :G * patch].reshape(B, C, G, patch, G, patch)
    x = x.permute(0, 2, 4, 1, 3, 5).reshape(B, G * G, C * patch * patch)
    return x


def mha_self_attention(x, in_w, in_b, out_w, out_b, heads, key_bias=None,
                       scale_q_first=True, taps=None, tap_prefix="", attn_mask=None):
    """Multi-head self-attention on batch-first x [B, L, D].

    ViT: ``nn.MultiheadAttention(d_model, n_head)`` called with
    need_weights=False, attn_mask=None (modeling_chineseclip.py:188,198-200):
    packed in-projection rows [0:D]=Q, [D:2D]=K, [2D:3D]=V, q scaled by
    hd**-0.5, softmax(QK^T)V, out_proj.  (The reference permutes to [L,B,D]
    first; attention is per-sample so the layout is immaterial.)
    """
    B, L, D = x.shape
    hd = D // heads
    qkv = linear(x, in_w, in_b)
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(B, L, heads, hd).transpose(1, 2)
    k = k.reshape(B, L, heads, hd).transpose(1, 2)
    v = v.reshape(B, L, heads, hd).transpose(1, 2)
    scale = hd ** -0.5
    if scale_q_first:
        s = (q * scale) @ k.transpose(-1, -2)
    else:
        s = (q @ k.transpose(-1, -2)) * scale
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    if attn_mask is not None:            # additive [L, L] mask (open_clip text tower: -inf above the diagonal)
        s = s + attn_mask
    p = torch.softmax(s, dim=-1)
    ctx = (p @ v).transpose(1, 2).reshape(B, L, D)
    if taps is not None:
        taps[tap_prefix + "qkv"] = qkv
        taps[tap_prefix + "ctx"] = ctx
    return linear(ctx, out_w, out_b)


def vit_forward(sd, cfg, pixels, taps: Optional[dict] = None):
    """``VisualTransformer.forward`` modeling_chineseclip.py:236-253 +
    ``ResidualAttentionBlock.forward`` :202-205.  Returns un-normalised image
    features [B, E]."""
    dt = sd["visual.proj"].dtype
    P, W = cfg["vision_patch_size"], cfg["vision_width"]
    heads = W // 64                                           # :289
    eps = cfg.get("block_ln_eps", VIT_LN_EPS)           # wukong builds its LayerNorms with 1e-7 (wukong_oracle.py)
    x = patchify(pixels.to(dt), P) @ sd["visual.conv1.weight"].reshape(W, -1).t()   # :237-239
    B = x.shape[0]
    cls = sd["visual.class_embedding"].expand(B, 1, W)         # :240
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]   # :241
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], eps)  # :242
    if taps is not None:
        taps["vit.ln_pre"] = x
    for i in range(cfg["vision_layers"]):
        p = f"visual.transformer.resblocks.{i}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
        x = x + mha_self_attention(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                                   sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"],
                                   heads, taps=taps, tap_prefix=f"vit.{i}.")       # :203
        h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
        h = quick_gelu(linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
        x = x + linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])   # :204
        if taps is not None:
            taps[f"vit.{i}.out"] = x
    x = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], eps)  # :248
    return x @ sd["visual.proj"]                                # :250-251


# --------------------------------------------------------------------------
# text tower
# --------------------------------------------------------------------------

def apply_dropout(x, keep, p):
    """``nn.Dropout`` in train mode with an explicit keep mask: ``x * keep / (1 - p)``
    (torch.nn.functional.dropout: survivors are scaled by 1/(1-p))."""
    return x * keep.to(x.dtype) * (1.0 / (1.0 - p))


def bert_forward(sd, cfg, input_ids, taps: Optional[dict] = None, dropout: Optional[dict] = None,
                 position_ids=None, token_type_ids=None, attention_mask=None, ln_eps: float = None):
    """``dropout`` (train mode, modeling_bert.py:128,238,266,344): ``{"p_hidden", "p_attn", "masks"}`` with keep masks
    ``masks["emb"]`` [B,L,H], ``masks[f"{i}.attn"]`` [B,heads,L,L], ``masks[f"{i}.self_out"]`` / ``masks[f"{i}.out"]``
    [B,L,H] -- explicit masks instead of an RNG stream, so that any implementation's masks can be replayed here.

    ``BertModel.forward`` bert/modeling_bert.py:792-920 as driven by
    ``CHINESE_CLIP.encode_text`` (modeling_chineseclip.py:346-350): mask =
    ids != 0, token types all 0, positions 0..L-1, post-LN layers, erf-GELU,
    additive mask -10000 (modeling_utils.py:438-439), scores scaled *after*
    QK^T (modeling_bert.py:210,228).  Returns last_hidden_state [B, L, H]
    (the pooler output is computed by the reference but unused on this path)."""
    dt = sd["text_projection"].dtype
    B, L = input_ids.shape
    H = cfg["text_hidden_size"]
    heads = cfg["text_num_attention_heads"]
    hd = H // heads
    # (huggingface_clip branch: RobertaModel is called with explicit token types and mask, and RobertaEmbeddings derives
    #  pad-aware position ids -- the three optional arguments; hf_clip_oracle.py passes them)
    eps = BERT_LN_EPS if ln_eps is None else ln_eps
    mask = (input_ids.ne(0) if attention_mask is None else attention_mask.ne(0)).to(dt)    # chineseclip:347-348
    key_bias = (1.0 - mask) * -10000.0                          # modeling_utils.py:438-439
    tt = (sd["bert.embeddings.token_type_embeddings.weight"][0] if token_type_ids is None
          else sd["bert.embeddings.token_type_embeddings.weight"][token_type_ids])
    pe = (sd["bert.embeddings.position_embeddings.weight"][:L] if position_ids is None
          else sd["bert.embeddings.position_embeddings.weight"][position_ids])
    x = sd["bert.embeddings.word_embeddings.weight"][input_ids] + tt + pe   # modeling_bert.py:117-125
    x = layer_norm(x, sd["bert.embeddings.LayerNorm.weight"], sd["bert.embeddings.LayerNorm.bias"], eps)
    dm = dropout["masks"] if dropout is not None else None
    ph = dropout["p_hidden"] if dropout is not None else 0.0
    pa = dropout["p_attn"] if dropout is not None else 0.0
    if dm is not None and ph > 0:
        x = apply_dropout(x, dm["emb"], ph)                      # modeling_bert.py:128
    if taps is not None:
        taps["bert.emb"] = x
    for i in range(cfg["text_num_hidden_layers"]):
        p = f"bert.encoder.layer.{i}."
        q = linear(x, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"])
        k = linear(x, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"])
        v = linear(x, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"])
        q = q.reshape(B, L, heads, hd).transpose(1, 2)
        k = k.reshape(B, L, heads, hd).transpose(1, 2)
        v = v.reshape(B, L, heads, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + key_bias[:, None, None, :]   # :210,228,231
        pr = torch.softmax(s, dim=-1)                                               # :234
        if dm is not None and pa > 0:
            pr = apply_dropout(pr, dm[f"{i}.attn"], pa)                              # :238
        ctx = (pr @ v).transpose(1, 2).reshape(B, L, H)                              # :244-248
        so = linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        if dm is not None and ph > 0:
            so = apply_dropout(so, dm[f"{i}.self_out"], ph)                          # :266
        a = layer_norm(so + x,
                       sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"],
                       eps)                                                  # :264-267
        h = gelu_erf(linear(a, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))  # :330-331
        o = linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        if dm is not None and ph > 0:
            o = apply_dropout(o, dm[f"{i}.out"], ph)                                 # :344
        x = layer_norm(o + a,
                       sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)  # :342-345
        if taps is not None:
            taps[f"bert.{i}.ctx"] = ctx
            taps[f"bert.{i}.out"] = x
    return x


# --------------------------------------------------------------------------
# dual encoder, similarity, loss, recall
# --------------------------------------------------------------------------

def l2_normalize(x):
    """x / x.norm(dim=-1, keepdim=True), no epsilon (modeling_chineseclip.py:360,363)."""
    return x / x.norm(dim=-1, keepdim=True)


def encode_image(sd, cfg, pixels, taps=None):
    return l2_normalize(vit_forward(sd, cfg, pixels, taps))


def encode_text(sd, cfg, input_ids, taps=None, dropout=None):
    x = bert_forward(sd, cfg, input_ids, taps, dropout)
    return l2_normalize(x[:, 0, :] @ sd["text_projection"])    # chineseclip:349-350,363


def clip_forward(sd, cfg, pixels, input_ids, taps=None, dropout=None):
    """``CLIPApp.forward`` easynlp/appzoo/clip/model.py:106-150 (chinese_clip
    branch): logits_per_text = T @ I^T * exp(logit_scale)."""
    img = encode_image(sd, cfg, pixels, taps)
    txt = encode_text(sd, cfg, input_ids, taps, dropout)
    lpt = (txt @ img.t()) * sd["logit_scale"].exp()              # model.py:148
    return {"logits_per_text": lpt, "logits_per_image": lpt.t(),
            "image_embeds": img, "text_embeds": txt}


def cross_entropy_diag(logits):
    """F.cross_entropy(logits, arange(N)), mean reduction (model.py:154-155)."""
    lse = torch.logsumexp(logits, dim=-1)
    return (lse - logits.diagonal()).mean()


def clip_loss(logits_per_text):
    """``CLIPApp.clip_loss`` model.py:157-160."""
    return (cross_entropy_diag(logits_per_text) + cross_entropy_diag(logits_per_text.t())) / 2.0


def global_clip_loss_rank(txt_all, img_all, logit_scale, rank, n_local):
    """Per-rank share of the *global* contrastive loss (the north-star's
    all-gathered mode; no reference code -- oracle = ``clip_loss`` on the
    concatenated batch).  Rank r owns rows [r*n, (r+1)*n) of both directions;
    summing the returned value over ranks and dividing by world size gives
    ``clip_loss`` of the full [N, N] logits."""
    ls = logit_scale.exp()
    sl = slice(rank * n_local, (rank + 1) * n_local)
    s_t = (txt_all[sl] @ img_all.t()) * ls      # local text rows vs all images
    s_i = (img_all[sl] @ txt_all.t()) * ls      # local image rows vs all texts
    idx = torch.arange(rank * n_local, (rank + 1) * n_local)
    lt = torch.logsumexp(s_t, -1) - s_t[torch.arange(n_local), idx]
    li = torch.logsumexp(s_i, -1) - s_i[torch.arange(n_local), idx]
    return (lt.mean() + li.mean()) / 2.0


def recall_at_k(text_embeds, image_embeds, ks=(1, 5, 10)):
    """``CLIPEvaluator.evaluate`` easynlp/appzoo/clip/evaluator.py:47-67:
    text->image retrieval; hit if the paired index is among the top-k of a
    *descending full sort* of row idx.  Returns (mean_recall, r1, r5, r10) as
    fractions (the reference multiplies by 100 only for printing)."""
    agreement = text_embeds @ image_embeds.t()
    n = agreement.shape[0]
    hits = [0 for _ in ks]
    for idx in range(n):
        _, ridx = torch.sort(agreement[idx], descending=True)
        for j, k in enumerate(ks):
            if idx in ridx[:k]:
                hits[j] += 1
    rs = [h / n for h in hits]
    return (sum(rs) / len(rs),) + tuple(rs)


def recall_ranks(text_embeds, image_embeds):
    """Ranks behind ``recall_at_k``, both retrieval directions, as int64 tensors:
    ``t2i[i]`` = position of image i in the descending STABLE sort of row i of ``text @ image.T`` (the reference's loop,
    easynlp/appzoo/clip/evaluator.py:53-61, sorts unstably: equal scores may fall either way there; the stable order -- ties
    resolved by index -- is the one the HIP path defines), ``i2t[j]`` = position of text j in the stable sort of column j
    (image -> text: not in the reference)."""
    agreement = text_embeds @ image_embeds.t()
    n = agreement.shape[0]
    t2i = torch.empty(n, dtype=torch.int64)
    i2t = torch.empty(n, dtype=torch.int64)
    for idx in range(n):
        order = torch.sort(agreement[idx], descending=True, stable=True).indices
        t2i[idx] = int((order == idx).nonzero()[0, 0])
        order = torch.sort(agreement[:, idx], descending=True, stable=True).indices
        i2t[idx] = int((order == idx).nonzero()[0, 0])
    return t2i, i2t


def to_dtype(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}


def forward_loss_backward(sd, cfg, pixels, input_ids, dtype=torch.float32, dropout=None):
    """fwd + InfoNCE + autograd backward of the restatement; returns
    (outputs, loss, grads by reference parameter name).  ``bert.pooler.*``
    gets no gradient, as in the reference."""
    sdd = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    out = clip_forward(sdd, cfg, pixels.to(dtype), input_ids, dropout=dropout)
    loss = clip_loss(out["logits_per_text"])
    loss.backward()
    grads = {k: (v.grad.detach() if v.grad is not None else None) for k, v in sdd.items()}
    return {k: v.detach() for k, v in out.items()}, loss.detach(), grads
