"""CPU oracle for the ``huggingface_clip`` branch of the reference ``CLIPApp``
(easynlp/appzoo/clip/model.py:73-104 construction, :128-144 forward): ``RobertaModel`` pooled output ->
``text_projection`` (nn.Linear with bias) -> L2 normalise; ``CLIPVisionModel`` pooled output, DETACHED, ->
``vision_projection`` -> L2 normalise.  TEST INFRASTRUCTURE ONLY (see clip_oracle.py).

The towers are the ones clip_oracle.py already restates under other parameter names:
  CLIPVisionTransformer (modelzoo/models/clip/modeling_clip.py:112-140 embeddings, :173-270 attention with separate
  q/k/v projections and q scaled by head_dim**-0.5, :287-335 pre-LN layer with quick_gelu MLP, :731-787 pre_layrnorm /
  post_layernorm(x[:, 0])) == VisualTransformer with in_proj = [q; k; v];
  RobertaModel (modelzoo/models/roberta/modeling_roberta.py) == BertModel layers + pad-aware position ids
  (:1497-1510) + explicit token types / mask + RobertaPooler tanh(dense(x[:, 0])) (:550-562).
``to_chinese_names`` is that renaming; ``tools/make_golden.py`` pins the result against the reference classes themselves
(tests/golden/hf_*.npz).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import clip_oracle as O

HF_CONFIGS: Dict[str, dict] = {
    "hf_tiny": dict(
        text_config=dict(vocab_size=211, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                         num_attention_heads=2, max_position_embeddings=64, type_vocab_size=2, pad_token_id=0,
                         layer_norm_eps=1e-12, hidden_act="gelu", hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0),
        vision_config=dict(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                           image_size=64, patch_size=16, hidden_act="quick_gelu", layer_norm_eps=1e-5),
        projection_dim=64),
    # 5x5+1 = 26 vision tokens; 3 heads; text pad id 1 (RoBERTa convention) so that padding_idx handling is exercised
    "hf_small": dict(
        text_config=dict(vocab_size=523, hidden_size=192, intermediate_size=768, num_hidden_layers=3,
                         num_attention_heads=3, max_position_embeddings=128, type_vocab_size=2, pad_token_id=1,
                         layer_norm_eps=1e-5, hidden_act="gelu", hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0),
        vision_config=dict(hidden_size=192, intermediate_size=768, num_hidden_layers=3, num_attention_heads=3,
                           image_size=80, patch_size=16, hidden_act="quick_gelu", layer_norm_eps=1e-5),
        projection_dim=128),
    # the LARGE text tower of the pai-clip-commercial-large checkpoints (CLIPTextConfig defaults, configuration_clip.py:90-95:
    # hidden 1024, 16 heads, FFN 4096; RoBERTa pad id 1, eps 1e-5), two layers deep over a small patch-14 vision tower
    "hf_large_text": dict(
        text_config=dict(vocab_size=523, hidden_size=1024, intermediate_size=4096, num_hidden_layers=2,
                         num_attention_heads=16, max_position_embeddings=128, type_vocab_size=2, pad_token_id=1,
                         layer_norm_eps=1e-5, hidden_act="gelu", hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0),
        vision_config=dict(hidden_size=256, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
                           image_size=56, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5),
        projection_dim=256),
}


def chinese_style_config(cfg: dict) -> dict:
    """The CHINESE_CLIP-kwargs view of a huggingface_clip config.json (what the towers' shapes are)."""
    t, v = cfg["text_config"], cfg["vision_config"]
    return dict(model_type="chinese_clip", embed_dim=cfg["projection_dim"], image_resolution=v["image_size"],
                vision_layers=v["num_hidden_layers"], vision_width=v["hidden_size"], vision_patch_size=v["patch_size"],
                vocab_size=t["vocab_size"], text_attention_probs_dropout_prob=t.get("attention_probs_dropout_prob", 0.0),
                text_hidden_act="gelu", text_hidden_dropout_prob=t.get("hidden_dropout_prob", 0.0),
                text_hidden_size=t["hidden_size"], text_initializer_range=0.02,
                text_intermediate_size=t["intermediate_size"], text_max_position_embeddings=t["max_position_embeddings"],
                text_num_attention_heads=t["num_attention_heads"], text_num_hidden_layers=t["num_hidden_layers"],
                text_type_vocab_size=t["type_vocab_size"])


def param_shapes(cfg: dict) -> Dict[str, tuple]:
    """state_dict of the reference CLIPApp in huggingface_clip mode (model.py:81-104), parameters only."""
    t, v, E = cfg["text_config"], cfg["vision_config"], cfg["projection_dim"]
    H, F, W, Fv, P = t["hidden_size"], t["intermediate_size"], v["hidden_size"], v["intermediate_size"], v["patch_size"]
    Lv = (v["image_size"] // P) ** 2 + 1
    s = {"text_encoder.embeddings.word_embeddings.weight": (t["vocab_size"], H),
         "text_encoder.embeddings.position_embeddings.weight": (t["max_position_embeddings"], H),
         "text_encoder.embeddings.token_type_embeddings.weight": (t["type_vocab_size"], H),
         "text_encoder.embeddings.LayerNorm.weight": (H,), "text_encoder.embeddings.LayerNorm.bias": (H,)}
    for i in range(t["num_hidden_layers"]):
        p = f"text_encoder.encoder.layer.{i}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (H, H), (H,)
        s[p + "attention.output.LayerNorm.weight"], s[p + "attention.output.LayerNorm.bias"] = (H,), (H,)
        s[p + "intermediate.dense.weight"], s[p + "intermediate.dense.bias"] = (F, H), (F,)
        s[p + "output.dense.weight"], s[p + "output.dense.bias"] = (H, F), (H,)
        s[p + "output.LayerNorm.weight"], s[p + "output.LayerNorm.bias"] = (H,), (H,)
    s["text_encoder.pooler.dense.weight"], s["text_encoder.pooler.dense.bias"] = (H, H), (H,)
    vm = "vision_encoder.vision_model."
    s[vm + "embeddings.class_embedding"] = (W,)
    s[vm + "embeddings.patch_embedding.weight"] = (W, 3, P, P)
    s[vm + "embeddings.position_embedding.weight"] = (Lv, W)
    s[vm + "pre_layrnorm.weight"], s[vm + "pre_layrnorm.bias"] = (W,), (W,)
    for i in range(v["num_hidden_layers"]):
        p = vm + f"encoder.layers.{i}."
        for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (W, W), (W,)
        s[p + "layer_norm1.weight"], s[p + "layer_norm1.bias"] = (W,), (W,)
        s[p + "mlp.fc1.weight"], s[p + "mlp.fc1.bias"] = (Fv, W), (Fv,)
        s[p + "mlp.fc2.weight"], s[p + "mlp.fc2.bias"] = (W, Fv), (W,)
        s[p + "layer_norm2.weight"], s[p + "layer_norm2.bias"] = (W,), (W,)
    s[vm + "post_layernorm.weight"], s[vm + "post_layernorm.bias"] = (W,), (W,)
    s["text_projection.weight"], s["text_projection.bias"] = (E, H), (E,)
    s["vision_projection.weight"], s["vision_projection.bias"] = (E, W), (E,)
    s["logit_scale"] = (1,)
    return s


def make_state_dict(cfg: dict, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Deterministic weights (numpy RandomState): gains around 1, everything else small random, biases non-zero."""
    rs = np.random.RandomState(seed)
    sd = {}
    for n, shp in param_shapes(cfg).items():
        if n == "logit_scale":
            v = np.array([np.log(1 / 0.07)], np.float32)
        elif n.endswith("orm.weight") or n.endswith("norm1.weight") or n.endswith("norm2.weight"):
            v = 1.0 + 0.1 * rs.standard_normal(shp)
        elif n.endswith(".bias"):
            v = 0.05 * rs.standard_normal(shp)
        elif len(shp) == 1:
            v = 0.1 * rs.standard_normal(shp)
        else:
            fan_in = int(np.prod(shp[1:]))
            v = rs.standard_normal(shp) * (0.7 / np.sqrt(fan_in) if "embeddings" not in n else 0.1)
        sd[n] = torch.from_numpy(np.asarray(v, np.float32).reshape(shp))
    return sd


def make_inputs(cfg: dict, batch: int, seq_len: int, seed: int = 0):
    """pixels, input_ids (padded with the config's pad id), token_type_ids (a 0/1 split inside each sentence),
    attention_mask -- what BertTokenizer + CLIPDataset.batch_fn hand to the branch (data.py:275-295)."""
    rs = np.random.RandomState(seed)
    t, v = cfg["text_config"], cfg["vision_config"]
    pad = t["pad_token_id"]
    R = v["image_size"]
    px = rs.standard_normal((batch, 3, R, R)).astype(np.float32)
    ids = rs.randint(2, t["vocab_size"], size=(batch, seq_len)).astype(np.int64)
    lens = rs.randint(min(4, seq_len), seq_len + 1, size=(batch,))
    lens[0] = seq_len
    tt = np.zeros((batch, seq_len), np.int64)
    am = np.zeros((batch, seq_len), np.int64)
    for b in range(batch):
        ids[b, lens[b]:] = pad
        am[b, :lens[b]] = 1
        tt[b, lens[b] // 2:lens[b]] = 1
    return torch.from_numpy(px), torch.from_numpy(ids), torch.from_numpy(tt), torch.from_numpy(am)


def position_ids_from_input_ids(input_ids: torch.Tensor, pad: int) -> torch.Tensor:
    """create_position_ids_from_input_ids, roberta/modeling_roberta.py:1497-1510"""
    mask = input_ids.ne(pad).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad


def to_chinese_names(sd: Dict[str, torch.Tensor], cfg: dict) -> Dict[str, torch.Tensor]:
    """The towers under the CHINESE_CLIP names clip_oracle.py uses (+ the two projection biases)."""
    out = {}
    for k, v in sd.items():
        if k.startswith("text_encoder."):
            out["bert." + k[len("text_encoder."):]] = v
    vm = "vision_encoder.vision_model."
    out["visual.class_embedding"] = sd[vm + "embeddings.class_embedding"]
    out["visual.conv1.weight"] = sd[vm + "embeddings.patch_embedding.weight"]
    out["visual.positional_embedding"] = sd[vm + "embeddings.position_embedding.weight"]
    for a, b in (("ln_pre", "pre_layrnorm"), ("ln_post", "post_layernorm")):
        out[f"visual.{a}.weight"], out[f"visual.{a}.bias"] = sd[vm + b + ".weight"], sd[vm + b + ".bias"]
    for i in range(cfg["vision_config"]["num_hidden_layers"]):
        s, d = vm + f"encoder.layers.{i}.", f"visual.transformer.resblocks.{i}."
        out[d + "attn.in_proj_weight"] = torch.cat([sd[s + f"self_attn.{x}_proj.weight"] for x in "qkv"], dim=0)
        out[d + "attn.in_proj_bias"] = torch.cat([sd[s + f"self_attn.{x}_proj.bias"] for x in "qkv"], dim=0)
        for a, b in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                     ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            out[d + a + ".weight"], out[d + a + ".bias"] = sd[s + b + ".weight"], sd[s + b + ".bias"]
    out["visual.proj"] = sd["vision_projection.weight"].t()
    out["visual.proj_bias"] = sd["vision_projection.bias"]
    out["text_projection"] = sd["text_projection.weight"].t()
    out["text_projection_bias"] = sd["text_projection.bias"]
    out["logit_scale"] = sd["logit_scale"].reshape(())
    return out


def hf_clip_forward(sd: Dict[str, torch.Tensor], cfg: dict, pixels, input_ids, token_type_ids, attention_mask):
    """``CLIPApp.forward`` huggingface_clip branch, model.py:128-150."""
    csd, ccfg = to_chinese_names(sd, cfg), chinese_style_config(cfg)
    t = cfg["text_config"]
    pos = position_ids_from_input_ids(input_ids, t["pad_token_id"])
    x = O.bert_forward(csd, ccfg, input_ids, position_ids=pos, token_type_ids=token_type_ids,
                       attention_mask=attention_mask, ln_eps=t["layer_norm_eps"])
    pooled = torch.tanh(O.linear(x[:, 0], csd["bert.pooler.dense.weight"], csd["bert.pooler.dense.bias"]))   # roberta :550-562
    txt = O.l2_normalize(pooled @ csd["text_projection"] + csd["text_projection_bias"])                     # model.py:135-136
    vis = O.vit_forward(dict(csd, **{"visual.proj": torch.eye(ccfg["vision_width"], dtype=csd["visual.proj"].dtype)}),
                        ccfg, pixels)                                          # pooled = post_layernorm(x[:, 0])
    vis = vis.detach()                                                         # model.py:140
    img = O.l2_normalize(vis @ csd["visual.proj"] + csd["visual.proj_bias"])   # model.py:141-142
    lpt = (txt @ img.t()) * csd["logit_scale"].exp()
    return {"logits_per_text": lpt, "logits_per_image": lpt.t(), "image_embeds": img, "text_embeds": txt}


def forward_loss_backward(sd, cfg, pixels, input_ids, token_type_ids, attention_mask, dtype=torch.float32):
    sdd = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    out = hf_clip_forward(sdd, cfg, pixels.to(dtype), input_ids, token_type_ids, attention_mask)
    loss = O.clip_loss(out["logits_per_text"])
    loss.backward()
    grads = {k: (v.grad.detach() if v.grad is not None else None) for k, v in sdd.items()}
    return {k: v.detach() for k, v in out.items()}, loss.detach(), grads
