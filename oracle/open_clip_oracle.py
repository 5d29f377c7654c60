"""CPU oracle for the ``open_clip`` branch of the reference ``CLIPApp`` (easynlp/appzoo/clip/model.py:56-64,124-125):
``OPEN_CLIP`` (easynlp/modelzoo/models/clip/modeling_openclip.py:255-385) = the VisualTransformer of clip_oracle.py +
a CLIP text transformer.  TEST INFRASTRUCTURE ONLY (see clip_oracle.py).

Text tower (``encode_text`` :354-368): ``token_embedding[text] + positional_embedding`` -> pre-LN residual attention
blocks (the ViT's block, with the additive causal mask of ``build_attention_mask`` :343-349) -> ``ln_final`` -> the row of
the EOT token (``text.argmax(-1)``) ``@ text_projection``; both features L2-normalised (:381-384).  Pinned against the
real ``OPEN_CLIP`` / ``CLIPApp`` by tools/make_golden.py (tests/golden/openclip_*.npz) and live when the checkout is there.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import clip_oracle as O

OPENCLIP_CONFIGS: Dict[str, dict] = {
    "oc_tiny": dict(model_type="open_clip", embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128,
                    vision_patch_size=16, context_length=20, vocab_size=301, transformer_width=128, transformer_heads=2,
                    transformer_layers=2),
    # 26 vision tokens, 77-token context (the real context length), 3 heads
    "oc_small": dict(model_type="open_clip", embed_dim=128, image_resolution=80, vision_layers=3, vision_width=192,
                     vision_patch_size=16, context_length=77, vocab_size=523, transformer_width=192, transformer_heads=3,
                     transformer_layers=3),
}


def chinese_style_config(cfg: dict) -> dict:
    """vision-tower view for clip_oracle.vit_forward"""
    out = dict(vision_patch_size=cfg["vision_patch_size"], vision_width=cfg["vision_width"], vision_layers=cfg["vision_layers"],
               image_resolution=cfg["image_resolution"], embed_dim=cfg["embed_dim"])
    if "block_ln_eps" in cfg:
        out["block_ln_eps"] = cfg["block_ln_eps"]
    return out


def param_shapes(cfg: dict) -> Dict[str, tuple]:
    W, E, P = cfg["vision_width"], cfg["embed_dim"], cfg["vision_patch_size"]
    Lv = (cfg["image_resolution"] // P) ** 2 + 1
    T, V, C = cfg["transformer_width"], cfg["vocab_size"], cfg["context_length"]
    s = {"visual.class_embedding": (W,), "visual.positional_embedding": (Lv, W), "visual.proj": (W, E),
         "visual.conv1.weight": (W, 3, P, P), "visual.ln_pre.weight": (W,), "visual.ln_pre.bias": (W,),
         "visual.ln_post.weight": (W,), "visual.ln_post.bias": (W,)}

    def blocks(prefix, n, D):
        for i in range(n):
            p = f"{prefix}transformer.resblocks.{i}."
            s[p + "attn.in_proj_weight"], s[p + "attn.in_proj_bias"] = (3 * D, D), (3 * D,)
            s[p + "attn.out_proj.weight"], s[p + "attn.out_proj.bias"] = (D, D), (D,)
            s[p + "ln_1.weight"], s[p + "ln_1.bias"], s[p + "ln_2.weight"], s[p + "ln_2.bias"] = (D,), (D,), (D,), (D,)
            s[p + "mlp.c_fc.weight"], s[p + "mlp.c_fc.bias"] = (4 * D, D), (4 * D,)
            s[p + "mlp.c_proj.weight"], s[p + "mlp.c_proj.bias"] = (D, 4 * D), (D,)
    blocks("visual.", cfg["vision_layers"], W)
    blocks("", cfg["transformer_layers"], T)
    s["token_embedding.weight"] = (V, T)
    s["positional_embedding"] = (C, T)
    s["ln_final.weight"], s["ln_final.bias"] = (T,), (T,)
    s["text_projection"] = (T, E)
    s["logit_scale"] = ()
    return s


def make_state_dict(cfg: dict, seed: int = 1234) -> Dict[str, torch.Tensor]:
    rs = np.random.RandomState(seed)
    sd = {}
    for n, shp in param_shapes(cfg).items():
        if n == "logit_scale":
            v = np.array(np.log(1 / 0.07), np.float32)
        elif n.endswith(".weight") and (".ln_" in n or n.startswith("ln_") or "ln_pre" in n or "ln_post" in n or "ln_final" in n):
            v = 1.0 + 0.1 * rs.standard_normal(shp)
        elif n.endswith("bias"):
            v = 0.05 * rs.standard_normal(shp)
        elif len(shp) <= 1:
            v = 0.1 * rs.standard_normal(shp)
        elif n in ("token_embedding.weight", "positional_embedding", "visual.positional_embedding"):
            v = 0.1 * rs.standard_normal(shp)
        elif n in ("visual.proj", "text_projection"):
            v = rs.standard_normal(shp) * shp[0] ** -0.5
        else:
            v = rs.standard_normal(shp) * 0.7 / np.sqrt(int(np.prod(shp[1:])))
        sd[n] = torch.from_numpy(np.asarray(v, np.float32).reshape(shp))
    return sd


def make_inputs(cfg: dict, batch: int, seed: int = 0):
    """pixels + BPE-style ids [B, context_length]: SOT (vocab-2), random tokens, EOT (vocab-1, the maximum), zero padding."""
    rs = np.random.RandomState(seed)
    R, C, V = cfg["image_resolution"], cfg["context_length"], cfg["vocab_size"]
    px = rs.standard_normal((batch, 3, R, R)).astype(np.float32)
    ids = np.zeros((batch, C), np.int64)
    lens = rs.randint(3, C + 1, size=(batch,))
    lens[0] = C
    for b in range(batch):
        ids[b, 0] = V - 2
        ids[b, 1:lens[b] - 1] = rs.randint(1, V - 2, size=(lens[b] - 2,))
        ids[b, lens[b] - 1] = V - 1
    return torch.from_numpy(px), torch.from_numpy(ids)


def text_forward(sd, cfg, text):
    """OPEN_CLIP.encode_text, modeling_openclip.py:354-368 (cfg keys block_ln_eps / eot_id: the wukong variant, wukong_oracle.py)"""
    T, heads = cfg["transformer_width"], cfg["transformer_heads"]
    eps = cfg.get("block_ln_eps", O.VIT_LN_EPS)
    B, L = text.shape
    x = sd["token_embedding.weight"][text] + sd["positional_embedding"][:L]                 # :355-357
    mask = torch.full((L, L), float("-inf"), dtype=x.dtype).triu_(1)                        # :343-349
    for i in range(cfg["transformer_layers"]):
        p = f"transformer.resblocks.{i}."
        h = O.layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
        x = x + O.mha_self_attention(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], sd[p + "attn.out_proj.weight"],
                                     sd[p + "attn.out_proj.bias"], heads, attn_mask=mask)
        h = O.layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
        h = O.quick_gelu(O.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
        x = x + O.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    x = O.layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"], eps)          # :361
    if cfg.get("eot_id") is not None:
        return x[(text == cfg["eot_id"]).nonzero(as_tuple=True)] @ sd["text_projection"]   # modeling_wukong.py:349,359-360
    return x[torch.arange(B), text.argmax(dim=-1)] @ sd["text_projection"]                  # :366


def open_clip_forward(sd, cfg, pixels, text):
    img = O.l2_normalize(O.vit_forward(sd, chinese_style_config(cfg), pixels))
    txt = O.l2_normalize(text_forward(sd, cfg, text))
    lpt = (txt @ img.t()) * sd["logit_scale"].exp()
    return {"logits_per_text": lpt, "logits_per_image": lpt.t(), "image_embeds": img, "text_embeds": txt}


def forward_loss_backward(sd, cfg, pixels, text, dtype=torch.float32):
    sdd = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    out = open_clip_forward(sdd, cfg, pixels.to(dtype), text)
    loss = O.clip_loss(out["logits_per_text"])
    loss.backward()
    grads = {k: (v.grad.detach() if v.grad is not None else None) for k, v in sdd.items()}
    return {k: v.detach() for k, v in out.items()}, loss.detach(), grads
