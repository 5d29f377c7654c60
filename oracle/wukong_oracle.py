"""CPU oracle for the reference ``WukongCLIP`` application (easynlp/appzoo/wukong_clip/model.py:8-73) =
``WukongModel`` (easynlp/modelzoo/models/wukong/modeling_wukong.py:238-433).  TEST INFRASTRUCTURE ONLY (see clip_oracle.py).

It is the open_clip architecture (open_clip_oracle.py) with three differences, all restated here:
* every LayerNorm is built with ``eps=1e-07`` (modeling_wukong.py:242,248,285,289,330);
* the text feature is the ``ln_final`` row of the token with id 102, ``x[(text == 102).nonzero(as_tuple=True)]``
  (:349,359) -- one [SEP] per row is the dataset's contract (wukong_clip/data.py) -- not ``argmax``;
* names: ``model.visual_encoder.*`` / ``model.text_encoder.{embedding_table, positional_embedding, transformer.*,
  ln_final.*, text_projection}`` / ``model.logit_scale`` (:366-421), config.json = ``{"model": {"visual": {...}, "text":
  {...}}}`` (VisualTransformer / TextTransformer ctor kwargs :268-275,311-318).
Application contract (wukong_clip/model.py:44-73): ``forward`` returns ``({'image_features', 'text_features',
'logit_scale': exp(logit_scale)}, [])``; ``compute_loss`` = (CE(s I T^t) + CE(s T I^t)) / 2.
Pinned against the real ``WukongCLIP`` by tools/make_golden.py (tests/golden/wukong_*.npz)."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import clip_oracle as O
from . import open_clip_oracle as OC

WUKONG_LN_EPS = 1e-7
WUKONG_TAIL_TOKEN = 102

WUKONG_CONFIGS: Dict[str, dict] = {
    "wk_tiny": {"model": {"visual": dict(input_resolution=64, patch_size=16, width=128, layers=2, heads=2, output_dim=64),
                          "text": dict(context_length=20, vocab_size=301, output_dim=64, width=192, layers=2, heads=3)}},
    # 26 vision tokens, the real context length (32), text narrower than vision as in Wukong ViT-L/14 (1024 / 768)
    "wk_small": {"model": {"visual": dict(input_resolution=80, patch_size=16, width=192, layers=3, heads=3, output_dim=128),
                           "text": dict(context_length=32, vocab_size=523, output_dim=128, width=128, layers=3, heads=2)}},
}


def open_clip_style_config(cfg: dict) -> dict:
    v, t = cfg["model"]["visual"], cfg["model"]["text"]
    assert v["output_dim"] == t["output_dim"]
    return dict(embed_dim=v["output_dim"], image_resolution=v["input_resolution"], vision_layers=v["layers"], vision_width=v["width"],
                vision_patch_size=v["patch_size"], context_length=t["context_length"], vocab_size=t["vocab_size"],
                transformer_width=t["width"], transformer_heads=t["heads"], transformer_layers=t["layers"],
                block_ln_eps=WUKONG_LN_EPS, eot_id=WUKONG_TAIL_TOKEN)


def to_open_clip_name(ref_name: str) -> str:
    """``model.*`` key of WukongCLIP.state_dict() -> the open_clip oracle's name"""
    n = ref_name[len("model."):]
    if n.startswith("visual_encoder."):
        return "visual." + n[len("visual_encoder."):]
    if n.startswith("text_encoder."):
        n = n[len("text_encoder."):]
        return {"embedding_table": "token_embedding.weight"}.get(n, n)
    return n        # logit_scale


def param_shapes(cfg: dict) -> Dict[str, tuple]:
    """reference names -> shapes"""
    oc = OC.param_shapes(open_clip_style_config(cfg))
    out = {}
    for n, shp in oc.items():
        if n.startswith("visual."):
            out["model.visual_encoder." + n[len("visual."):]] = shp
        elif n == "logit_scale":
            out["model.logit_scale"] = shp
        elif n == "token_embedding.weight":
            out["model.text_encoder.embedding_table"] = shp
        else:
            out["model.text_encoder." + n] = shp
    return out


def make_state_dict(cfg: dict, seed: int = 1234, small_embeddings: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded weights.  ``small_embeddings``: token / positional / class embeddings and the patch projection are scaled so
    that the first LayerNorm of each tower sees a variance of ~1e-6: eps 1e-7 vs the 1e-5 default then changes the
    features by tens of percent, i.e. the fixture discriminates the eps."""
    sd_oc = OC.make_state_dict(open_clip_style_config(cfg), seed)
    out = {}
    for n in param_shapes(cfg):
        v = sd_oc[to_open_clip_name(n)].clone()
        if small_embeddings and n.split(".")[-1] in ("embedding_table", "positional_embedding", "class_embedding") or \
                (small_embeddings and n.endswith("conv1.weight")):
            v = v * 0.01
        out[n] = v
    return out


def make_inputs(cfg: dict, batch: int, seed: int = 0):
    """pixels + WordPiece-style ids [B, context_length]: [CLS]=101, tokens, [SEP]=102 exactly once per row, zero padding."""
    rs = np.random.RandomState(seed)
    v, t = cfg["model"]["visual"], cfg["model"]["text"]
    R, C, V = v["input_resolution"], t["context_length"], t["vocab_size"]
    px = rs.standard_normal((batch, 3, R, R)).astype(np.float32)
    ids = np.zeros((batch, C), np.int64)
    lens = rs.randint(3, C + 1, size=(batch,))
    lens[0] = C
    if batch > 1:
        lens[1] = 2
    for b in range(batch):
        ids[b, 0] = 101
        body = rs.randint(103, V, size=(max(0, lens[b] - 2),))       # ids above [SEP]: argmax pooling would pick another row
        ids[b, 1:lens[b] - 1] = body
        ids[b, lens[b] - 1] = WUKONG_TAIL_TOKEN
    return torch.from_numpy(px), torch.from_numpy(ids)


def wukong_forward(sd: Dict[str, torch.Tensor], cfg: dict, pixels, text):
    """WukongCLIP.forward (wukong_clip/model.py:44-57) on reference-named tensors"""
    oc = open_clip_style_config(cfg)
    s = {to_open_clip_name(n): v for n, v in sd.items()}
    img = txt = None
    if pixels is not None:
        img = O.l2_normalize(O.vit_forward(s, OC.chinese_style_config(oc), pixels))       # :45-47
    if text is not None:
        txt = O.l2_normalize(OC.text_forward(s, oc, text))                                # :51-53
    return {"image_features": img, "text_features": txt, "logit_scale": s["logit_scale"].exp()}


def compute_loss(fo) -> torch.Tensor:
    """wukong_clip/model.py:59-73"""
    scale = fo["logit_scale"].mean()
    lpi = scale * fo["image_features"] @ fo["text_features"].t()
    lpt = scale * fo["text_features"] @ fo["image_features"].t()
    return (O.cross_entropy_diag(lpi) + O.cross_entropy_diag(lpt)) / 2


def forward_loss_backward(sd, cfg, pixels, text, dtype=torch.float32):
    sdd = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    fo = wukong_forward(sdd, cfg, pixels.to(dtype), text)
    loss = compute_loss(fo)
    loss.backward()
    grads = {k: (v.grad.detach() if v.grad is not None else None) for k, v in sdd.items()}
    return {k: v.detach() for k, v in fo.items()}, loss.detach(), grads
