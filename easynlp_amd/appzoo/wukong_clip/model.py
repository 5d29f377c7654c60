"""MI355X-native drop-in for ``easynlp.appzoo.wukong_clip.model.WukongCLIP`` (easynlp/appzoo/wukong_clip/model.py:8-73).

``WukongModel`` (easynlp/modelzoo/models/wukong/modeling_wukong.py:238-433) is the CLIP architecture the library already
runs for the open_clip flavour -- ViT + causal text transformer of pre-LN residual attention blocks -- with

* every LayerNorm built with eps 1e-7 (modeling_wukong.py:242,248,285,289,330)  -> ``EZCLIP_OPT_BLOCK_LN_EPS``,
* the text feature taken at the token with id 102, ``x[(text == 102).nonzero()]`` (:349,359) -> ``EZCLIP_OPT_TEXT_EOT_ID``,
* its own names: ``config.json`` = ``{"model": {"visual": {...}, "text": {...}}}`` (ctor kwargs :268-275,311-318) and
  state-dict keys ``model.visual_encoder.*`` / ``model.text_encoder.*`` / ``model.logit_scale`` (:366-421),

and the application contract of wukong_clip/model.py:44-73: ``forward(inputs)`` returns the tuple
``({'image_features', 'text_features', 'logit_scale': exp(logit_scale)}, [])`` and ``compute_loss`` the symmetric
cross-entropy of ``logit_scale * I T^t`` / its transpose.  Parameters are ``nn.Parameter``s under the reference names;
every FLOP of the towers, the similarity and the loss runs in ``libezclip_hip.so`` (no CPU path).
"""
from __future__ import annotations

import json
import os
from typing import Dict

import torch
import torch.nn as nn

from ... import lib as L
from ..clip.model import CLIPApp, Config_Wrapper, HipClipEngine, _InfoNCEFn, _ParamTree, _SimilarityFn

WUKONG_LN_EPS = 1e-7          # modeling_wukong.py:242
WUKONG_TAIL_TOKEN = 102       # modeling_wukong.py:349  ([SEP] of the BERT vocabulary the Wukong tokenizer uses)


def library_config(raw: dict) -> dict:
    """``config.json`` of a Wukong checkpoint -> the library's config fields (include/ezclip.h)"""
    try:
        v, t = raw["model"]["visual"], raw["model"]["text"]
    except (KeyError, TypeError):
        raise L.EzclipError("WukongCLIP: config.json must hold {'model': {'visual': {...}, 'text': {...}}}")
    if int(v["output_dim"]) != int(t["output_dim"]):
        raise L.EzclipError("WukongCLIP: visual / text output_dim differ (%s vs %s)" % (v["output_dim"], t["output_dim"]))
    if int(v.get("heads", int(v["width"]) // 64)) * 64 != int(v["width"]) or int(t["heads"]) * 64 != int(t["width"]):
        raise L.EzclipError("WukongCLIP: heads must be width / 64 on the HIP path")
    T = int(t["width"])
    return dict(embed_dim=int(v["output_dim"]), image_resolution=int(v["input_resolution"]), vision_layers=int(v["layers"]),
                vision_width=int(v["width"]), vision_patch_size=int(v["patch_size"]), vocab_size=int(t["vocab_size"]),
                text_hidden_size=T, text_intermediate_size=4 * T, text_max_position_embeddings=int(t["context_length"]),
                text_num_attention_heads=int(t["heads"]), text_num_hidden_layers=int(t["layers"]), text_type_vocab_size=1)


def reference_name(lib_name: str) -> str:
    """library (= OPEN_CLIP) parameter name -> key under ``WukongCLIP.model`` (modeling_wukong.py:366-421)"""
    if lib_name.startswith("visual."):
        return "visual_encoder." + lib_name[len("visual."):]
    if lib_name == "logit_scale":
        return lib_name
    if lib_name == "token_embedding.weight":
        return "text_encoder.embedding_table"
    return "text_encoder." + lib_name


class WukongCLIP(CLIPApp):

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, user_defined_parameters={}, **kwargs):
        return cls(pretrained_model_name_or_path, user_defined_parameters)

    def __init__(self, pretrained_model_name_or_path=None, user_defined_parameters=None, **kwargs):
        super().__init__(None, user_defined_parameters, **kwargs)
        self.model_type = "wukong"
        if pretrained_model_name_or_path is None:
            return
        path = pretrained_model_name_or_path
        with open(os.path.join(path, "config.json"), "r") as f:
            self._build_wukong(json.load(f))
        ckpt = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(ckpt):
            state = torch.load(ckpt, map_location="cpu")
            # WukongModel.__init__ loads each tower strictly from the 'model.*' keys (:370-381,416-421)
            self.load_state_dict({k: v for k, v in state.items() if k.startswith("model.")}, strict=True)

    def _build_wukong(self, raw: dict) -> None:
        self.raw_config = raw
        self.config = Config_Wrapper(raw)                 # WukongConfig.to_json_string: the raw dict (configuration_wukong.py:29-38)
        ecfg = library_config(raw)
        eng = HipClipEngine(ecfg, self.compute_dtype, text_arch=1)
        eng.set_option(L.OPT_BLOCK_LN_EPS, WUKONG_LN_EPS)
        eng.set_option(L.OPT_TEXT_EOT_ID, WUKONG_TAIL_TOKEN)
        tree = _ParamTree()
        for n in eng.names:
            t = torch.zeros(eng.shapes[n], dtype=torch.float32)
            if n == "logit_scale":
                t.fill_(float(torch.log(torch.tensor(1.0 / 0.07))))
            tree.add(reference_name(n), t)
        self.model = tree
        self._engine = eng
        named = dict(tree.named_parameters())
        self._params: Dict[str, nn.Parameter] = {n: named[reference_name(n)] for n in eng.names}

    def _check_tail_tokens(self, input_ids: torch.Tensor) -> None:
        # x[(text == 102).nonzero()] yields one row per sample only if every row holds exactly one such token; the reference
        # would silently return a different number of rows otherwise -- here it is an error
        n = (input_ids == WUKONG_TAIL_TOKEN).sum(dim=1)
        if not bool((n == 1).all()):
            raise L.EzclipError("WukongCLIP: every row of input_ids must hold exactly one token %d" % WUKONG_TAIL_TOKEN)

    def forward(self, inputs):
        dev = self.logit_scale.device
        if inputs.get("pixel_values") is None and inputs.get("images") is not None:
            # batches of the drop-in WukongCLIPDataset: decoded uint8 images -> the reference's float32 pixel_values on the GPU
            R = int(inputs.get("image_size") or self._engine.cfg["image_resolution"])
            inputs["pixel_values"] = L.preprocess_images(inputs["images"], size=R, crop=R, device=dev)
        px = inputs["pixel_values"].to(dev) if inputs.get("pixel_values") is not None else None     # model.py:45-46
        ids = None
        if inputs.get("input_ids") is not None:                                                      # model.py:51-52
            self._check_tail_tokens(inputs["input_ids"])
            ids = inputs["input_ids"].to(dev)
        if px is None and ids is None:
            raise L.EzclipError("WukongCLIP.forward: neither 'pixel_values' nor 'input_ids'")
        image_features, text_features = self.encode(px, ids)
        return {"image_features": image_features, "text_features": text_features,
                "logit_scale": self.logit_scale.exp()}, []

    def compute_loss(self, forward_outputs, label_ids, **kwargs):
        img, txt = forward_outputs["image_features"], forward_outputs["text_features"]
        scale = forward_outputs["logit_scale"].mean()                                                # model.py:62
        # logits_per_text = scale * T I^t, logits_per_image its transpose; (CE rows + CE columns) / 2   (:64-71)
        logits_per_text = _SimilarityFn.apply(txt, img, scale.log())
        return {"loss": _InfoNCEFn.apply(logits_per_text)}

    def contrastive_step(self, pixel_values, input_ids, process_group=None, backward=False, **kw):
        self._check_tail_tokens(input_ids)
        return super().contrastive_step(pixel_values, input_ids, process_group=process_group, backward=backward, **kw)
