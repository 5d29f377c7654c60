"""``huggingface_clip`` branch of the drop-in ``CLIPApp`` (reference: easynlp/appzoo/clip/model.py:73-104 construction,
:128-144 forward) -- the pai-clip-commercial-* checkpoints: ``RobertaModel`` text tower (pooled output) +
``CLIPVisionModel`` vision tower (pooled output, DETACHED: only the projections and the text tower train) + biased
``text_projection`` / ``vision_projection`` + ``logit_scale``.

The towers are the ones the library already runs for ``chinese_clip`` under other parameter names, so this module is
a NAME MAP plus four switches of the library (include/ezclip.h: pooler + tanh, frozen vision tower, text LayerNorm
eps, embedding padding index) and the per-token inputs RobertaModel takes (``ezclip_encode_text_ex``):

* parameters live as real ``nn.Parameter``s under the reference's names (``text_encoder.*``,
  ``vision_encoder.vision_model.*``, ``text_projection.{weight,bias}``, ``vision_projection.{weight,bias}``,
  ``logit_scale``): checkpoints load and save unchanged;
* most library parameters alias them directly (same storage); three kinds are DERIVED copies, rebuilt when their
  sources change: the packed ``in_proj_weight/bias = [q; k; v]`` of every vision layer and the two projection
  matrices transposed to the ``[width, embed_dim]`` layout of ``visual.proj`` / ``text_projection``;
* gradients come back under the library's names and are mapped to the reference parameters (the projections
  transposed back); the frozen vision tower gets none, exactly like ``vision_outputs[1].detach()``.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from ... import lib as L


def chinese_style_config(cfg: dict) -> dict:
    """CHINESE_CLIP-kwargs view of a huggingface_clip config.json (CLIPTextConfig / CLIPVisionConfig defaults:
    modelzoo/models/clip/configuration_clip.py:86-127,200-235)."""
    t, v = dict(cfg.get("text_config", {})), dict(cfg.get("vision_config", {}))
    W = int(v.get("hidden_size", 768))
    if int(v.get("intermediate_size", 4 * W)) != 4 * W:
        raise L.EzclipError("vision intermediate_size must be 4 * hidden_size on the HIP path")
    if v.get("hidden_act", "quick_gelu") != "quick_gelu" or t.get("hidden_act", "gelu") != "gelu":
        raise L.EzclipError("the HIP path implements quick_gelu (vision) / gelu (text) activations")
    if int(v.get("num_attention_heads", W // 64)) * 64 != W:
        raise L.EzclipError("vision head_dim must be 64")
    if abs(float(v.get("layer_norm_eps", 1e-5)) - 1e-5) > 1e-12:
        raise L.EzclipError("vision layer_norm_eps must be 1e-5")
    E = cfg.get("projection_dim")
    if E is None:
        raise L.EzclipError("config.json needs projection_dim (rows of text_projection.weight)")
    return dict(model_type="chinese_clip", embed_dim=int(E), image_resolution=int(v.get("image_size", 224)),
                vision_layers=int(v.get("num_hidden_layers", 12)), vision_width=W,
                vision_patch_size=int(v.get("patch_size", 32)), vocab_size=int(t.get("vocab_size", 21128)),
                text_hidden_size=int(t.get("hidden_size", 1024)), text_intermediate_size=int(t.get("intermediate_size", 4096)),
                text_max_position_embeddings=int(t.get("max_position_embeddings", 512)),
                text_num_attention_heads=int(t.get("num_attention_heads", 16)),
                text_num_hidden_layers=int(t.get("num_hidden_layers", 24)),
                text_type_vocab_size=int(t.get("type_vocab_size", 2)),
                text_hidden_dropout_prob=float(t.get("hidden_dropout_prob", 0.1)),
                text_attention_probs_dropout_prob=float(t.get("attention_probs_dropout_prob", 0.1)))


def reference_param_shapes(ccfg: dict) -> Dict[str, tuple]:
    """state_dict parameters of the reference CLIPApp in huggingface_clip mode."""
    H, F, W, E, P = (ccfg["text_hidden_size"], ccfg["text_intermediate_size"], ccfg["vision_width"], ccfg["embed_dim"],
                     ccfg["vision_patch_size"])
    Lv = (ccfg["image_resolution"] // P) ** 2 + 1
    s = {"text_encoder.embeddings.word_embeddings.weight": (ccfg["vocab_size"], H),
         "text_encoder.embeddings.position_embeddings.weight": (ccfg["text_max_position_embeddings"], H),
         "text_encoder.embeddings.token_type_embeddings.weight": (ccfg["text_type_vocab_size"], H),
         "text_encoder.embeddings.LayerNorm.weight": (H,), "text_encoder.embeddings.LayerNorm.bias": (H,)}
    for i in range(ccfg["text_num_hidden_layers"]):
        p = "text_encoder.encoder.layer.%d." % i
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
    for i in range(ccfg["vision_layers"]):
        p = vm + "encoder.layers.%d." % i
        for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (W, W), (W,)
        s[p + "layer_norm1.weight"], s[p + "layer_norm1.bias"] = (W,), (W,)
        s[p + "mlp.fc1.weight"], s[p + "mlp.fc1.bias"] = (4 * W, W), (4 * W,)
        s[p + "mlp.fc2.weight"], s[p + "mlp.fc2.bias"] = (W, 4 * W), (W,)
        s[p + "layer_norm2.weight"], s[p + "layer_norm2.bias"] = (W,), (W,)
    s[vm + "post_layernorm.weight"], s[vm + "post_layernorm.bias"] = (W,), (W,)
    s["text_projection.weight"], s["text_projection.bias"] = (E, H), (E,)
    s["vision_projection.weight"], s["vision_projection.bias"] = (E, W), (E,)
    s["logit_scale"] = (1,)
    return s


class NameMap:
    """library parameter name -> how it is obtained from the reference-named parameters."""

    def __init__(self, ccfg: dict):
        self.alias: Dict[str, str] = {}            # library name -> reference name (same storage)
        self.packed: Dict[str, List[str]] = {}     # library name -> reference names concatenated along dim 0
        self.transposed: Dict[str, str] = {}       # library name -> reference name, transposed
        a = self.alias
        vm = "vision_encoder.vision_model."
        a["visual.class_embedding"] = vm + "embeddings.class_embedding"
        a["visual.conv1.weight"] = vm + "embeddings.patch_embedding.weight"
        a["visual.positional_embedding"] = vm + "embeddings.position_embedding.weight"
        for lib, ref in (("ln_pre", "pre_layrnorm"), ("ln_post", "post_layernorm")):
            a["visual.%s.weight" % lib], a["visual.%s.bias" % lib] = vm + ref + ".weight", vm + ref + ".bias"
        for i in range(ccfg["vision_layers"]):
            s, d = vm + "encoder.layers.%d." % i, "visual.transformer.resblocks.%d." % i
            self.packed[d + "attn.in_proj_weight"] = [s + "self_attn.%s_proj.weight" % x for x in "qkv"]
            self.packed[d + "attn.in_proj_bias"] = [s + "self_attn.%s_proj.bias" % x for x in "qkv"]
            for lib, ref in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                             ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
                a[d + lib + ".weight"], a[d + lib + ".bias"] = s + ref + ".weight", s + ref + ".bias"
        self.transposed["visual.proj"] = "vision_projection.weight"
        a["visual.proj_bias"] = "vision_projection.bias"
        self.transposed["text_projection"] = "text_projection.weight"
        a["text_projection_bias"] = "text_projection.bias"
        a["logit_scale"] = "logit_scale"
        self.ccfg = ccfg

    def reference_name(self, lib_name: str):
        if lib_name.startswith("bert."):
            return "text_encoder." + lib_name[len("bert."):]
        return self.alias.get(lib_name)


class HFState:
    """Holds the reference-named parameters' view for the library and keeps the derived copies fresh."""

    def __init__(self, app, ccfg: dict):
        self.map = NameMap(ccfg)
        self.app = app
        self._derived: Dict[str, torch.Tensor] = {}
        self._sig = None

    def params(self) -> Dict[str, nn.Parameter]:
        return self.app._hf_params

    def library_tensors(self, names: List[str]) -> Dict[str, torch.Tensor]:
        """library name -> float32 tensor (alias or fresh derived copy), in the library's parameter order."""
        rp = self.params()
        srcs = [n for lst in self.map.packed.values() for n in lst] + list(self.map.transposed.values())
        sig = tuple((rp[n].data_ptr(), rp[n]._version) for n in srcs)
        # (a backward pass since the last pack: the reference's optimizers step through p.data, which bumps no version
        # counter -- core/optimizers.py:367,451,462 -- so the derived copies are rebuilt whenever the engine is dirty)
        if sig != self._sig or self.app._engine._weights_dirty:
            with torch.no_grad():
                fresh = {lib: torch.cat([rp[n].detach() for n in lst], dim=0) for lib, lst in self.map.packed.items()}
                fresh.update({lib: rp[ref].detach().t() for lib, ref in self.map.transposed.items()})
                for lib, v in fresh.items():
                    old = self._derived.get(lib)
                    if old is not None and old.shape == v.shape and old.device == v.device:
                        old.copy_(v)        # in place: same address, bumped version -> the engine re-packs its copies
                    else:
                        self._derived[lib] = v.contiguous().clone()
            self._sig = sig
        out = {}
        for n in names:
            if n in self._derived:
                out[n] = self._derived[n]
            elif n == "logit_scale":
                out[n] = rp["logit_scale"].detach().view(())        # [1] in the reference, a scalar in the library
            else:
                out[n] = rp[self.map.reference_name(n)]
        return out

    def trainable_library_names(self, names: List[str]) -> List[str]:
        """library parameters that receive gradients: the text tower and the two projections (+ biases)."""
        return [n for n in names if n.startswith("bert.") or n in ("text_projection", "text_projection_bias", "visual.proj",
                                                                 "visual.proj_bias")]

    def map_grads(self, lib_grads: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """gradients under the library's names -> under the reference names."""
        out = {}
        for n, g in lib_grads.items():
            if n in self.map.transposed:
                out[self.map.transposed[n]] = g.t().contiguous()
            else:
                ref = self.map.reference_name(n)
                if ref is not None:
                    out[ref] = g
        return out


def position_ids_from_input_ids(input_ids: torch.Tensor, pad: int) -> torch.Tensor:
    """create_position_ids_from_input_ids (roberta/modeling_roberta.py:1497-1510): integer index plumbing on the device."""
    mask = input_ids.ne(pad).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad


class HFEncodeFn(torch.autograd.Function):
    """(pixels, ids, token_type_ids, attention_mask, *reference parameters) -> (image_embeds, text_embeds)."""

    @staticmethod
    def forward(ctx, app, need_grad, pixels, ids, tt, am, *params):
        eng, st = app._engine, app._hf
        tensors = st.library_tensors(eng.names)
        eng.sync_params(tensors, with_backward=need_grad)
        img = txt = None
        ctx.ws_img = ctx.ws_txt = None
        from .model import _WsToken
        ctx.token = _WsToken() if need_grad else None
        pack = None
        if ids is not None:
            ids = ids.contiguous().long()
            tt, am = tt.contiguous().long(), am.contiguous().long()
            pos = position_ids_from_input_ids(ids, app._hf_pad_id)
            ctx.drop = app._next_dropout()
            eng.set_text_dropout(*ctx.drop)
            ctx.extras = (pos, tt, am)
            if eng.can_pack(need_grad) and ids.shape[1] >= 8:
                pack = eng.pack_meta(ids, am) or False        # (one launch; the scalars are read after the image tower is enqueued)
        if pixels is not None:
            pixels = pixels.contiguous().float()
            img, ctx.ws_img = eng.encode_image(pixels, need_grad, owner=ctx.token)
        if ids is not None:
            txt, ctx.ws_txt = eng.encode_text(ids, need_grad, extras=ctx.extras, owner=ctx.token, pack=pack)
            ctx.pack = eng.last_pack            # packed rows (bf16, no dropout): the backward runs on the same rows
        ctx.app, ctx.pixels, ctx.ids = app, pixels, ids
        ctx.has = (img is not None, txt is not None)
        dev = params[0].device
        outs = tuple(o if o is not None else torch.zeros(0, device=dev) for o in (img, txt))
        if img is None:
            ctx.mark_non_differentiable(outs[0])
        if txt is None:
            ctx.mark_non_differentiable(outs[1])
        return outs

    @staticmethod
    def backward(ctx, d_img, d_txt):
        app = ctx.app
        eng, st = app._engine, app._hf
        tensors = st.library_tensors(eng.names)
        dev = next(iter(tensors.values())).device
        want = st.trainable_library_names(eng.names)
        total = sum(tensors[n].numel() + (-tensors[n].numel()) % 4 for n in want)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grads, off = {}, 0
        for n in want:
            k = tensors[n].numel()
            grads[n] = flat[off:off + k].view(tensors[n].shape)
            off += k + (-k) % 4
        eng.sync_params(tensors, with_backward=True, grads=grads, refresh_if_dirty=False)
        if ctx.has[0]:
            eng.backward_image(ctx.pixels, d_img.contiguous(), ctx.ws_img)      # frozen tower: projection gradients only
        if ctx.has[1]:
            eng.set_text_dropout(*ctx.drop)
            eng.backward_text(ctx.ids, d_txt.contiguous(), ctx.ws_txt, extras=ctx.extras, pack=getattr(ctx, "pack", None))
        if ctx.token is not None:
            ctx.token.released = True
        by_ref = st.map_grads(grads)
        out = []
        for name in app._hf_param_order:
            p = app._hf_params[name]
            g = by_ref.get(name)
            out.append(g if (g is not None and p.requires_grad) else None)
        return (None, None, None, None, None, None) + tuple(out)
