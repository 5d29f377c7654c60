"""MI355X-native drop-in for ``easynlp.appzoo.clip.model.CLIPApp``.

Mirrors the reference application contract (easynlp/appzoo/clip/model.py:40-164,
base class easynlp/appzoo/application.py:26-38):

* ``CLIPApp(pretrained_model_name_or_path, user_defined_parameters=None)`` and
  ``CLIPApp.from_pretrained(path, user_defined_parameters={})`` read
  ``config.json`` (``model_type == "chinese_clip"``) + ``pytorch_model.bin``
  (keys prefixed ``chinese_clip.``), model.py:52-72;
* ``forward(inputs, feat=None)`` returns ``{'logits_per_text',
  'logits_per_image', 'image_embeds', 'text_embeds'}`` (or only the embeds with
  ``feat=True``), mutating ``inputs`` like the reference does (model.py:106-150);
* ``compute_loss(forward_outputs, label_ids) -> {'loss': tensor}`` (model.py:154-164);
* parameters are real ``nn.Parameter``s under the reference's names
  (``chinese_clip.visual.*``, ``chinese_clip.bert.*``, ``chinese_clip.text_projection``,
  ``chinese_clip.logit_scale``) so Trainer/AdamW/grad-clip/checkpoints work unchanged.

Python/PyTorch here is orchestration only: device memory, streams, autograd
plumbing, ``torch.distributed``.  Every FLOP of the dual-encoder, the similarity
and the InfoNCE loss (forward and backward) runs in ``libezclip_hip.so``.
There is no CPU fallback: without the HIP library / a GPU the model raises.
"""
from __future__ import annotations

import json
import os
import weakref
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from ... import lib as L
from ... import parallel as P
from ..application import Application

_CFG_FIELDS = ("embed_dim", "image_resolution", "vision_layers", "vision_width", "vision_patch_size",
               "vocab_size", "text_hidden_size", "text_intermediate_size", "text_max_position_embeddings",
               "text_num_attention_heads", "text_num_hidden_layers", "text_type_vocab_size")


class Config_Wrapper:
    """Same helper as the reference (model.py:32-38): Trainer calls
    ``model.config.to_json_string()`` and reads ``model.config.__dict__``."""

    def __init__(self, json_data):
        self.json_data = json_data

    def to_json_string(self):
        return json.dumps(self.json_data, ensure_ascii=False)


class RnEngineNames:
    """name predicates of the ModifiedResNet parameter tree (reference names, modeling_chineseclip.py:27-167)"""

    @staticmethod
    def is_norm(name: str) -> bool:
        """BatchNorm2d gains / biases (``bn1..3``, the ``downsample.1`` of a stage's first block)"""
        stem = name.rsplit(".", 1)[0]
        return stem.rsplit(".", 1)[-1].startswith("bn") or stem.endswith("downsample.1")


class _ParamTree(nn.Module):
    """A bare module tree that only holds parameters under dotted reference names."""

    def add(self, dotted: str, value: torch.Tensor, buffer: bool = False):
        head, _, rest = dotted.partition(".")
        if rest:
            if head not in self._modules:
                self.add_module(head, _ParamTree())
            self._modules[head].add(rest, value, buffer)
        elif buffer:
            self.register_buffer(head, value, persistent=True)
        else:
            self.register_parameter(head, nn.Parameter(value))


def _cfg_struct(cfg: dict, dtype_code: int) -> L.EzclipConfig:
    c = L.EzclipConfig()
    # a tuple of stage depths = ModifiedResNet (modeling_chineseclip.py:279-287): it runs behind its own handle
    # (rn_tower.RnEngine) and this one is created text-only (vision_layers = 0; the ViT fields are then not read)
    resnet = isinstance(cfg.get("vision_layers"), (list, tuple))
    for f in _CFG_FIELDS:
        if resnet and f in ("vision_layers", "vision_patch_size"):
            setattr(c, f, 0)
        else:
            setattr(c, f, int(cfg[f]))
    c.compute_dtype = dtype_code
    return c


class HipClipEngine:
    """Owns the C handle, the packed-weight shadow and the workspaces for one
    module instance on one device."""

    # parameters only the huggingface_clip branch has (include/ezclip.h); never created / bound for chinese_clip
    HF_ONLY_PARAMS = ("visual.proj_bias", "text_projection_bias")

    def __init__(self, cfg: dict, dtype_code: int, hf_branch: bool = False, text_arch: int = 0):
        self.lib = L.load()
        self.cfg = cfg
        self.dtype_code = dtype_code
        self._cstruct = _cfg_struct(cfg, dtype_code)
        h = L.C.c_void_p()
        L.check(self.lib.ezclip_create_ex(L.C.byref(self._cstruct), int(text_arch), L.C.byref(h)), "ezclip_create")
        self.handle = h
        self.names: List[str] = []
        self.shapes: Dict[str, tuple] = {}
        name = L.C.c_char_p()
        shape = (L.C.c_int64 * 8)()
        ndim = L.C.c_int()
        for i in range(self.lib.ezclip_num_params(h)):
            L.check(self.lib.ezclip_param_info(h, i, L.C.byref(name), shape, L.C.byref(ndim)), "param_info")
            n = name.value.decode()
            if n in self.HF_ONLY_PARAMS and not hf_branch:
                continue
            self.names.append(n)
            self.shapes[n] = tuple(int(shape[j]) for j in range(ndim.value))
        self._shadow = None
        self._shadow_backward = False
        self._bind_sig = None
        self._content_sig = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._ws_owner: Dict[tuple, "weakref.ref"] = {}      # save=True workspaces: the autograd ctx token that still needs them
        # Set by every backward pass: an optimizer step normally follows, and the reference's optimizers update through
        # ``p.data`` (easynlp/core/optimizers.py:367,451,462), which does NOT bump ``p._version`` -- the packed copies are
        # refreshed at the next forward whether or not the version counters moved.
        self._weights_dirty = False
        self._arenas: Dict[str, "P.GradArena"] = {}
        self._progress_cb = None
        self.text_arch = int(text_arch)
        self._drop = (0.0, 0.0)
        # inference runs the BERT tower on the unmasked tokens only (ezclip_encode_text_packed); EZCLIP_PACK_TEXT=0 /
        # clip_pack_text=0 feeds every padded position through it as the reference does -- same embeddings
        self.pack_text = os.environ.get("EZCLIP_PACK_TEXT", "1") not in ("0", "false", "False")
        # huggingface_clip batches (explicit position / type / mask tensors) packed while train-mode dropout is armed
        self.pack_hf_dropout = os.environ.get("EZCLIP_PACK_HF_DROPOUT", "1") not in ("0", "false", "False")
        # contrastive step through the tiled kernels (csrc/nce.hip); EZCLIP_NCE_TILED=0: the materialising path of rounds 1-2
        self.nce_tiled = os.environ.get("EZCLIP_NCE_TILED", "1") not in ("0", "false", "False")
        self.last_text_rows = None
        self.last_pack = None
        self._pack_cache = None
        self.uses_pooler = bool(hf_branch)
        self.embed_dim = int(cfg["embed_dim"])

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ezclip_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- parameter binding ---------------------------------------------------------------
    def sync_params(self, params: Dict[str, torch.Tensor], with_backward: bool,
                    grads: Optional[Dict[str, torch.Tensor]] = None, refresh_if_dirty: bool = True) -> None:
        """(Re)bind device pointers when tensors moved, and refresh the packed
        weights when the parameter *contents* may have changed: a version counter
        moved (``load_state_dict``, torch.optim steps), a backward pass ran since
        the last pack (``p.data`` updates of the reference's optimizers leave
        ``_version`` alone), or ``mark_weights_dirty()`` was called."""
        plist = [params[n] for n in self.names]
        ptrs = tuple(p.data_ptr() for p in plist)
        gsig = None if grads is None else tuple((grads[n].data_ptr() if grads.get(n) is not None else 0) for n in self.names)
        dev = plist[0].device
        rebound = False
        # (a call without `grads` -- a forward -- leaves the gradient buffers of the last backward bound: with the persistent
        # gradient arena the ~400 bind calls happen once, not twice per step)
        if self._bind_sig is None or ptrs != self._bind_sig[0] or (gsig is not None and gsig != self._bind_sig[1]):
            for n, p in zip(self.names, plist):
                if not p.is_cuda:
                    raise L.EzclipError("parameter %s is on %s: move the model to a GPU (no CPU path)" % (n, p.device))
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.EzclipError("parameter %s must be contiguous float32" % n)
                shp = (L.C.c_int64 * max(1, p.dim()))(*p.shape)
                g = grads.get(n) if grads is not None else None
                L.check(self.lib.ezclip_bind_param(self.handle, n.encode(), L.ptr(p), L.ptr(g), shp, p.dim()),
                        "bind_param(%s)" % n)
            rebound = self._bind_sig is None or ptrs != self._bind_sig[0]
            self._bind_sig = (ptrs, gsig)
        grow = self._shadow is None or (with_backward and not self._shadow_backward) or self._shadow.device != dev
        if grow:
            wb = with_backward or self._shadow_backward
            nbytes = self.lib.ezclip_shadow_bytes(self.handle, 1 if wb else 0)
            self._shadow = L.alloc_bytes(nbytes, dev)
            self._shadow_backward = wb
            L.check(self.lib.ezclip_set_shadow(self.handle, L.ptr(self._shadow), self._shadow.numel(), 1 if wb else 0),
                    "set_shadow")
        content_sig = tuple(p._version for p in plist)
        if grow or rebound or content_sig != self._content_sig or (self._weights_dirty and refresh_if_dirty):
            L.check(self.lib.ezclip_refresh_weights(self.handle, L.stream_ptr()), "refresh_weights")
            self._content_sig = content_sig
            self._weights_dirty = False

    def mark_weights_dirty(self) -> None:
        """The parameter values changed in a way torch cannot see (writes through ``p.data`` / raw pointers): re-pack the
        library's copies at the next forward."""
        self._weights_dirty = True

    def workspace(self, kind: str, batch: int, seq_len: int, save: bool, device, owner=None) -> torch.Tensor:
        """The activation workspace of one tower pass.  One buffer per (kind, save) is cached and reused -- except that a
        ``save=True`` buffer belongs to the forward that filled it until the matching backward has run (``owner``: the
        token the autograd ctx holds): a second grad-enabled forward in between (two micro-batches whose losses are
        summed, an extra feature call between forward and backward) gets a buffer of its own instead of overwriting
        the saved activations."""
        key = (kind, batch, seq_len, save, str(device))
        if kind.startswith("image"):
            nbytes = self.lib.ezclip_image_workspace_bytes(self.handle, batch, 1 if save else 0)
        else:
            nbytes = self.lib.ezclip_text_workspace_bytes(self.handle, batch, seq_len, 1 if save else 0)
        ws = self._ws.get(key)
        if save and ws is not None:
            prev = self._ws_owner.get(key)
            prev = prev() if prev is not None else None
            if prev is not None and prev is not owner and not prev.released:
                return L.alloc_bytes(nbytes, device)          # lives as long as the caller's ctx holds it
        if ws is None:
            # keep at most one cached workspace per (kind, save): batch shape changes are rare
            for k in [k for k in self._ws if k[0] == kind and k[3] == save]:
                del self._ws[k]
                self._ws_owner.pop(k, None)
            ws = L.alloc_bytes(nbytes, device)
            self._ws[key] = ws
        if save:
            if owner is not None:
                self._ws_owner[key] = weakref.ref(owner)
            else:
                self._ws_owner.pop(key, None)
        return ws

    # -- streams: the two towers are independent until the similarity ----------------------------------
    def side_stream(self, device, index: int = 1) -> "torch.cuda.Stream":
        """A second HIP stream for the text tower: the image tower's persistent GEMMs leave CUs idle in their last
        partial round of tiles (2364 tiles on 256 CUs = 9.23 rounds for the N = 768 products), which the other tower's
        kernels fill when both are in flight."""
        return pick_side_stream(device, index)          # (one per process and device, measured to run beside the current stream)

    # -- gradient arena / progress hook -------------------------------------------------------------------
    def grad_names(self) -> List[str]:
        """Parameters the backward pass writes: everything but the BertPooler of a chinese_clip model (computed by the
        reference's BertModel but unused, so its ``.grad`` stays None there too -- modeling_chineseclip.py:349-350)."""
        return [n for n in self.names if not (n.startswith("bert.pooler.") and not self.uses_pooler)]

    def grad_arena(self, which: str, device) -> "P.GradArena":
        """'step': contrastive_step's (its views ARE the parameters' ``.grad``); 'autograd': _EncodeFn.backward's (views
        are handed to autograd and reused once nothing references them any more)."""
        a = self._arenas.get(which)
        if a is None or a.flat.device != torch.device(device):
            a = P.GradArena(self.grad_names(), self.shapes, device, keep_views=(which == "step"))
            self._arenas[which] = a
        return a

    def progress_events(self, enable: bool) -> None:
        """(Re)arm or switch off the library's progress event log (include/ezclip.h: ezclip_backward_progress_events): the
        backward calls then record one event per finished parameter group instead of calling back into Python."""
        L.check(self.lib.ezclip_backward_progress_events(self.handle, 1 if enable else 0), "backward_progress_events")

    def drain_progress(self):
        """[(tower, stage, event handle)] logged by the backward calls since the last drain, in completion order."""
        cap = 2 * (len(self.names) // 4 + 8)
        cap = 256 if cap < 256 else cap
        tw, st, ev, n = (L.C.c_int * cap)(), (L.C.c_int * cap)(), (L.C.c_void_p * cap)(), L.C.c_int(0)
        L.check(self.lib.ezclip_backward_progress_drain(self.handle, tw, st, ev, cap, L.C.byref(n)), "backward_progress_drain")
        return [(int(tw[i]), int(st[i]), ev[i]) for i in range(n.value)]

    def set_progress_hook(self, fn) -> None:
        """fn(tower, stage) from inside ezclip_backward_* (include/ezclip.h: ezclip_set_backward_progress); None removes it."""
        if fn is None:
            self._progress_cb = None
            L.check(self.lib.ezclip_set_backward_progress(self.handle, None, None), "set_backward_progress")
            return
        self._progress_cb = L.PROGRESS_FN(lambda user, tower, stage: fn(int(tower), int(stage)))
        L.check(self.lib.ezclip_set_backward_progress(self.handle, L.C.cast(self._progress_cb, L.C.c_void_p), None),
                "set_backward_progress")

    # -- forward -----------------------------------------------------------------------------
    def encode_image(self, pixels: torch.Tensor, save: bool, owner=None, stream=None) -> (torch.Tensor, torch.Tensor):
        pixels = pixels.contiguous()
        if pixels.dtype != torch.float32:
            pixels = pixels.float()
        B = pixels.shape[0]
        R = int(self.cfg["image_resolution"])
        if tuple(pixels.shape[1:]) != (3, R, R):
            raise L.EzclipError("pixel_values must be [B,3,%d,%d], got %s" % (R, R, tuple(pixels.shape)))
        out = torch.empty((B, self.embed_dim), dtype=torch.float32, device=pixels.device)
        ws = self.workspace("image", B, 0, save, pixels.device, owner)
        L.check(self.lib.ezclip_encode_image(self.handle, L.ptr(pixels), B, L.ptr(out), L.ptr(ws), ws.numel(),
                                             1 if save else 0, L.stream_ptr(stream)), "encode_image")
        return out, ws

    # -- packed text batches ---------------------------------------------------------------------------------
    def pack_meta(self, ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, device=None, stream=None):
        """Which tokens of a [B, S] batch the text tower has to see (ezclip_encode_text_packed, include/ezclip.h): every
        unmasked token (mask = ids != 0, modeling_chineseclip.py:347, or the explicit attention mask), every CLS token, and
        whole sentences without any unmasked key.
        * HOST ids (the DataLoader hands ``forward`` CPU tensors): computed on the host, no GPU involved; cached per tensor object.
        * DEVICE ids: ONE kernel on the current stream (ezclip_pack_text_meta) fills rowmap / cu / lens; the three scalars the
          host needs to size the launches (rows, longest, prefix) arrive in pinned host memory and are read by
          ``resolve_pack`` -- which the callers invoke AFTER they have enqueued the image tower, so the host never waits on an
          idle device and no stream is synchronised.  Nothing is cached: every step pays the launch.
        Returns a dict (possibly still unresolved: key 'ticket'), or None when packing does not apply."""
        B, S = ids.shape
        if ids.is_cuda and S <= 512 and B <= 12288 and getattr(self, "handle", None):
            ids = ids.contiguous()
            am = attention_mask.contiguous() if attention_mask is not None else None
            rowmap = torch.empty(B * S, dtype=torch.int32, device=ids.device)
            cu = torch.empty(B, dtype=torch.int32, device=ids.device)
            lens = torch.empty(B, dtype=torch.int32, device=ids.device)
            ticket = L.C.c_int(0)
            L.check(self.lib.ezclip_pack_text_meta(self.handle, L.ptr(ids), L.ptr(am), B, S, L.ptr(rowmap), L.ptr(cu), L.ptr(lens),
                                                   L.C.byref(ticket), L.stream_ptr(stream)), "pack_text_meta")
            return {"rowmap": rowmap, "cu": cu, "lens": lens, "shape": (B, S), "ticket": int(ticket.value), "_keep": (ids, am)}
        # the same tensor OBJECT, unmodified (version counter), as last time: same answer.  (The weak reference is alive only
        # while that object -- and with it its memory -- is; a new tensor at a recycled address is another object.)
        dev_key = None if device is None else str(torch.device(device))
        c = self._pack_cache
        if (c is not None and c[0]() is ids and c[1] == ids._version and c[5] == dev_key
                and (attention_mask is None) == (c[2] is None)
                and (attention_mask is None or (c[2]() is attention_mask and c[3] == attention_mask._version))):
            return c[4]
        meta = self._pack_meta_uncached(ids, attention_mask, device)
        self._pack_cache = (weakref.ref(ids), ids._version, None if attention_mask is None else weakref.ref(attention_mask),
                            None if attention_mask is None else attention_mask._version, meta, dev_key)
        return meta

    def resolve_pack(self, pack):
        """Finish a packing started on the device: read (rows, longest, prefix) of its launch from the pinned result words
        (a polled flag, no stream synchronisation).  Returns the usable dict, or False when packing would not pay."""
        if not pack:
            return pack
        if pack.get("unusable"):                 # (resolved before, and found not worth packing)
            return False
        if "ticket" not in pack:
            return pack
        rows, longest, prefix = L.C.c_int(0), L.C.c_int(0), L.C.c_int(0)
        L.check(self.lib.ezclip_pack_text_meta_result(self.handle, pack.pop("ticket"), L.C.byref(rows), L.C.byref(longest),
                                                      L.C.byref(prefix)), "pack_text_meta_result")
        pack.pop("_keep", None)
        B, S = pack["shape"]
        pack.update(rows=int(rows.value), longest=int(longest.value), prefix=bool(prefix.value))
        if pack["longest"] > 256 or pack["rows"] > 0.9 * B * S:
            pack["unusable"] = True
            return False
        return pack

    def _pack_meta_uncached(self, ids, attention_mask, device):
        B, S = ids.shape
        keep = ids.ne(0) if attention_mask is None else attention_mask.ne(0)
        keep = keep | ~keep.any(1, keepdim=True)
        keep[:, 0] = True
        lens = keep.sum(1, dtype=torch.int32)
        cs = torch.cumsum(lens, 0, dtype=torch.int32)
        # are the kept tokens of every sentence a prefix?  (train-mode dropout numbers its masks by padded positions: only then
        # is a packed position the padded one -- `usable_with_dropout`)
        prefix = (keep == (torch.arange(S, device=keep.device)[None, :] < lens[:, None])).all().to(torch.int32)
        total, longest, prefix = (int(v) for v in torch.stack([cs[-1], lens.max(), prefix]).tolist())
        if longest > 256 or total > 0.9 * B * S:
            return None
        cu = cs - lens
        within = torch.cumsum(keep, 1, dtype=torch.int32) - 1
        dst = torch.where(keep, cu[:, None] + within, torch.full_like(within, B * S))      # dropped tokens -> a dummy slot
        rowmap = torch.empty(B * S + 1, dtype=torch.int32, device=ids.device)
        rowmap.scatter_(0, dst.flatten().long(), torch.arange(B * S, dtype=torch.int32, device=ids.device))
        dev = device if device is not None else ids.device
        return {"rowmap": rowmap[:total].contiguous().to(dev), "cu": cu.contiguous().to(dev), "lens": lens.contiguous().to(dev),
                "rows": total, "longest": longest, "shape": (B, S), "prefix": bool(prefix)}

    def can_pack(self, save: bool) -> bool:
        """bf16 BERT towers, forward with or without save_for_backward -- the backward then runs on the same packed rows.  With
        train-mode dropout armed only batches whose kept tokens are prefixes are packed (``usable``)."""
        return self.pack_text and self.dtype_code == L.DTYPE_BF16 and self.text_arch == 0

    def usable(self, pack, extras=None) -> bool:
        """May this packing (pack_meta) be used under the dropout state armed for the next call?  With dropout: kept tokens must
        be prefixes, and only the chinese_clip text branch (no explicit position / type / mask tensors): that combination is the
        one verified on hardware against the padded run; the huggingface_clip branch stays on padded rows while dropout is armed."""
        if not pack or pack.get("unusable"):
            return False
        if self._drop == (0.0, 0.0):
            return True
        return bool(pack.get("prefix")) and (extras is None or self.pack_hf_dropout)

    def encode_text(self, ids: torch.Tensor, save: bool, extras=None, owner=None, stream=None, pack=None) -> (torch.Tensor, torch.Tensor):
        """extras: (position_ids, token_type_ids, attention_mask) int64 [B, S] device tensors (huggingface_clip branch).
        pack: what ``pack_meta`` returned for these ids (computed here when None and packing applies)."""
        ids = ids.contiguous()
        if ids.dtype != torch.int64:
            ids = ids.long()
        B, S = ids.shape
        out = torch.empty((B, self.embed_dim), dtype=torch.float32, device=ids.device)
        ws = self.workspace("text", B, S, save, ids.device, owner)
        if self.can_pack(save) and S >= 8 and pack is not False:
            if pack is None:
                pack = self.pack_meta(ids, None if extras is None else extras[2], stream=stream)   # same stream as the tower that reads it
            elif pack.get("shape") != (B, S):
                raise L.EzclipError("packing metadata of another batch")
            pack = self.resolve_pack(pack)       # (device-built metadata: the scalars are read here, after the image tower was enqueued)
            if self.usable(pack, extras):
                pos, tt, am = extras if extras is not None else (None, None, None)
                L.check(self.lib.ezclip_encode_text_packed(self.handle, L.ptr(ids), L.ptr(pos), L.ptr(tt), L.ptr(am),
                                                           L.ptr(pack["rowmap"]), L.ptr(pack["cu"]), L.ptr(pack["lens"]), B, S,
                                                           pack["rows"], pack["longest"], L.ptr(out), L.ptr(ws), ws.numel(),
                                                           1 if save else 0, L.stream_ptr(stream)), "encode_text_packed")
                self.last_text_rows = (pack["rows"], B * S)
                self.last_pack = pack
                return out, ws
        self.last_text_rows = (B * S, B * S)
        self.last_pack = None
        if extras is None:
            L.check(self.lib.ezclip_encode_text(self.handle, L.ptr(ids), B, S, L.ptr(out), L.ptr(ws), ws.numel(),
                                                1 if save else 0, L.stream_ptr(stream)), "encode_text")
        else:
            pos, tt, am = extras
            L.check(self.lib.ezclip_encode_text_ex(self.handle, L.ptr(ids), L.ptr(pos), L.ptr(tt), L.ptr(am), B, S, L.ptr(out),
                                                   L.ptr(ws), ws.numel(), 1 if save else 0, L.stream_ptr(stream)), "encode_text_ex")
        return out, ws

    def set_text_dropout(self, hidden_p: float, attn_p: float, seed: int) -> None:
        """Arm (or, with zeros, disarm) the BERT train-mode dropout for the next encode_text / backward_text."""
        L.check(self.lib.ezclip_set_text_dropout(self.handle, float(hidden_p), float(attn_p), int(seed)),
                "set_text_dropout")
        self._drop = (float(hidden_p), float(attn_p))

    def backward_image(self, pixels, d_emb, ws, stream=None):
        self._weights_dirty = True
        L.check(self.lib.ezclip_backward_image(self.handle, L.ptr(pixels), pixels.shape[0], L.ptr(d_emb),
                                               L.ptr(ws), ws.numel(), L.stream_ptr(stream)), "backward_image")

    def backward_text(self, ids, d_emb, ws, extras=None, stream=None, pack=None):
        """pack: the packing metadata the matching forward ran with (encode_text), or None / False"""
        self._weights_dirty = True
        if pack:
            pos, tt, am = extras if extras is not None else (None, None, None)
            L.check(self.lib.ezclip_backward_text_packed(self.handle, L.ptr(ids), L.ptr(pos), L.ptr(tt), L.ptr(am),
                                                         L.ptr(pack["rowmap"]), L.ptr(pack["cu"]), L.ptr(pack["lens"]),
                                                         ids.shape[0], ids.shape[1], pack["rows"], pack["longest"], L.ptr(d_emb),
                                                         L.ptr(ws), ws.numel(), L.stream_ptr(stream)), "backward_text_packed")
            return
        if extras is None:
            L.check(self.lib.ezclip_backward_text(self.handle, L.ptr(ids), ids.shape[0], ids.shape[1],
                                                  L.ptr(d_emb), L.ptr(ws), ws.numel(), L.stream_ptr(stream)),
                    "backward_text")
        else:
            pos, tt, am = extras
            L.check(self.lib.ezclip_backward_text_ex(self.handle, L.ptr(ids), L.ptr(pos), L.ptr(tt), L.ptr(am), ids.shape[0],
                                                     ids.shape[1], L.ptr(d_emb), L.ptr(ws), ws.numel(),
                                                     L.stream_ptr(stream)), "backward_text_ex")

    def set_option(self, key: int, value: float) -> None:
        L.check(self.lib.ezclip_set_option(self.handle, int(key), float(value)), "set_option")


# ---- the text tower's stream ---------------------------------------------------------------------------------------------------------
# HIP maps streams onto a handful of hardware queues, round robin at creation (GPU_MAX_HW_QUEUES, 4 by default).  A side stream that lands
# on the queue of the stream the image tower runs on does not run BESIDE it: the two towers serialise and a training step reads 135.9
# instead of 130.8 ms -- exactly what `app.two_streams = False` reads.  Round 6 found this behind the "autograd step is 4 % slower as the
# 7th workload of a process" of round 5: every engine created its own stream, and every fourth creation aliased (3 of 8 instances in
# profiles/r6_autograd_side_stream_alias.log).  So: ONE side stream per (process, device, index), chosen by a measurement -- a candidate is
# kept only if a trivial kernel on it finishes while the main stream is still busy.
_SIDE_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}
_SIDE_REJECTED: list = []           # (kept referenced: their pool slots stay taken)


def _runs_beside(main: "torch.cuda.Stream", cand: "torch.cuda.Stream", device) -> bool:
    """Does work on ``cand`` execute while ``main`` is busy?  ~2 ms of elementwise passes over 128 MiB on ``main``, one small fill on
    ``cand``, events on both (a one-off measurement at first use: plumbing, no kernel of the compute path)."""
    with torch.cuda.device(device):
        a = torch.ones(32 << 20, dtype=torch.float32, device=device)
        flag = torch.empty(64, dtype=torch.float32, device=device)
        with torch.cuda.stream(cand):       # (the first launch on a new stream creates its hardware queue: milliseconds, not part of the question)
            flag.fill_(0.0)
        torch.cuda.synchronize(device)
        t0, main_done, side_done = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(main):
            t0.record(main)
            for _ in range(24):
                a.mul_(1.0001)
            main_done.record(main)
        with torch.cuda.stream(cand):
            flag.fill_(1.0)
            side_done.record(cand)
        torch.cuda.synchronize(device)
        return t0.elapsed_time(side_done) < 0.5 * t0.elapsed_time(main_done)


def pick_side_stream(device, index: int = 1) -> "torch.cuda.Stream":
    key = (str(torch.device(device)), index, int(torch.cuda.current_stream(device).cuda_stream))
    st = _SIDE_STREAMS.get(key)
    if st is None:
        main = torch.cuda.current_stream(device)
        prio = int(os.environ.get("EZCLIP_SIDE_STREAM_PRIORITY", "0"))          # (A/B switch: -1 = a high-priority stream for the text tower)
        for _ in range(8):
            st = torch.cuda.Stream(device=device, priority=prio)
            if os.environ.get("EZCLIP_SIDE_STREAM_NO_PROBE") or _runs_beside(main, st, device):
                break
            _SIDE_REJECTED.append(st)
        _SIDE_STREAMS[key] = st
    return st


class _WsToken:
    """Ownership of the save-for-backward workspaces of one forward (HipClipEngine.workspace)."""
    __slots__ = ("released", "__weakref__")

    def __init__(self):
        self.released = False


def _run_towers(eng, two_streams, run_image, run_text, image_first=False, prep_text=None):
    """Enqueue the image tower on the current stream and the text tower on the engine's side stream (``two_streams``) or
    both on the current stream.  Everything enqueued before is visible to both; on return the current stream has joined
    the side stream.  Tensors are allocated under the current stream in either case (the callables only pass an explicit
    stream handle to the library).  ``prep_text(stream)``: work the text tower needs first (the packing-metadata launch), put
    on the TEXT stream before the image tower is enqueued -- it then runs beside the image tower's first kernels, and its
    result words are in host memory long before ``run_text`` asks for them; what it returns is handed to ``run_text``."""
    if not two_streams or run_image is None or run_text is None:
        prep = prep_text(None) if (prep_text is not None and run_text is not None) else None
        a = run_image(None) if run_image is not None else None
        b = (run_text(None, prep) if prep_text is not None else run_text(None)) if run_text is not None else None
        return a, b
    main = torch.cuda.current_stream()
    side = eng.side_stream(main.device)
    side.wait_stream(main)
    prep = prep_text(side) if prep_text is not None else None
    text = (lambda: run_text(side, prep)) if prep_text is not None else (lambda: run_text(side))
    if image_first:                # (host-side enqueue order only: the backward pass reports the image tower's groups first;
                                   #  the forward enqueues the image tower before it reads device-built packing metadata)
        a = run_image(None)
        b = text()
    else:
        b = text()
        a = run_image(None)
    main.wait_stream(side)
    return a, b


class _EncodeFn(torch.autograd.Function):
    """(pixels, ids, *params) -> (image_embeds, text_embeds); backward runs the HIP backward kernels into a flat gradient
    arena (one memset, completion-ordered: parallel.GradArena) and hands autograd views of it."""

    @staticmethod
    def forward(ctx, app, need_grad, pack_hint, pixels, ids, *params):
        eng = app._engine   # (grad mode is always off inside Function.forward: the caller decides need_grad)
        named = dict(zip(eng.names, params))
        eng.sync_params(named, with_backward=need_grad)
        ctx.ws_img = ctx.ws_txt = None
        ctx.token = _WsToken() if need_grad else None
        run_i = run_t = None
        if pixels is not None:
            pixels = pixels.contiguous().float()
            run_i = lambda st: eng.encode_image(pixels, need_grad, owner=ctx.token, stream=st)
        if ids is not None:
            ids = ids.contiguous().long()
            ctx.drop = app._next_dropout()          # (hidden_p, attn_p, seed); zeros in eval mode
            eng.set_text_dropout(*ctx.drop)
            want_pack = pack_hint is None and eng.can_pack(need_grad) and ids.shape[1] >= 8
            # device ids: one launch on the text stream now, its scalars are read in encode_text (after the image tower is enqueued);
            # pack_hint: forward() computed the metadata on the host ids (False: do not pack)
            prep_t = lambda st: (eng.pack_meta(ids, stream=st) or False) if want_pack else pack_hint
            run_t = lambda st, pack: eng.encode_text(ids, need_grad, owner=ctx.token, stream=st, pack=pack)
        ri, rt = _run_towers(eng, app.two_streams, run_i, run_t, image_first=True, prep_text=prep_t if ids is not None else None)
        ctx.pack = eng.last_pack if ids is not None else None
        img, ctx.ws_img = ri if ri is not None else (None, None)
        txt, ctx.ws_txt = rt if rt is not None else (None, None)
        ctx.app, ctx.pixels, ctx.ids = app, pixels, ids
        ctx.n_params = len(params)
        ctx.has = (img is not None, txt is not None)
        outs = tuple(o if o is not None else torch.zeros(0, device=params[0].device) for o in (img, txt))
        if img is None:
            ctx.mark_non_differentiable(outs[0])
        if txt is None:
            ctx.mark_non_differentiable(outs[1])
        return outs

    @staticmethod
    def backward(ctx, d_img, d_txt):
        app = ctx.app
        eng = app._engine
        params = dict(zip(eng.names, [app._params[n] for n in eng.names]))
        dev = next(iter(params.values())).device
        arena = eng.grad_arena("autograd", dev)
        # The persistent arena is reused when nobody references the views handed out last time any more (the Trainer's
        # optimizer.zero_grad() dropped them: core/trainer.py:337).  Otherwise -- gradient accumulation, two forwards
        # feeding one backward -- this pass writes into a buffer of its own and autograd adds it to the live gradients.
        if not arena.lent():
            arena.zero()
            views = arena.make_views(arena.flat)
        else:
            _, views = arena.fresh()
            app._arena_fresh_backwards = getattr(app, "_arena_fresh_backwards", 0) + 1      # (diagnostics: tools/autograd_step_timeline.py)
        eng.sync_params(params, with_backward=True, grads=views, refresh_if_dirty=False)
        run_i = run_t = None
        if ctx.has[0]:
            d_img = d_img.contiguous()
            run_i = lambda st: eng.backward_image(ctx.pixels, d_img, ctx.ws_img, stream=st)
        if ctx.has[1]:
            d_txt = d_txt.contiguous()
            eng.set_text_dropout(*ctx.drop)         # the masks of the matching forward
            run_t = lambda st: eng.backward_text(ctx.ids, d_txt, ctx.ws_txt, stream=st, pack=ctx.pack)
        _run_towers(eng, app.two_streams, run_i, run_t, image_first=True)
        if ctx.token is not None:
            ctx.token.released = True
        out = []
        for n in eng.names:
            # None where the reference's autograd leaves None: frozen parameters, the unused BertPooler, logit_scale (its
            # gradient comes from the similarity), the tower that did not run
            tower = P.grad_group(n)[0]
            ran = tower < 2 and ctx.has[tower]
            out.append(views[n] if (n in views and ran and params[n].requires_grad) else None)
        return (None, None, None, None, None) + tuple(out)


class _RnEncodeFn(torch.autograd.Function):
    """ModifiedResNet image tower in training mode: ``RnEngine.encode_image_train`` / ``backward`` (csrc/resnet.hip); the gradient
    tensors are allocated here and WRITTEN by the library, autograd accumulates them into ``.grad``.  Every forward owns the
    workspace its activations are saved in until its backward has run (``_WsToken``, as ``_EncodeFn``): two training forwards
    before one backward each keep their own, and the library re-derives every pointer from the workspace it is handed."""

    @staticmethod
    def forward(ctx, app, pixels, names, *params):
        ctx.token = _WsToken()
        out, ctx.saved_ws = app._rn.encode_image_train(pixels, owner=ctx.token, return_saved=True)
        ctx.app, ctx.names = app, names
        ctx.need = [p.requires_grad for p in params]
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (out,) = ctx.saved_tensors
        rn = ctx.app._rn
        grads = {n: torch.empty(rn.shapes[n], dtype=torch.float32, device=out.device) for n in ctx.names}
        rn.backward(out, d_out, grads, saved=ctx.saved_ws)
        ctx.token.released = True
        return (None, None, None) + tuple(grads[n] if need else None for n, need in zip(ctx.names, ctx.need))


class _SimilarityFn(torch.autograd.Function):
    """logits_per_text = T @ I^T * exp(logit_scale)  (model.py:148) on the f32 MFMA GEMM."""

    @staticmethod
    def forward(ctx, txt, img, logit_scale):
        txt, img = txt.contiguous(), img.contiguous()
        out = L.similarity(txt, img, logit_scale)
        ctx.save_for_backward(txt, img, logit_scale, out)
        return out

    @staticmethod
    def backward(ctx, g):
        txt, img, ls, out = ctx.saved_tensors
        g = g.contiguous()
        # dT = s * g @ I ; dI = s * g^T @ T ; d ls = sum(g * out)   (NT GEMMs on transposed copies;
        # the contraction dim -- the batch -- is zero-padded to the f32 tile multiple of 32)
        pad = (-g.shape[1]) % 32
        padk = (lambda t: torch.nn.functional.pad(t, (0, pad)).contiguous()) if pad else (lambda t: t.contiguous())
        d_txt = L.similarity(padk(g), padk(img.t()), ls)
        pad = (-g.shape[0]) % 32
        padk = (lambda t: torch.nn.functional.pad(t, (0, pad)).contiguous()) if pad else (lambda t: t.contiguous())
        d_img = L.similarity(padk(g.t()), padk(txt.t()), ls)
        d_ls = (g * out).sum().reshape(ls.shape)
        return d_txt, d_img, d_ls


class _InfoNCEFn(torch.autograd.Function):
    """0.5 * (CE(S, arange) + CE(S^T, arange))  (model.py:154-160)."""

    @staticmethod
    def forward(ctx, logits):
        lib = L.load()
        logits = logits.contiguous()
        n = logits.shape[0]
        if logits.dim() != 2 or logits.shape[1] != n:
            raise L.EzclipError("compute_loss expects square logits_per_text")
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        scratch = torch.empty(4 * n, dtype=torch.float32, device=logits.device)
        L.check(lib.ezclip_infonce_from_logits(L.ptr(logits), n, L.ptr(loss), L.ptr(scratch), L.stream_ptr()),
                "infonce_from_logits")
        ctx.save_for_backward(logits)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        (logits,) = ctx.saved_tensors
        n = logits.shape[0]
        d = torch.empty_like(logits)
        scratch = torch.empty(4 * n, dtype=torch.float32, device=logits.device)
        g = g.contiguous().float()
        L.check(lib.ezclip_infonce_from_logits_bwd(L.ptr(logits), n, L.ptr(g), L.ptr(d), L.ptr(scratch),
                                                   L.stream_ptr()), "infonce_from_logits_bwd")
        return d


class _CrossEntropyDiagFn(torch.autograd.Function):
    """F.cross_entropy(logits, arange(len(logits)))  (CLIPApp.contrastive_loss, appzoo/clip/model.py:154-155): one direction."""

    @staticmethod
    def forward(ctx, logits):
        lib = L.load()
        if logits.dim() != 2 or logits.shape[0] > logits.shape[1]:
            raise L.EzclipError("contrastive_loss expects [rows, cols >= rows] logits (labels arange(rows)), got %s" % (tuple(logits.shape),))
        if not logits.is_cuda:
            raise L.EzclipError("contrastive_loss: logits must be on the GPU (no CPU fallback)")
        x = logits.float()
        if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
            x = x.contiguous()                       # (e.g. similarity.T: clip_loss itself never takes this route)
        rows, cols = x.shape
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        scratch = torch.empty(2 * rows, dtype=torch.float32, device=x.device)
        L.check(lib.ezclip_cross_entropy_diag(L.ptr(x), rows, cols, x.stride(0), L.ptr(loss), L.ptr(scratch), L.stream_ptr()),
                "cross_entropy_diag")
        ctx.save_for_backward(x, scratch)
        ctx.in_dtype = logits.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        x, scratch = ctx.saved_tensors
        rows, cols = x.shape
        d = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
        g = g.contiguous().float()
        L.check(lib.ezclip_cross_entropy_diag_bwd(L.ptr(x), rows, cols, x.stride(0), L.ptr(g), L.ptr(scratch), L.ptr(d), L.stream_ptr()),
                "cross_entropy_diag_bwd")
        return d.to(ctx.in_dtype)


def fused_infonce_shard(eng, txt_all, img_all, n, off, logit_scale, grad_scale, want_grads):
    """This rank's ``n`` rows (starting at ``off``) of both directions against all ``N`` columns:
    (loss = mean over the local rows, d_text [N, E], d_image [N, E], d_logit_scale) -- the gradients only when asked.
    ``ezclip_infonce_tiled`` (no [n, N] buffer; operands split into bf16 hi + lo on the f32 pipeline, plain bf16 on the bf16
    one) where the embedding width allows it, else ``ezclip_infonce_fused``'s materialising path."""
    lib = eng.lib
    N, e = img_all.shape
    # bf16 pipeline: always tiled -- split operands (float32-class loss and gradients, 0.09 ms at 1024 x 1024) while the block is
    # small, plain bf16 operands beyond 2^21 logits (0.22 instead of 0.47 ms at 1024 x 8192; the towers' own bf16 noise on the
    # embeddings is two orders above the rounding of a unit vector's components).  f32 pipeline: the exact-f32 materialising
    # path while the two logit blocks are small (its gradient bound of 1e-4 rel-L2 per parameter sits below what split-bf16
    # operands hold after the towers' backward: 2^-16 per product, amplified by the cancellation in sum_j (p_ij - d_ij) x_j);
    # tiled with split operands beyond 2^24 logits per block
    bf16 = eng.dtype_code == L.DTYPE_BF16
    want_tiled = eng.nce_tiled and (bf16 or n * N >= (1 << 24))
    split = 0 if (bf16 and n * N > (1 << 21)) else 1
    tiled_bytes = lib.ezclip_infonce_tiled_workspace_bytes(n, N, e) if want_tiled else 0
    key = ("nce", n, N, e, bool(tiled_bytes))
    ws = eng._ws.get(key)
    if ws is None:
        ws = L.alloc_bytes(tiled_bytes or lib.ezclip_infonce_workspace_bytes(n, N, e), img_all.device)
        eng._ws[key] = ws
    dev = img_all.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    d_txt = d_img = d_ls = None
    if want_grads:
        d_txt = torch.empty((N, e), dtype=torch.float32, device=dev)
        d_img = torch.empty((N, e), dtype=torch.float32, device=dev)
        d_ls = torch.empty((), dtype=torch.float32, device=dev)
    gs = float(grad_scale) if want_grads else 1.0
    if tiled_bytes:
        L.check(lib.ezclip_infonce_tiled(L.ptr(txt_all), L.ptr(img_all), n, N, off, e, L.ptr(logit_scale), gs,
                                         split, L.ptr(loss), L.ptr(d_txt), L.ptr(d_img),
                                         L.ptr(d_ls), L.ptr(ws), ws.numel(), L.stream_ptr()), "infonce_tiled")
    else:
        L.check(lib.ezclip_infonce_fused(L.ptr(txt_all), L.ptr(img_all), n, N, off, e, L.ptr(logit_scale), gs, L.ptr(loss),
                                         L.ptr(d_txt), L.ptr(d_img), L.ptr(d_ls), L.ptr(ws), ws.numel(), L.stream_ptr()),
                "infonce_fused")
    return loss, d_txt, d_img, d_ls


class _GlobalInfoNCEFn(torch.autograd.Function):
    """contrastive_scope='global' on the autograd path (what Trainer + DDP drive): all-gather the embeddings, this rank's
    rows of both directions against every column (row-local log-sum-exps: no second collective), reduce-scatter the
    embedding gradients back to their owners.  The loss is the mean over the LOCAL rows and the gradients are those of the
    SUM of the ranks' losses w.r.t. the local embeddings, so that DDP's gradient averaging yields the gradient of the global
    mean loss.  ``shard_fn`` computes the rank-local part (the HIP kernel; the CPU oracle in the gloo tests)."""

    @staticmethod
    def forward(ctx, shard_fn, group, txt, img, logit_scale):
        n = txt.shape[0]
        img_all, txt_all, off = P.gather_embeddings(img.detach().contiguous(), txt.detach().contiguous(), group)
        need = any(ctx.needs_input_grad[2:5])
        loss, d_txt, d_img, d_ls = shard_fn(txt_all, img_all, n, off, logit_scale.detach(), 1.0, need)
        ctx.need = need
        if need:
            d_img_l, d_txt_l = P.scatter_embedding_grads(d_img, d_txt, n, group)
            ctx.save_for_backward(d_txt_l, d_img_l, d_ls.reshape(logit_scale.shape))
        return loss

    @staticmethod
    def backward(ctx, g):
        d_txt, d_img, d_ls = ctx.saved_tensors
        return None, None, g * d_txt, g * d_img, g * d_ls


class CLIPApp(Application):

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, user_defined_parameters={}, **kwargs):
        return cls(pretrained_model_name_or_path, user_defined_parameters)

    def __init__(self, pretrained_model_name_or_path=None, user_defined_parameters=None, **kwargs):
        super().__init__()
        udp = user_defined_parameters or {}
        if isinstance(udp, dict) and "app_parameters" in udp:
            udp = dict(udp, **udp["app_parameters"])
        self.compute_dtype = L.dtype_code(kwargs.get("compute_dtype", udp.get("clip_compute_dtype", "bf16")))
        # 'local' = the reference's behaviour under DDP (each rank's own pairs, model.py:154-164);
        # 'global' = every rank's rows against the pairs of ALL ranks (RCCL all-gather of the embeddings)
        self.contrastive_scope = str(kwargs.get("contrastive_scope", udp.get("contrastive_scope", "local")))
        if self.contrastive_scope not in ("local", "global"):
            raise L.EzclipError("contrastive_scope must be 'local' or 'global', got %r" % self.contrastive_scope)
        # contrastive_scope='global' only: also return this rank's own [n, n] logits block in the training forward's dict (for
        # logging; DETACHED -- the gradient flows through `loss`, unlike the reference's differentiable logits_per_text).  It
        # costs one extra [n, n, E] f32 product per step that the loss does not use: user_defined_parameters
        # global_scope_logits=0 drops it (the dict then carries None under both logits keys).
        self.global_scope_logits = bool(int(kwargs.get("global_scope_logits", udp.get("global_scope_logits", 1))))
        self._engine = None
        self._params: Dict[str, nn.Parameter] = {}
        # image tower on the current stream, text tower on a second one (they meet at the similarity): on by default,
        # --user_defined_parameters 'clip_two_streams=0' or EZCLIP_TWO_STREAMS=0 runs them back to back
        self.two_streams = str(kwargs.get("two_streams", udp.get("clip_two_streams", os.environ.get("EZCLIP_TWO_STREAMS", "1")))) \
            not in ("0", "False", "false")
        self._pack_text_opt = kwargs.get("pack_text", udp.get("clip_pack_text"))
        # ModifiedResNet image tower: trained as the reference trains it (default since round 5: BatchNorm batch statistics and moving
        # running statistics in train() mode, gradients for visual.*: modeling_chineseclip.py:27-167 under core/trainer.py:658-661);
        # 'clip_rn_train=0' = a frozen, eval-mode tower (LiT-style tuning of the text tower; the behaviour of rounds 3-4)
        self.rn_train = str(kwargs.get("rn_train", udp.get("clip_rn_train", "1"))) not in ("0", "False", "false")
        if pretrained_model_name_or_path is None:
            return
        path = pretrained_model_name_or_path
        with open(os.path.join(path, "config.json"), "r") as f:
            self.raw_config = json.load(f)
        if self.raw_config.get("model_type") == "open_clip":
            # reference model.py:56-64: OPEN_CLIP(**config), checkpoint keys prefixed 'open_clip.'
            self.model_type = "open_clip"
            self.config = Config_Wrapper(self.raw_config)
            self._build(self.raw_config)
            ckpt = os.path.join(path, "pytorch_model.bin")
            if os.path.exists(ckpt):
                checkpoint = torch.load(ckpt, map_location="cpu")
                state = {k.replace("open_clip.", ""): v for k, v in checkpoint.items()}
                self.open_clip.load_state_dict(state)                                         # strict, as the reference (:64)
            return
        if self.raw_config.get("model_type") != "chinese_clip":
            # reference model.py:73: anything else is the huggingface_clip flavour (text_config / vision_config)
            self.model_type = "huggingface_clip"
            ckpt = os.path.join(path, "pytorch_model.bin")
            state = torch.load(ckpt, map_location="cpu") if os.path.exists(ckpt) else None        # model.py:78
            if state is not None and "text_projection.weight" in state:
                # the reference sizes the projections from the checkpoint tensors (model.py:93-96), not from config.json --
                # and a config.json written by its Trainer carries CLIPConfig's default projection_dim (512) whatever the model
                self.raw_config = dict(self.raw_config, projection_dim=int(state["text_projection.weight"].shape[0]))
            self._build_hf(self.raw_config)
            if state is not None:
                own = self.state_dict()
                # (a checkpoint without 'logit_scale' keeps the ln(1 / 0.07) initialisation, model.py:101-104)
                missing = [k for k in self._hf_params if k not in state and k != "logit_scale"]
                if missing:
                    raise L.EzclipError("checkpoint lacks %d parameters, e.g. %s" % (len(missing), missing[:3]))
                self.load_state_dict({k: v for k, v in state.items() if k in own}, strict=False)
            return
        self.model_type = "chinese_clip"
        self.config = Config_Wrapper(self.raw_config)
        self._build(self.raw_config)
        ckpt = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(ckpt):
            checkpoint = torch.load(ckpt, map_location="cpu")
            state = {k.replace("chinese_clip.", ""): v for k, v in checkpoint.items()}   # model.py:69-70
            self.chinese_clip.load_state_dict(state, strict=False)                          # model.py:72

    # ------------------------------------------------------------------------------------
    def _build(self, cfg: dict) -> None:
        open_clip = cfg.get("model_type") == "open_clip"
        if open_clip:
            # OPEN_CLIP ctor kwargs (modeling_openclip.py:256-271) -> the library's config fields (include/ezclip.h)
            if isinstance(cfg.get("vision_layers"), (list, tuple)):
                raise L.EzclipError("ModifiedResNet vision towers are not on the HIP path (SURVEY.md 8f)")
            T = int(cfg["transformer_width"])
            if int(cfg["transformer_heads"]) * 64 != T:
                raise L.EzclipError("open_clip: transformer_heads must be transformer_width / 64 on the HIP path")
            ecfg = dict(cfg, text_hidden_size=T, text_intermediate_size=4 * T,
                        text_max_position_embeddings=int(cfg["context_length"]),
                        text_num_attention_heads=int(cfg["transformer_heads"]),
                        text_num_hidden_layers=int(cfg["transformer_layers"]), text_type_vocab_size=1)
            eng = HipClipEngine(ecfg, self.compute_dtype, text_arch=1)
        else:
            eng = HipClipEngine(cfg, self.compute_dtype)
        tree = _ParamTree()
        self._rn = None
        if not open_clip and isinstance(cfg.get("vision_layers"), (list, tuple)):
            # ModifiedResNet image tower (frozen, eval-mode BatchNorm): reference-named parameters / statistics in the same tree,
            # in the reference module's order (visual.* first)
            from .rn_tower import RnEngine
            self._rn = RnEngine(cfg["vision_layers"], int(cfg["vision_width"]), int(cfg["embed_dim"]), int(cfg["image_resolution"]),
                                self.compute_dtype)
            for n in self._rn.names:
                t = torch.zeros(self._rn.shapes[n], dtype=torch.float32)
                if n.endswith("running_var") or (n.endswith(".weight") and (".bn" in n or "downsample.1" in n)):
                    t.fill_(1.0)                                        # nn.BatchNorm2d defaults
                tree.add(n, t, buffer=RnEngine.is_statistic(n))
                if n.endswith(".running_var"):                          # the step counter BatchNorm2d keeps in its state_dict
                    tree.add(n[:-len("running_var")] + "num_batches_tracked", torch.zeros((), dtype=torch.int64), buffer=True)
        for n in eng.names:
            shape = eng.shapes[n]
            t = torch.zeros(shape, dtype=torch.float32)
            if n == "logit_scale":
                t.fill_(float(torch.log(torch.tensor(1.0 / 0.07))))       # modeling_chineseclip.py:316
            tree.add(n, t)
        if open_clip:
            self.open_clip = tree
        else:
            # persistent buffer the reference BertEmbeddings carries in its state_dict (modeling_bert.py:88)
            tree.add("bert.embeddings.position_ids",
                     torch.arange(int(cfg["text_max_position_embeddings"])).expand((1, -1)).clone(), buffer=True)
            self.chinese_clip = tree
        self._engine = eng
        if self._pack_text_opt is not None:
            eng.pack_text = str(self._pack_text_opt) not in ("0", "False", "false")
        named = dict(tree.named_parameters())
        self._params = {n: named[n] for n in eng.names}
        if self._rn is not None and not getattr(self, "rn_train", False):
            for n in self._rn.names:                                    # 'clip_rn_train=0': frozen tower
                if n in named:
                    named[n].requires_grad_(False)

    def _encode_image_resnet(self, pixel_values):
        """ModifiedResNet tower: eval-mode BatchNorm, no gradient (rn_tower.py).  The tensors are looked up per call: ``.to()`` /
        ``.cuda()`` REPLACE a module's buffers (the BatchNorm statistics), so references taken at construction go stale."""
        both = dict(self.chinese_clip.named_parameters())
        both.update(self.chinese_clip.named_buffers())
        tensors = {n: both[n] for n in self._rn.names}
        if getattr(self, "rn_train", False) and self.training:
            # the reference module in train() mode: batch statistics, running statistics moved, autograd through the tower
            names = [n for n in self._rn.names if not self._rn.is_statistic(n)]
            self._rn.sync_train(tensors)
            out = _RnEncodeFn.apply(self, pixel_values, names, *[tensors[n] for n in names])
            with torch.no_grad():                                       # nn.BatchNorm2d.train(): one more batch tracked per forward
                torch._foreach_add_([b for n, b in both.items() if n.startswith("visual.") and n.endswith("num_batches_tracked")], 1)
            return out
        self._rn.sync(tensors)
        with torch.no_grad():
            return self._rn.encode_image(pixel_values)

    def _build_hf(self, raw: dict) -> None:
        """huggingface_clip branch (model.py:73-104): reference-named parameters + the name map of hf_branch.py."""
        from . import hf_branch as HB
        ccfg = HB.chinese_style_config(raw)
        tcfg = dict(raw.get("text_config", {}))
        # dropout probabilities of the text tower live in text_config there (CLIPTextConfig)
        self.raw_config = dict(raw, text_hidden_dropout_prob=ccfg["text_hidden_dropout_prob"],
                               text_attention_probs_dropout_prob=ccfg["text_attention_probs_dropout_prob"],
                               image_resolution=ccfg["image_resolution"])
        self.config = Config_Wrapper(raw)
        eng = HipClipEngine(ccfg, self.compute_dtype, hf_branch=True)
        eng.set_option(L.OPT_TEXT_POOLER, 1)
        eng.set_option(L.OPT_VISION_FROZEN, 1)
        eng.set_option(L.OPT_TEXT_LN_EPS, float(tcfg.get("layer_norm_eps", 1e-12)))
        self._hf_pad_id = int(tcfg.get("pad_token_id", 0))
        eng.set_option(L.OPT_TEXT_PAD_ID, self._hf_pad_id)
        tree_t, tree_v = _ParamTree(), _ParamTree()
        self.text_projection, self.vision_projection = _ParamTree(), _ParamTree()
        shapes = HB.reference_param_shapes(ccfg)
        for n, shp in shapes.items():
            t = torch.zeros(shp, dtype=torch.float32)
            if n == "logit_scale":
                self.logit_scale_param = nn.Parameter(t.fill_(float(torch.log(torch.tensor(1.0 / 0.07)))))   # model.py:104
            elif n.startswith("text_encoder."):
                tree_t.add(n[len("text_encoder."):], t)
            elif n.startswith("vision_encoder."):
                tree_v.add(n[len("vision_encoder."):], t)
            elif n.startswith("text_projection."):
                self.text_projection.add(n[len("text_projection."):], t)
            else:
                self.vision_projection.add(n[len("vision_projection."):], t)
        # persistent buffers of the reference modules' state_dict
        tree_t.add("embeddings.position_ids", torch.arange(ccfg["text_max_position_embeddings"]).expand((1, -1)).clone(), buffer=True)
        Lv = (ccfg["image_resolution"] // ccfg["vision_patch_size"]) ** 2 + 1
        tree_v.add("vision_model.embeddings.position_ids", torch.arange(Lv).expand((1, -1)).clone(), buffer=True)
        self.text_encoder, self.vision_encoder = tree_t, tree_v
        self._engine = eng
        if self._pack_text_opt is not None:
            eng.pack_text = str(self._pack_text_opt) not in ("0", "False", "false")
        named = dict(self.named_parameters())
        named["logit_scale"] = named.pop("logit_scale_param")
        self._hf_params = {n: named[n] for n in shapes}
        self._hf_param_order = list(shapes)
        self._hf = HB.HFState(self, ccfg)
        self._params = {"logit_scale": self._hf_params["logit_scale"], "text_projection": self._hf_params["text_projection.weight"]}

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        if getattr(self, "model_type", None) == "huggingface_clip":      # reference key: 'logit_scale' (model.py:102-104)
            prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
            sd = type(sd)((prefix + "logit_scale" if k == prefix + "logit_scale_param" else k, v) for k, v in sd.items())
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        if getattr(self, "model_type", None) == "huggingface_clip" and "logit_scale" in state_dict:
            state_dict = dict(state_dict)
            state_dict["logit_scale_param"] = state_dict.pop("logit_scale").reshape(1)
        return super().load_state_dict(state_dict, strict=strict, **kwargs)

    @classmethod
    def from_config(cls, config: dict, seed: int = 0, device="cuda", compute_dtype="bf16", **kwargs):
        """Random-init model of a given architecture directly on the device (benchmarks,
        smoke tests): reference-like init scales (VisualTransformer.__init__
        modeling_chineseclip.py:226-234, BertPreTrainedModel._init_weights modeling_bert.py:624-638)."""
        app = cls(None, compute_dtype=compute_dtype, **kwargs)
        app.raw_config = dict(config)
        app.model_type = "chinese_clip"
        app.config = Config_Wrapper(app.raw_config)
        app._build(app.raw_config)
        app.to(device)
        g = torch.Generator(device=device).manual_seed(seed)
        W, H = int(config["vision_width"]), int(config["text_hidden_size"])
        with torch.no_grad():
            for n, p in app._params.items():
                if n == "logit_scale":
                    continue
                if n.endswith("LayerNorm.weight") or (".ln_" in n and n.endswith(".weight")):
                    p.fill_(1.0)
                elif n.endswith(".bias") or n.endswith("in_proj_bias"):
                    p.zero_()
                else:
                    if n in ("visual.class_embedding", "visual.positional_embedding", "visual.proj"):
                        std = W ** -0.5
                    elif n == "text_projection":
                        std = H ** -0.5
                    elif n == "visual.conv1.weight":
                        std = (p.shape[1] * p.shape[2] * p.shape[3]) ** -0.5
                    elif n.startswith("visual."):
                        std = p.shape[-1] ** -0.5
                    else:
                        std = float(config.get("text_initializer_range", 0.02))
                    p.copy_(torch.randn(p.shape, generator=g, device=device) * std)
            if app._rn is not None:
                # ModifiedResNet tower (CHINESE_CLIP.initialize_parameters, modeling_chineseclip.py:323-334): attnpool projections
                # N(0, in_features^-0.5), every bn3 gain zero; convolutions drawn uniform(-fan_in^-0.5, fan_in^-0.5) -- nn.Conv2d's default
                # kaiming_uniform(a = sqrt 5), variance 1 / (3 fan_in) -- attnpool.positional_embedding randn / sqrt(embed) (:62),
                # BatchNorm gains / statistics at their defaults (_build).  Deviation kept: the attnpool BIASES are zero here, the
                # reference leaves nn.Linear's uniform(-in^-0.5, in^-0.5) default.  from_config only (bench / smoke): a checkpoint
                # load overwrites all of it.
                both = dict(app.chinese_clip.named_parameters())
                attn_std = None
                for n in app._rn.names:
                    if n == "visual.attnpool.c_proj.weight":
                        attn_std = app._rn.shapes[n][1] ** -0.5
                for n in app._rn.names:
                    if n not in both or RnEngineNames.is_norm(n):
                        if n in both and n.endswith("bn3.weight") and ".layer" in n:
                            both[n].zero_()
                        continue
                    p = both[n]
                    if n.endswith(".bias"):
                        p.zero_()
                    elif n.startswith("visual.attnpool.") and n.endswith("_proj.weight"):
                        p.copy_(torch.randn(p.shape, generator=g, device=device) * attn_std)
                    elif n == "visual.attnpool.positional_embedding":
                        p.copy_(torch.randn(p.shape, generator=g, device=device) * (p.shape[1] ** -0.5))
                    elif p.dim() == 4:
                        bound = (p.shape[1] * p.shape[2] * p.shape[3]) ** -0.5
                        p.copy_((torch.rand(p.shape, generator=g, device=device) * 2.0 - 1.0) * bound)
                    else:
                        p.copy_(torch.randn(p.shape, generator=g, device=device) * (p.shape[-1] ** -0.5))
                app._rn.mark_dirty()
        return app

    @classmethod
    def from_hf_config(cls, config: dict, seed: int = 0, device="cuda", compute_dtype="bf16"):
        """Random-init huggingface_clip-flavoured model (config.json schema: text_config / vision_config /
        projection_dim) directly on the device -- benchmarks and smoke tests."""
        app = cls(None, compute_dtype=compute_dtype)
        app.model_type = "huggingface_clip"
        app._build_hf(dict(config))
        app.to(device)
        g = torch.Generator(device=device).manual_seed(seed)
        with torch.no_grad():
            for n, p in app._hf_params.items():
                if n == "logit_scale":
                    continue
                if n.endswith("LayerNorm.weight") or n.endswith("layrnorm.weight") or "layer_norm" in n and n.endswith(".weight") \
                        or n.endswith("post_layernorm.weight"):
                    p.fill_(1.0)
                elif n.endswith(".bias"):
                    p.zero_()
                elif p.dim() == 1:
                    p.copy_(torch.randn(p.shape, generator=g, device=device) * p.shape[0] ** -0.5)
                else:
                    fan_in = p[0].numel()
                    std = 0.02 if n.startswith("text_encoder.") else fan_in ** -0.5
                    p.copy_(torch.randn(p.shape, generator=g, device=device) * std)
        return app

    # ------------------------------------------------------------------------------------
    def contrastive_step(self, pixel_values, input_ids, process_group=None, backward=False, token_type_ids=None,
                         attention_mask=None, zero_grad=False, reduce_gradients=False, bucket_bytes=64 << 20,
                         bucket_dtype=None):
        """Fast path without autograd bookkeeping: dual-encoder forward + InfoNCE
        (+ full backward into ``.grad`` when ``backward=True``), one C call per stage.

        With a ``process_group`` the contrastive batch is the *global* one
        (SURVEY.md 8e): image/text embeddings are all-gathered over RCCL, each rank
        evaluates its own rows of both directions (row-local LSE, no second
        collective in the forward), and the embedding gradients are summed back
        with a reduce-scatter.  Returns the (rank-local mean) loss tensor.

        ``zero_grad``: clear the gradients first (one memset of the flat gradient arena the ``.grad`` tensors are views
        of) instead of accumulating.  ``reduce_gradients``: sum the parameter gradients over the ranks, bucket by bucket
        WHILE the backward pass runs (parallel.OverlappedGradReducer driven by ezclip_set_backward_progress) -- what
        DistributedDataParallel does for the reference (core/trainer.py:101-108).  The embedding gradients are those of
        the global MEAN loss, so the sum is the gradient of that loss.
        """
        import torch.distributed as dist
        eng = self._engine
        lib = eng.lib
        if getattr(self, "_rn", None) is not None:
            raise L.EzclipError("contrastive_step drives the ViT image tower; a ModifiedResNet model goes through forward() / "
                                "compute_loss() / backward() (the autograd path the reference Trainer drives)")
        hf = getattr(self, "model_type", None) == "huggingface_clip"
        extras, transposed = None, []
        pixel_values = pixel_values.contiguous()
        input_ids = input_ids.contiguous()
        if pixel_values.dtype != torch.float32:
            pixel_values = pixel_values.float()
        if input_ids.dtype != torch.int64:
            input_ids = input_ids.long()
        world, rank, pg = 1, 0, None
        if process_group is not False and dist.is_available() and dist.is_initialized():
            pg = None if process_group in (None, True) else process_group
            world, rank = dist.get_world_size(pg), dist.get_rank(pg)
        if hf:
            # huggingface_clip: the library's parameters are views / derived copies of the reference-named ones
            # (hf_branch.py); gradients land in the reference parameters' .grad (projections through a transposed scratch)
            from . import hf_branch as HB
            st = self._hf
            params = st.library_tensors(eng.names)
            pad = self._hf_pad_id
            am = input_ids.ne(pad).long() if attention_mask is None else attention_mask.to(input_ids.device).long()
            tt = torch.zeros_like(input_ids) if token_type_ids is None else token_type_ids.to(input_ids.device).long()
            extras = (HB.position_ids_from_input_ids(input_ids, pad), tt.contiguous(), am.contiguous())
        else:
            params = self._params
        reducer = None
        if backward:
            grads = {}
            if hf:
                if self.logit_scale.grad is None:
                    self.logit_scale.grad = torch.zeros_like(self.logit_scale)
                for n in st.trainable_library_names(eng.names):
                    if n in st.map.transposed:
                        p = self._hf_params[st.map.transposed[n]]
                        scratch = torch.zeros(params[n].shape, dtype=torch.float32, device=p.device)
                        grads[n] = scratch
                        transposed.append((p, scratch))
                    else:
                        p = self._hf_params[st.map.reference_name(n)]
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                        elif zero_grad:
                            p.grad.zero_()
                        grads[n] = p.grad
                if zero_grad:
                    self.logit_scale.grad.zero_()
            else:
                # .grad tensors are views of ONE flat arena in completion order (parallel.GradArena): zeroing is one
                # memset, an all-reduce bucket is one slice
                arena = eng.grad_arena("step", pixel_values.device)
                mine = all(p.grad is None or arena.owns(n, p.grad) for n, p in params.items() if n in arena.views)
                fresh = [n for n in arena.views if params[n].grad is None]
                if mine and (zero_grad or len(fresh) == len(arena.views)):
                    arena.zero()
                else:
                    for n in arena.views:
                        g = params[n].grad
                        if g is None:
                            arena.views[n].zero_()
                        elif zero_grad:
                            g.zero_()
                for n in fresh:
                    params[n].grad = arena.views[n]
                grads = {n: params[n].grad for n in arena.views}
                if reduce_gradients and (world > 1 or reduce_gradients == "force"):      # ("force": single-rank hardware tests)
                    if not all(arena.owns(n, g) for n, g in grads.items()):
                        raise L.EzclipError("reduce_gradients needs the gradients in the engine's arena: drop foreign "
                                            ".grad tensors first (optimizer.zero_grad(set_to_none=True))")
                    reducer = P.OverlappedGradReducer(arena, pg, bucket_bytes, bucket_dtype=bucket_dtype)
            eng.sync_params(params, with_backward=True, grads=grads)
        else:
            eng.sync_params(params, with_backward=False)
        drop = self._next_dropout()
        eng.set_text_dropout(*drop)
        want_pack = eng.can_pack(backward) and input_ids.shape[1] >= 8
        eng.last_pack = None
        # (device ids: the packing metadata is one launch on the text stream; its scalars are read in encode_text, after the image
        #  tower has been enqueued)
        (img, ws_i), (txt, ws_t) = _run_towers(
            eng, self.two_streams,
            lambda s_: eng.encode_image(pixel_values, backward, stream=s_),
            lambda s_, pack: eng.encode_text(input_ids, backward, extras=extras, stream=s_, pack=pack), image_first=True,
            prep_text=lambda s_: (eng.pack_meta(input_ids, None if extras is None else extras[2], stream=s_) or False) if want_pack else False)
        n = img.shape[0]
        e = img.shape[1]
        if world > 1:
            img_all, txt_all, _ = P.gather_embeddings(img, txt, pg)     # one RCCL all-gather for both towers
        else:
            img_all, txt_all = img, txt
        N = world * n
        loss, d_txt, d_img, d_ls = fused_infonce_shard(eng, txt_all, img_all, n, rank * n, self.logit_scale, 1.0 / world, backward)
        if not backward:
            return loss
        if world > 1:
            d_img_l, d_txt_l = P.scatter_embedding_grads(d_img, d_txt, n, pg)   # one RCCL reduce-scatter
        else:
            d_img_l, d_txt_l = d_img, d_txt
        self.logit_scale.grad.add_(d_ls.reshape(self.logit_scale.grad.shape))
        two = self.two_streams
        main = torch.cuda.current_stream()
        side = eng.side_stream(main.device) if two else None
        if reducer is not None:
            # The library logs one event per finished parameter group on the stream that produced it (no Python runs inside
            # ezclip_backward_*: round 4); each tower's log is drained as soon as its backward CALL has returned -- the host is then
            # tens of milliseconds ahead of the device -- and every bucket of the all-reduce is ordered behind the events of its groups.
            reducer.notify(2, P.STAGE_HEAD)       # logit_scale (already final: written by the contrastive step)
            eng.progress_events(True)
            lib = eng.lib

            def drain():
                for tower, stage, ev in eng.drain_progress():
                    reducer.notify(tower, stage, wait=lambda ev=ev: L.check(lib.ezclip_stream_wait_event(L.stream_ptr(), ev), "stream_wait_event"))
        else:
            def drain():
                pass

        def bwd_image(s_):
            eng.backward_image(pixel_values, d_img_l, ws_i, stream=s_)
            drain()

        def bwd_text(s_):
            eng.backward_text(input_ids, d_txt_l, ws_t, extras=extras, stream=s_, pack=eng.last_pack)
            drain()
        try:
            _run_towers(eng, two, bwd_image, bwd_text, image_first=True)
        finally:
            if reducer is not None:
                eng.progress_events(False)
        if reducer is not None:
            reducer.finish()
            self.last_grad_buckets = list(reducer.buckets)
        for p, scratch in transposed:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            elif zero_grad:
                p.grad.zero_()
            p.grad.add_(scratch.t())
        return loss

    def _next_dropout(self):
        """(hidden_p, attention_p, seed) of the next text-tower pass.  Like the reference, dropout is active only in
        train mode (nn.Dropout in BertEmbeddings / BertSelfAttention / BertSelfOutput / BertOutput,
        modeling_bert.py:128,238,266,344) with the probabilities of config.json; the seed of each pass is drawn from
        torch's global CPU generator, so ``torch.manual_seed`` makes a run reproducible.  ``self.dropout_seed`` (an
        int) pins the seed instead (tests)."""
        cfg = getattr(self, "raw_config", None) or {}
        hp = float(cfg.get("text_hidden_dropout_prob", 0.0) or 0.0)
        ap = float(cfg.get("text_attention_probs_dropout_prob", 0.0) or 0.0)
        if not self.training or (hp == 0.0 and ap == 0.0):
            return (0.0, 0.0, 0)
        seed = getattr(self, "dropout_seed", None)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        return (hp, ap, int(seed))

    @property
    def logit_scale(self):
        return self._params["logit_scale"]

    def _plist(self):
        return [self._params[n] for n in self._engine.names]

    # ------------------------------------------------------------------------------------
    def encode(self, pixel_values=None, input_ids=None, token_type_ids=None, attention_mask=None, pack_hint=None):
        """pack_hint: packing metadata of ``input_ids`` computed from their host copy (``forward``), False = do not pack, None =
        decide here."""
        if getattr(self, "model_type", None) == "huggingface_clip":
            from .hf_branch import HFEncodeFn
            plist = [self._hf_params[n] for n in self._hf_param_order]
            need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in plist)
            img, txt = HFEncodeFn.apply(self, need_grad, pixel_values, input_ids, token_type_ids, attention_mask, *plist)
            return (img if pixel_values is not None else None), (txt if input_ids is not None else None)
        plist = self._plist()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in plist)
        if getattr(self, "_rn", None) is not None:
            img = self._encode_image_resnet(pixel_values) if pixel_values is not None else None
            txt = None
            if input_ids is not None:
                _, txt = _EncodeFn.apply(self, need_grad, pack_hint, None, input_ids, *plist)
            return img, txt
        img, txt = _EncodeFn.apply(self, need_grad, pack_hint, pixel_values, input_ids, *plist)
        return (img if pixel_values is not None else None), (txt if input_ids is not None else None)

    def forward(self, inputs, feat=None):
        _device = self._params["text_projection"].device
        if inputs.get("pixel_values") is None and inputs.get("images") is not None:
            # batches of the drop-in CLIPDataset (gpu_preprocess): decoded uint8 images -> the reference's float32
            # pixel_values (data.py:256-262) on the GPU, bit for bit
            R = int(inputs.get("image_size") or self._engine.cfg["image_resolution"])
            inputs["pixel_values"] = L.preprocess_images(inputs["images"], size=R, crop=R, device=_device)
        if "pixel_values" in inputs and inputs["pixel_values"] is not None:
            inputs["pixel_values"] = inputs["pixel_values"].to(_device)
        else:
            inputs["pixel_values"] = None
        if "input_ids" in inputs and inputs["input_ids"] is not None:
            ids_in = inputs["input_ids"]
            pack_hint = None
            if (not ids_in.is_cuda and ids_in.dim() == 2 and ids_in.shape[1] >= 8 and self._engine is not None
                    and self._engine.pack_text and getattr(self, "model_type", None) == "chinese_clip"):
                # which tokens the text tower has to see, from the host copy the DataLoader delivered (no device sync later);
                # handed to encode() explicitly -- it belongs to THIS call's ids
                pack_hint = self._engine.pack_meta(ids_in, device=_device) or False
            inputs["input_ids"] = ids_in.to(_device)
        else:
            inputs["input_ids"] = None
            pack_hint = None
        assert inputs["pixel_values"] is not None or inputs["input_ids"] is not None, \
            "text and image cannot both be None!"
        if getattr(self, "model_type", None) == "huggingface_clip" and inputs["input_ids"] is not None:
            # RobertaModel(input_ids, token_type_ids, attention_mask): the reference requires both keys (model.py:131-133)
            image_embeds, text_embeds = self.encode(inputs["pixel_values"], inputs["input_ids"],
                                                    inputs["token_type_ids"].to(_device), inputs["attention_mask"].to(_device))
        else:
            kw = {} if pack_hint is None else {"pack_hint": pack_hint}
            image_embeds, text_embeds = self.encode(inputs["pixel_values"], inputs["input_ids"], **kw)
        if feat is True:
            return {"image_embeds": image_embeds, "text_embeds": text_embeds}
        if self.contrastive_scope == "global" and self.training:
            # the [n, N] logits of the global batch stay inside the fused kernel's workspace: the dict carries the loss.  What a caller
            # can still LOG is this rank's own [n, n] block -- the tensor the reference returns under DDP (model.py:148) -- detached:
            # the gradient flows through `loss` (round 4; rounds 2-3 returned None here, which broke callers that read the logits)
            eng = self._engine
            loss = _GlobalInfoNCEFn.apply(lambda *a: fused_infonce_shard(eng, *a), None, text_embeds, image_embeds, self.logit_scale)
            local_logits = None
            if self.global_scope_logits:
                with torch.no_grad():
                    local_logits = _SimilarityFn.apply(text_embeds.detach(), image_embeds.detach(), self.logit_scale.detach())
            return {"loss": loss, "logits_per_text": local_logits, "logits_per_image": None if local_logits is None else local_logits.T,
                    "image_embeds": image_embeds, "text_embeds": text_embeds}
        logits_per_text = _SimilarityFn.apply(text_embeds, image_embeds, self.logit_scale)
        logits_per_image = logits_per_text.T
        return {"logits_per_text": logits_per_text, "logits_per_image": logits_per_image,
                "image_embeds": image_embeds, "text_embeds": text_embeds}

    def contrastive_loss(self, logits: torch.Tensor) -> torch.Tensor:
        """One direction: ``F.cross_entropy(logits, arange(len(logits)))`` (reference model.py:154-155), autograd-capable.
        ``clip_loss`` does not call it twice: both directions of a square block are one fused HIP call there.
        GPU logits only (``EzclipError`` otherwise): the reference's version runs on any device, but this package has NO CPU compute
        path by construction -- a torch fallback here would be one (ADVICE r4 suggested it; declined for that reason)."""
        return _CrossEntropyDiagFn.apply(logits)

    def clip_loss(self, similarity: torch.Tensor) -> torch.Tensor:
        return _InfoNCEFn.apply(similarity)

    def compute_loss(self, forward_outputs, label_ids, **kwargs):
        if forward_outputs.get("loss") is not None:          # contrastive_scope='global': computed with the exchange in forward
            return {"loss": forward_outputs["loss"]}
        loss = self.clip_loss(forward_outputs["logits_per_text"])
        return {"loss": loss}
