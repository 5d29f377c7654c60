"""Host side of the ModifiedResNet image tower (include/ezclip.h: ezclip_rn_*; csrc/resnet.hip).

Reference: CHINESE_CLIP builds ``ModifiedResNet(vision_layers, embed_dim, vision_width * 32 // 64, image_resolution, vision_width)``
when ``vision_layers`` is a tuple (easynlp/modelzoo/models/clip/modeling_chineseclip.py:279-287).  EVAL mode (BatchNorm with its
running statistics: evaluation, prediction, a frozen tower) and TRAIN mode (batch statistics, running statistics moved, backward
pass: ``encode_image_train`` / ``backward``).  Parameters and BatchNorm statistics keep the reference checkpoint's
names (``visual.conv1.weight``, ``visual.bn1.running_mean``, ``visual.layer1.0.downsample.1.weight``, ...) so that
``state_dict`` / ``load_state_dict`` exchange checkpoints with the reference."""
from __future__ import annotations

import weakref
from typing import Dict, List, Sequence

import torch

from ... import lib as L


class RnEngine:
    """Owns the C handle, the BatchNorm-folded packed weights and the workspace of one ModifiedResNet tower on one device."""

    def __init__(self, layers: Sequence[int], width: int, output_dim: int, resolution: int, dtype_code: int):
        self.lib = L.load()
        if len(layers) != 4:
            raise L.EzclipError("ModifiedResNet: vision_layers must hold 4 stage depths, got %r" % (tuple(layers),))
        c = L.EzclipRnConfig()
        for i, n in enumerate(layers):
            c.layers[i] = int(n)
        c.width, c.output_dim, c.image_resolution, c.compute_dtype = int(width), int(output_dim), int(resolution), int(dtype_code)
        self._cstruct = c
        h = L.C.c_void_p()
        L.check(self.lib.ezclip_rn_create(L.C.byref(c), L.C.byref(h)), "ezclip_rn_create")
        self.handle = h
        self.resolution, self.output_dim, self.dtype_code = int(resolution), int(output_dim), int(dtype_code)
        self.names: List[str] = []
        self.shapes: Dict[str, tuple] = {}
        name, shape, ndim = L.C.c_char_p(), (L.C.c_int64 * 8)(), L.C.c_int()
        for i in range(self.lib.ezclip_rn_num_params(h)):
            L.check(self.lib.ezclip_rn_param_info(h, i, L.C.byref(name), shape, L.C.byref(ndim)), "rn_param_info")
            n = name.value.decode()
            self.names.append(n)
            self.shapes[n] = tuple(int(shape[j]) for j in range(ndim.value))
        self._shadow = None
        self._ws = {}
        self._sig = None
        self._saved = self._saved_owner = self._last_saved = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ezclip_rn_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @staticmethod
    def is_statistic(name: str) -> bool:
        """BatchNorm running statistics: buffers of the reference module, not parameters"""
        return name.endswith(".running_mean") or name.endswith(".running_var")

    def sync(self, tensors: Dict[str, torch.Tensor]) -> None:
        """Bind the tensors (float32, on the GPU) and re-fold / re-pack when any of them moved or changed."""
        sig = tuple((tensors[n].data_ptr(), tensors[n]._version) for n in self.names)
        if sig == self._sig:
            return
        dev = tensors[self.names[0]].device
        if self._shadow is None or self._shadow.device != dev:
            self._shadow = L.alloc_bytes(self.lib.ezclip_rn_shadow_bytes(self.handle), dev)
            L.check(self.lib.ezclip_rn_set_shadow(self.handle, L.ptr(self._shadow), self._shadow.numel()), "rn_set_shadow")
        for n in self.names:
            t = tensors[n]
            if t.dtype != torch.float32 or tuple(t.shape) != self.shapes[n]:
                raise L.EzclipError("ModifiedResNet parameter %s: expected float32 %s, got %s %s" % (n, self.shapes[n], t.dtype, tuple(t.shape)))
            L.check(self.lib.ezclip_rn_bind_param(self.handle, n.encode(), L.ptr(t)), "rn_bind_param")
        L.check(self.lib.ezclip_rn_refresh_weights(self.handle, L.stream_ptr()), "rn_refresh_weights")
        self._sig = sig

    def mark_dirty(self) -> None:
        self._sig = None

    def encode_image(self, pixels: torch.Tensor, stream=None) -> torch.Tensor:
        pixels = pixels.contiguous()
        if pixels.dtype != torch.float32:
            pixels = pixels.float()
        B = pixels.shape[0]
        if tuple(pixels.shape[1:]) != (3, self.resolution, self.resolution):
            raise L.EzclipError("pixel_values must be [B,3,%d,%d], got %s" % (self.resolution, self.resolution, tuple(pixels.shape)))
        out = torch.empty((B, self.output_dim), dtype=torch.float32, device=pixels.device)
        key = (B, str(pixels.device))
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            ws = L.alloc_bytes(self.lib.ezclip_rn_workspace_bytes(self.handle, B), pixels.device)
            self._ws[key] = ws
        L.check(self.lib.ezclip_rn_encode_image(self.handle, L.ptr(pixels), B, L.ptr(out), L.ptr(ws), ws.numel(),
                                                L.stream_ptr(stream)), "rn_encode_image")
        return out

    # ---- training path (BatchNorm batch statistics, backward pass) -------------------------------------------------------------
    def sync_train(self, tensors: Dict[str, torch.Tensor]) -> None:
        """``sync`` + the unfolded / input-gradient copies of the training path (re-packed whenever a parameter changed)"""
        before = self._sig
        self.sync(tensors)
        dev = tensors[self.names[0]].device
        if getattr(self, "_tshadow", None) is None or self._tshadow.device != dev:
            self._tshadow = L.alloc_bytes(self.lib.ezclip_rn_train_shadow_bytes(self.handle), dev)
            L.check(self.lib.ezclip_rn_set_train_shadow(self.handle, L.ptr(self._tshadow), self._tshadow.numel()), "rn_set_train_shadow")
            before = None
        if before != self._sig or not getattr(self, "_tfresh", False):
            L.check(self.lib.ezclip_rn_refresh_train_weights(self.handle, L.stream_ptr()), "rn_refresh_train_weights")
            self._tfresh = True

    def _train_scratch(self, B: int, device):
        key = (B, str(device))
        if getattr(self, "_tkey", None) != key:
            self._scratch = L.alloc_bytes(self.lib.ezclip_rn_train_scratch_bytes(self.handle, B), device)
            self._saved, self._saved_owner = None, None
            self._tkey = key
        return self._scratch

    def _saved_workspace(self, B: int, device, owner):
        """The workspace a training forward keeps its activations in.  One buffer is cached and reused -- but it BELONGS to the forward
        that filled it until the matching backward has run (``owner``: a token the autograd ctx holds, ``released`` set by its backward;
        the same rule as HipClipEngine.workspace): a second training forward in between (two micro-batches summed into one loss, a
        feature call between forward and backward) gets a buffer of its own instead of overwriting the first one's activations."""
        self._train_scratch(B, device)
        nbytes = self.lib.ezclip_rn_train_saved_bytes(self.handle, B)
        prev = self._saved_owner() if getattr(self, "_saved_owner", None) is not None else None
        if self._saved is not None and prev is not None and prev is not owner and not prev.released:
            return L.alloc_bytes(nbytes, device)              # lives as long as the caller's ctx holds it
        if self._saved is None:
            self._saved = L.alloc_bytes(nbytes, device)
        self._saved_owner = weakref.ref(owner) if owner is not None else None
        return self._saved

    def encode_image_train(self, pixels: torch.Tensor, owner=None, return_saved: bool = False):
        """Forward with BatchNorm batch statistics (``module.train()`` semantics): moves the bound running statistics, keeps the
        activations for ``backward``.  The inference copies are stale afterwards (the statistics changed): marked dirty here.
        ``return_saved``: also return the saved workspace (hand it to ``backward``; without it ``backward`` uses the last one)."""
        pixels = pixels.contiguous().float()
        B = pixels.shape[0]
        if tuple(pixels.shape[1:]) != (3, self.resolution, self.resolution):
            raise L.EzclipError("pixel_values must be [B,3,%d,%d], got %s" % (self.resolution, self.resolution, tuple(pixels.shape)))
        scratch = self._train_scratch(B, pixels.device)
        saved = self._saved_workspace(B, pixels.device, owner)
        out = torch.empty((B, self.output_dim), dtype=torch.float32, device=pixels.device)
        L.check(self.lib.ezclip_rn_encode_image_train(self.handle, L.ptr(pixels), B, L.ptr(out), L.ptr(saved), saved.numel(), L.ptr(scratch),
                                                      scratch.numel(), L.stream_ptr()), "rn_encode_image_train")
        self.mark_dirty()
        self._last_saved = saved
        return (out, saved) if return_saved else out

    def backward(self, features: torch.Tensor, d_features: torch.Tensor, grads: Dict[str, torch.Tensor], saved: torch.Tensor = None) -> None:
        """Gradients of every parameter (float32 tensors of ``grads``, WRITTEN) for the ``encode_image_train`` that filled ``saved``
        (default: the last one)"""
        B = features.shape[0]
        saved = saved if saved is not None else getattr(self, "_last_saved", None)
        if saved is None:
            raise L.EzclipError("ModifiedResNet backward: no training forward has run on this engine")
        for n in self.names:
            if self.is_statistic(n):
                continue
            g = grads[n]
            if g.dtype != torch.float32 or tuple(g.shape) != self.shapes[n] or not g.is_contiguous():
                raise L.EzclipError("ModifiedResNet gradient %s: expected contiguous float32 %s" % (n, self.shapes[n]))
            L.check(self.lib.ezclip_rn_bind_grad(self.handle, n.encode(), L.ptr(g)), "rn_bind_grad")
        scratch = self._train_scratch(B, features.device) if getattr(self, "_tkey", None) == (B, str(features.device)) else \
            L.alloc_bytes(self.lib.ezclip_rn_train_scratch_bytes(self.handle, B), features.device)
        L.check(self.lib.ezclip_rn_backward(self.handle, L.ptr(features.contiguous()), L.ptr(d_features.contiguous().float()), B, L.ptr(saved),
                                            saved.numel(), L.ptr(scratch), scratch.numel(), L.stream_ptr()), "rn_backward")
