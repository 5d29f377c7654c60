"""CPU oracle for the reference ``Text2VideoRetrieval`` application (easynlp/appzoo/text2video_retrieval/model.py:39-121):
``OPEN_CLIP`` (open_clip_oracle.py) per frame + masked mean pooling over the frames of a clip.  TEST INFRASTRUCTURE ONLY.
Pinned against the real application by tools/make_golden.py (tests/golden/t2v_*.npz)."""
from __future__ import annotations

import numpy as np
import torch

from . import clip_oracle as O
from . import open_clip_oracle as OC


def make_inputs(cfg: dict, batch: int, frames: int, seed: int = 0):
    """clips [B, T, 3, R, R], masks [B, T] (prefix of valid frames; clip 0 full, clip 1 a single frame), ids [B, C]"""
    rs = np.random.RandomState(seed)
    px, ids = OC.make_inputs(cfg, batch * frames, seed)
    R = cfg["image_resolution"]
    px = px.reshape(batch, frames, 3, R, R)
    n = rs.randint(1, frames + 1, size=(batch,))
    n[0] = frames
    if batch > 1:
        n[1] = 1
    mask = (np.arange(frames)[None, :] < n[:, None]).astype(np.int64)
    return px, torch.from_numpy(mask), ids[:batch].clone()


def mean_pooling(visual_output, video_mask):
    """``_mean_pooling_for_similarity_visual`` model.py:101-107"""
    m = video_mask.to(dtype=torch.float).unsqueeze(-1)
    s = torch.sum(m, dim=1, dtype=torch.float)
    s = torch.where(s == 0.0, torch.ones_like(s), s)
    return torch.sum(visual_output * m, dim=1) / s


def forward(sd, cfg, pixels, masks, text):
    """model.py:64-99"""
    B, T = pixels.shape[:2]
    f = O.vit_forward(sd, OC.chinese_style_config(cfg), pixels.reshape(B * T, *pixels.shape[2:])).view(B, T, -1)   # :82-83
    f = f / f.norm(dim=-1, keepdim=True)                                                                            # :84
    v = mean_pooling(f, masks)                                                                                      # :85
    video = v / v.norm(dim=-1, keepdim=True)                                                                        # :86
    txt = O.l2_normalize(OC.text_forward(sd, cfg, text))                                                            # :88-89
    lpt = (txt @ video.t()) * sd["logit_scale"].exp()                                                               # :96
    return {"logits_per_text": lpt, "logits_per_video": lpt.t(), "video_embeds": video, "text_embeds": txt}


def forward_loss_backward(sd, cfg, pixels, masks, text, dtype=torch.float32):
    sdd = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    out = forward(sdd, cfg, pixels.to(dtype), masks, text)
    loss = O.clip_loss(out["logits_per_text"])
    loss.backward()
    grads = {k: (v.grad.detach() if v.grad is not None else None) for k, v in sdd.items()}
    return {k: v.detach() for k, v in out.items()}, loss.detach(), grads
