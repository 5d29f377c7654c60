"""MI355X-native drop-in for ``easynlp.appzoo.text2video_retrieval.model.Text2VideoRetrieval``
(easynlp/appzoo/text2video_retrieval/model.py:39-121).

The reference application is ``OPEN_CLIP`` applied per frame: ``pixel_values`` [B, T, C, H, W] are encoded as B*T images,
each frame feature is L2-normalised, the frames of a clip are averaged under ``video_masks`` [B, T]
(``_mean_pooling_for_similarity_visual`` :101-107), the mean is L2-normalised again; the text side is
``encode_text`` + L2 norm; logits / loss are CLIP's (:96-121).  Only ``model_type == 'open_clip'`` checkpoints exist for it
(:54-62).

Here the B*T frame encodes, the text encode, the similarity and the InfoNCE loss (forward and backward) are the library's
open_clip path (``libezclip_hip.so``); the frame pooling -- B*T*E multiply-adds, 1e-7 of the step -- is a handful of torch
ops on the device tensors between two library calls, differentiated by autograd.
"""
from __future__ import annotations

import torch

from ... import lib as L
from ..clip.model import CLIPApp, _SimilarityFn


def mean_pooling_for_similarity_visual(visual_output: torch.Tensor, video_mask: torch.Tensor) -> torch.Tensor:
    """masked mean over the frame axis; a clip with no valid frame divides by 1 (model.py:101-107)"""
    m = video_mask.to(dtype=torch.float).unsqueeze(-1)
    count = torch.sum(m, dim=1, dtype=torch.float)
    count = torch.where(count == 0.0, torch.ones_like(count), count)
    return torch.sum(visual_output * m, dim=1) / count


class Text2VideoRetrieval(CLIPApp):

    def __init__(self, pretrained_model_name_or_path=None, user_defined_parameters=None, **kwargs):
        super().__init__(pretrained_model_name_or_path, user_defined_parameters, **kwargs)
        if pretrained_model_name_or_path is not None and getattr(self, "model_type", None) != "open_clip":
            raise L.EzclipError("Text2VideoRetrieval: only open_clip checkpoints (config.json model_type) are defined "
                                "by the reference application (model.py:54-62)")

    def forward(self, inputs, feat=None):
        dev = self._params["text_projection"].device
        B = T = None
        if inputs.get("pixel_values") is None and inputs.get("images") is not None:
            # batches of the drop-in Text2VideoRetrievalDataset: per clip a list of decoded uint8 frames -> the reference's
            # float32 pixel_values [B, T, 3, R, R] on the GPU (bit-identical to its PIL sequence)
            clips = inputs["images"]
            R = int(inputs.get("image_size") or self._engine.cfg["image_resolution"])
            if L.is_packed_images(clips):              # packed by the dataset's batch_fn: all frames in one buffer
                n_clips = int(clips["clips"])
                flat = L.preprocess_images(clips, size=R, crop=R, device=dev)
                if n_clips <= 0 or flat.shape[0] % n_clips:
                    raise L.EzclipError("Text2VideoRetrieval: %d packed frames do not split into %d clips" % (flat.shape[0], n_clips))
                inputs["pixel_values"] = flat.view(n_clips, flat.shape[0] // n_clips, *flat.shape[1:])
            else:
                if len({len(c) for c in clips}) != 1:
                    raise L.EzclipError("Text2VideoRetrieval: every clip of a batch must hold the same number of frames")
                flat = L.preprocess_images([f for c in clips for f in c], size=R, crop=R, device=dev)
                inputs["pixel_values"] = flat.view(len(clips), len(clips[0]), *flat.shape[1:])
        if inputs.get("pixel_values") is not None:                                     # model.py:69-73
            px = inputs["pixel_values"].to(dev)
            if px.dim() != 5:
                raise L.EzclipError("Text2VideoRetrieval: pixel_values must be [B, T, C, H, W], got %s" % (tuple(px.shape),))
            if inputs.get("video_masks") is None:
                raise L.EzclipError("Text2VideoRetrieval: 'video_masks' [B, T] must come with 'pixel_values'")
            inputs["video_masks"] = inputs["video_masks"].to(dev)
            B, T = int(px.shape[0]), int(px.shape[1])
            if tuple(inputs["video_masks"].shape) != (B, T):
                raise L.EzclipError("Text2VideoRetrieval: video_masks must be [%d, %d], got %s"
                                    % (B, T, tuple(inputs["video_masks"].shape)))
            inputs["pixel_values"] = px.reshape(B * T, *px.shape[2:])
        else:
            inputs["pixel_values"] = None
        inputs["input_ids"] = inputs["input_ids"].to(dev) if inputs.get("input_ids") is not None else None
        if inputs["pixel_values"] is None and inputs["input_ids"] is None:
            raise L.EzclipError("Text2VideoRetrieval.forward: neither 'pixel_values' nor 'input_ids'")
        # frame features come back L2-normalised from the library (encode_image + norm, model.py:83-85)
        frame_embeds, text_embeds = self.encode(inputs["pixel_values"], inputs["input_ids"])
        video_embeds = None
        if frame_embeds is not None:
            video = mean_pooling_for_similarity_visual(frame_embeds.view(B, T, -1), inputs["video_masks"])   # :86
            video_embeds = video / video.norm(dim=-1, keepdim=True)                                          # :87
        if feat is True:
            return {"video_embeds": video_embeds, "text_embeds": text_embeds}
        logits_per_text = _SimilarityFn.apply(text_embeds, video_embeds, self.logit_scale)                   # :96
        return {"logits_per_text": logits_per_text, "logits_per_video": logits_per_text.T,
                "video_embeds": video_embeds, "text_embeds": text_embeds}

    def contrastive_step(self, *args, **kwargs):
        raise L.EzclipError("Text2VideoRetrieval: use forward / compute_loss (the fused step has no frame pooling)")
