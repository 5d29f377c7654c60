"""Mirror of ``easynlp.appzoo.text2video_retrieval.predictor.Text2VideoRetrievalPredictor`` (predictor.py:33-143).
As in the reference the NAME of ``first_sequence`` selects the modality: ``'text'`` records hold a caption (77 BPE ids),
``'image'`` records hold the directory of a clip's frames (decoded on the CPU, padded to 12, resized / cropped / normalised
on the GPU); ``postprocess`` writes ``video_feat`` / ``text_feat`` as tab-joined floats."""
from __future__ import annotations

import os

import torch

from ... import lib as L
from ..clip.bpe_tokenizer import SimpleTokenizer, openclip_tokenize
from ..clip.predictor import Predictor
from .data import MAX_FRAMES, load_clip_frames, video_mask


class Text2VideoRetrievalPredictor(Predictor):

    def __init__(self, model_dir, model_cls=None, first_sequence=None, second_sequence=None, sequence_length=128,
                 user_defined_parameters=None, *args, **kwargs):
        super().__init__()
        if model_cls is None:
            from .model import Text2VideoRetrieval as model_cls
        self.multi_modal = model_cls.from_pretrained(model_dir, user_defined_parameters=user_defined_parameters or {}).cuda()
        self.multi_modal.eval()
        self.model_type = self.multi_modal.model_type
        self.openclip_tokenizer = SimpleTokenizer(bpe_path=os.path.join(model_dir, "vocab.txt"))           # predictor.py:52
        self.first_sequence = first_sequence or "first_sequence"
        self.second_sequence = second_sequence or "second_sequence"
        self.sequence_length = sequence_length
        self.size = self.crop_size = int(self.multi_modal._engine.cfg["image_resolution"])                 # reference: 224
        self.max_frames = MAX_FRAMES

    def preprocess(self, in_data):
        if not in_data:
            raise RuntimeError("Input data should not be None.")
        if not isinstance(in_data, list):
            in_data = [in_data]
        clips, owners = [], []
        for record in in_data:
            content = record.get(self.first_sequence, None)
            if self.first_sequence == "text":                                                              # predictor.py:78-81
                record["input_ids"] = openclip_tokenize([content], context_length=77, _tokenizer=self.openclip_tokenizer)
            elif self.first_sequence == "image":                                                           # predictor.py:83-102
                frames, n = load_clip_frames(content, self.size, self.max_frames)
                record["video_masks"] = video_mask(n, self.max_frames)
                clips.append(frames)
                owners.append(record)
        if clips:
            flat = [f for clip in clips for f in clip]
            px = L.preprocess_images(flat, size=self.size, crop=self.crop_size)
            px = px.view(len(clips), self.max_frames, *px.shape[1:])
            for i, record in enumerate(owners):
                record["pixel_values"] = px[i:i + 1]
        return in_data

    def predict(self, in_data):
        output = {}
        if "pixel_values" in in_data[0]:                                                                   # predictor.py:106-112
            output = {"pixel_values": torch.cat([d["pixel_values"] for d in in_data], dim=0),
                      "video_masks": torch.cat([d["video_masks"] for d in in_data], dim=0)}
        if "input_ids" in in_data[0]:                                                                      # predictor.py:113-125
            output = {"input_ids": torch.cat([d["input_ids"] for d in in_data], dim=0)}
        with torch.no_grad():
            return self.multi_modal(output, feat=True)

    def postprocess(self, result):
        if result["video_embeds"] is not None:                                                            # predictor.py:129-135
            arr = result["video_embeds"].detach().cpu().numpy()
            return [{"video_feat": "\t".join([str(x) for x in one])} for one in arr]
        if result["text_embeds"] is not None:                                                             # predictor.py:137-143
            arr = result["text_embeds"].detach().cpu().numpy()
            return [{"text_feat": "\t".join([str(x) for x in one])} for one in arr]
