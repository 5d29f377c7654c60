"""Feature-export predictor -- mirror of ``easynlp.appzoo.clip.predictor.CLIPPredictor``
(easynlp/appzoo/clip/predictor.py:32-153; base contract ``Predictor.run =
postprocess(predict(preprocess(x)))``, easynlp/core/predictor.py:69-70).

``predict`` / ``postprocess`` keep the reference's I/O: one modality per call,
``model(output, feat=True)``, embeddings written as tab-joined ``str(float)``.
``preprocess`` takes the reference's records (predictor.py:77-116): ``first_sequence``
text is tokenised with the checkpoint's WordPiece vocabulary (CPU, as in the
reference), ``second_sequence`` urlsafe-base64 images are decoded by PIL (CPU, as in
the reference) and then resized / cropped / normalised ON THE GPU
(``ezclip_preprocess_images``: bit-identical to ``_resize`` / ``_center_crop`` /
``_normalize`` of appzoo/clip/data.py running on Pillow).  Records that already
carry ``input_ids`` / ``pixel_values`` tensors pass through.
"""
from __future__ import annotations

import base64
import os
from io import BytesIO

import torch

from ... import lib as L
from .data import decoded_pixels


class Predictor(object):
    def preprocess(self, in_data):
        raise NotImplementedError

    def predict(self, in_data):
        raise NotImplementedError

    def postprocess(self, result):
        raise NotImplementedError

    def run(self, in_data):
        return self.postprocess(self.predict(self.preprocess(in_data)))


class CLIPPredictor(Predictor):

    def __init__(self, model_dir, model_cls=None, first_sequence=None, second_sequence=None, sequence_length=128,
                 user_defined_parameters=None, *args, **kwargs):
        super().__init__()
        if model_cls is None:
            from .model import CLIPApp as model_cls
        self.multi_modal = model_cls.from_pretrained(model_dir, user_defined_parameters=user_defined_parameters or {}).cuda()
        self.multi_modal.eval()
        self.first_sequence = first_sequence or "first_sequence"
        self.second_sequence = second_sequence or "second_sequence"
        self.sequence_length = sequence_length
        self.model_dir = model_dir
        self._tokenizer = None
        cfg = getattr(self.multi_modal, "raw_config", {}) or {}
        self.size = self.crop_size = int(cfg.get("image_resolution", 224))      # reference: 224 (predictor.py:66-70)

    @property
    def openclip_tokenizer(self):
        if getattr(self, "_bpe", None) is None:
            from .bpe_tokenizer import SimpleTokenizer
            self._bpe = SimpleTokenizer(bpe_path=os.path.join(self.model_dir, "vocab.txt"))       # predictor.py:55
        return self._bpe

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from .data import load_wordpiece_tokenizer
            self._tokenizer = load_wordpiece_tokenizer(os.path.join(self.model_dir, "vocab.txt"))   # predictor.py:52
        return self._tokenizer

    def preprocess(self, in_data):
        if not in_data:
            raise RuntimeError("Input data should not be None.")
        if not isinstance(in_data, list):
            in_data = [in_data]
        max_seq_length = -1
        for record in in_data:                                            # predictor.py:83-87
            if "sequence_length" not in record:
                break
            max_seq_length = max(max_seq_length, record["sequence_length"])
        max_seq_length = self.sequence_length if max_seq_length == -1 else max_seq_length
        images, owners = [], []
        for record in in_data:
            text = record.get(self.first_sequence, None)
            if text is not None and "input_ids" not in record and getattr(self.multi_modal, "model_type", "") == "open_clip":
                from .bpe_tokenizer import openclip_tokenize               # predictor.py:91-94: BPE, always 77 ids
                record["input_ids"] = openclip_tokenize([text], context_length=77, _tokenizer=self.openclip_tokenizer)
            if text is not None and "input_ids" not in record:            # predictor.py:95-101
                tked = self.tokenizer(text, padding="max_length", truncation=True, max_length=max_seq_length,
                                      return_tensors="pt")
                record["input_ids"] = tked["input_ids"]
                record["token_type_ids"] = tked["token_type_ids"]
                record["attention_mask"] = tked["attention_mask"]
            blob = record.get(self.second_sequence, None)
            if blob is not None and "pixel_values" not in record:         # predictor.py:102-103
                from PIL import Image
                img = Image.open(BytesIO(base64.urlsafe_b64decode(blob)))
                # (palette / alpha / CMYK images: resized and cropped on the CPU in their own mode, as the reference does)
                images.append(decoded_pixels(img, self.size, self.crop_size))
                owners.append(record)
        if images:                                                        # predictor.py:104-113, batched on the GPU
            px = L.preprocess_images(images, size=self.size, crop=self.crop_size)
            for i, record in enumerate(owners):
                record["pixel_values"] = px[i:i + 1]
        for record in in_data:
            if "input_ids" not in record and "pixel_values" not in record:
                raise RuntimeError("records must carry text (%r), an image (%r), or ready 'input_ids' / 'pixel_values'"
                                   % (self.first_sequence, self.second_sequence))
        return in_data

    def predict(self, in_data):
        # reference predictor.py:118-138 (a record with both keys exports its text: the image dict is overwritten)
        output = {}
        if "pixel_values" in in_data[0]:
            output = {"pixel_values": torch.cat([d["pixel_values"] for d in in_data], dim=0)}
        if "input_ids" in in_data[0]:
            output = {"input_ids": torch.cat([d["input_ids"] for d in in_data], dim=0)}
            for k in ("token_type_ids", "attention_mask"):                  # predictor.py:124-134 (huggingface_clip inputs)
                if all(k in d for d in in_data):
                    output[k] = torch.cat([d[k] for d in in_data], dim=0)
        with torch.no_grad():
            return self.multi_modal(output, feat=True)

    def postprocess(self, result):
        if result["image_embeds"] is not None:
            arr = result["image_embeds"].detach().cpu().numpy()
            return [{"image_feat": "\t".join([str(x) for x in one])} for one in arr]
        if result["text_embeds"] is not None:
            arr = result["text_embeds"].detach().cpu().numpy()
            return [{"text_feat": "\t".join([str(x) for x in one])} for one in arr]
