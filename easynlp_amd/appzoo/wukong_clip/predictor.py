"""Mirror of ``easynlp.appzoo.wukong_clip.predictor.WukongCLIPPredictor`` (wukong_clip/predictor.py:30-139):
``preprocess`` tokenises ``first_sequence`` captions with the Wukong WordPiece tokenizer (32 ids, [CLS] .. [SEP]) and
decodes ``second_sequence`` urlsafe-base64 images (resize / crop / normalise on the GPU, bit-identical to the reference's
PIL sequence); ``predict`` runs one modality through ``WukongCLIP``; ``postprocess`` writes tab-joined features."""
from __future__ import annotations

import base64
import os
from io import BytesIO

import torch

from ... import lib as L
from ..clip.predictor import Predictor
from .tokenizer import FullTokenizer


class WukongCLIPPredictor(Predictor):

    def __init__(self, model_dir, model_cls=None, first_sequence=None, second_sequence=None, sequence_length=128,
                 user_defined_parameters=None, *args, **kwargs):
        super().__init__()
        if model_cls is None:
            from .model import WukongCLIP as model_cls
        self.tokenizer = FullTokenizer(vocab_file=os.path.join(model_dir, "vocab.txt"))                      # :43
        self.model = model_cls.from_pretrained(model_dir, user_defined_parameters=user_defined_parameters or {}).cuda()
        self.model.eval()
        self.first_sequence = first_sequence or "first_sequence"
        self.second_sequence = second_sequence or "second_sequence"
        self.sequence_length = sequence_length
        self.size = self.crop_size = int(self.model._engine.cfg["image_resolution"])                       # reference: 224

    def tokenize(self, texts, context_length: int = 32) -> torch.Tensor:
        return self.tokenizer.tokenize_batch(texts, context_length)

    def preprocess(self, in_data):
        if not in_data:
            raise RuntimeError("Input data should not be None.")
        if not isinstance(in_data, list):
            in_data = [in_data]
        images, owners = [], []
        for record in in_data:
            text = record.get(self.first_sequence, None)
            if text is not None and "input_ids" not in record:
                record["input_ids"] = self.tokenize(text)                                                    # :94-95
            blob = record.get(self.second_sequence, None)
            if blob is not None and "pixel_values" not in record:
                from PIL import Image
                img = Image.open(BytesIO(base64.urlsafe_b64decode(blob)))                                    # :97
                if img.mode != "RGB":       # no convert('RGB') in the reference's Wukong pipeline: others fail there too
                    raise L.EzclipError("WukongCLIPPredictor: image mode %r -- only RGB images are defined by the reference "
                                        "pipeline; convert('RGB') upstream" % img.mode)
                images.append(img)
                owners.append(record)
        if images:                                                                                           # :99-107, batched
            px = L.preprocess_images(images, size=self.size, crop=self.crop_size)
            for i, record in enumerate(owners):
                record["pixel_values"] = px[i:i + 1]
        for record in in_data:
            if "input_ids" not in record and "pixel_values" not in record:
                raise RuntimeError("records must carry text (%r) or an image (%r)" % (self.first_sequence, self.second_sequence))
        return in_data

    def predict(self, in_data):
        # :111-123 (a record with both keys exports its text: the image dict is overwritten)
        output = {}
        if "pixel_values" in in_data[0]:
            output = {"pixel_values": torch.cat([d["pixel_values"] for d in in_data], dim=0)}
        if "input_ids" in in_data[0]:
            output = {"input_ids": torch.cat([d["input_ids"] for d in in_data], dim=0)}
        with torch.no_grad():
            forward_result, _ = self.model(output)
        return forward_result

    def postprocess(self, result):
        if result["image_features"] is not None:                                                            # :125-131
            arr = result["image_features"].detach().cpu().numpy()
            return [{"image_feat": "\t".join([str(x) for x in one])} for one in arr]
        if result["text_features"] is not None:                                                             # :133-139
            arr = result["text_features"].detach().cpu().numpy()
            return [{"text_feat": "\t".join([str(x) for x in one])} for one in arr]
