"""Feature-export predictor -- mirror of ``easynlp.appzoo.clip.predictor.CLIPPredictor``
(easynlp/appzoo/clip/predictor.py:32-153; base contract ``Predictor.run =
postprocess(predict(preprocess(x)))``, easynlp/core/predictor.py:69-70).

``predict`` / ``postprocess`` keep the reference's I/O: one modality per call,
``model(output, feat=True)``, embeddings written as tab-joined ``str(float)``.
Pre-processing (tokenizer / PIL image decode) is the caller's data format and is
out of this path's scope: ``preprocess`` accepts records
that already carry ``input_ids`` / ``pixel_values`` tensors, as the reference's
own ``preprocess`` emits them (predictor.py:77-116).
"""
from __future__ import annotations

import torch


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

    def __init__(self, model_dir, model_cls=None, first_sequence=None, second_sequence=None, sequence_length=64,
                 user_defined_parameters=None, *args, **kwargs):
        super().__init__()
        if model_cls is None:
            from .model import CLIPApp as model_cls
        self.multi_modal = model_cls.from_pretrained(model_dir, user_defined_parameters=user_defined_parameters or {}).cuda()
        self.multi_modal.eval()
        self.first_sequence = first_sequence
        self.second_sequence = second_sequence
        self.sequence_length = sequence_length

    def preprocess(self, in_data):
        if not in_data:
            raise RuntimeError("Input data should not be None.")
        if not isinstance(in_data, list):
            in_data = [in_data]
        for record in in_data:
            if "input_ids" not in record and "pixel_values" not in record:
                raise RuntimeError("records must carry tokenised 'input_ids' or decoded 'pixel_values' tensors")
        return in_data

    def predict(self, in_data):
        # reference predictor.py:118-138 (a record with both keys exports its text: the image dict is overwritten)
        output = {}
        if "pixel_values" in in_data[0]:
            output = {"pixel_values": torch.cat([d["pixel_values"] for d in in_data], dim=0)}
        if "input_ids" in in_data[0]:
            output = {"input_ids": torch.cat([d["input_ids"] for d in in_data], dim=0)}
        with torch.no_grad():
            return self.multi_modal(output, feat=True)

    def postprocess(self, result):
        if result["image_embeds"] is not None:
            arr = result["image_embeds"].detach().cpu().numpy()
            return [{"image_feat": "\t".join([str(x) for x in one])} for one in arr]
        if result["text_embeds"] is not None:
            arr = result["text_embeds"].detach().cpu().numpy()
            return [{"text_feat": "\t".join([str(x) for x in one])} for one in arr]
