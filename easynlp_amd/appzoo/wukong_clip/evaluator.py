"""Drop-in for ``easynlp.appzoo.wukong_clip.evaluator.WukongCLIPEvaluator`` (wukong_clip/evaluator.py:26-80): same
text->image R@1/5/10 + mean recall as the clip evaluator, over ``outputs['image_features'] / ['text_features']`` of the
tuple WukongCLIP.forward returns; the per-query sort loop is the library's fused rank sweep (``ezclip_recall_ranks_fused``;
``both_directions=True`` adds the image->text ranks of the same sweep, as in the clip evaluator).
``cosine_similarity == 'True'`` prints the mean paired similarity and returns None, as the reference does (:57-62)."""
from __future__ import annotations

import time

import torch

from ..clip import evaluator as _clip_evaluator
from ..clip.evaluator import Evaluator


class WukongCLIPEvaluator(Evaluator):

    def __init__(self, valid_dataset, user_defined_parameters=None, **kwargs):
        super().__init__(valid_dataset, **kwargs)
        udp = user_defined_parameters or {}
        self.metrics = ["accuracy", "f1"]
        self.before = 0.0
        self.cal_sim = udp.get("cosine_similarity") == "True"
        self.both_directions = bool(kwargs.get("both_directions", False))

    def evaluate(self, model):
        model.eval()
        total_spent_time = 0.0
        image_all, text_all = [], []
        for _step, batch in enumerate(self.valid_loader):
            t0 = time.time()
            with torch.no_grad():
                outputs, _ = model(batch)
            total_spent_time += time.time() - t0
            image_all.append(outputs["image_features"])
            text_all.append(outputs["text_features"])
        image_embeds, text_embeds = torch.cat(image_all, dim=0), torch.cat(text_all, dim=0)
        if self.cal_sim:
            similarity = (text_embeds * image_embeds).sum(dim=1)
            print("pair number: ", similarity.shape)
            print(similarity)
            print("averaged consine similarity ", similarity.mean())
            return None
        return _clip_evaluator.recall_report(text_embeds, image_embeds, total_spent_time, both_directions=self.both_directions)
