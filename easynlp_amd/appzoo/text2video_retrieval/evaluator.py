"""Drop-in for ``easynlp.appzoo.text2video_retrieval.evaluator.Text2VideoRetrievalEvaluator`` (evaluator.py:28-73):
text->video R@1/5/10 and their mean over the validation set; the per-query sort loop is the fused rank sweep
(``ezclip_recall_ranks_fused``; ``both_directions=True`` adds video->text)."""
from __future__ import annotations

import time

import torch

from ..clip import evaluator as _clip_evaluator
from ..clip.evaluator import Evaluator


class Text2VideoRetrievalEvaluator(Evaluator):

    def __init__(self, valid_dataset, **kwargs):
        super().__init__(valid_dataset, **kwargs)
        self.metrics = ["accuracy", "f1"]
        self.before = 0.0
        self.both_directions = bool(kwargs.get("both_directions", False))

    def evaluate(self, model):
        model.eval()
        total_spent_time = 0.0
        video_all, text_all = [], []
        for _step, batch in enumerate(self.valid_loader):
            t0 = time.time()
            with torch.no_grad():
                outputs = model(batch, feat=True) if getattr(model, "_engine", None) is not None else model(batch)
            total_spent_time += time.time() - t0
            video_all.append(outputs["video_embeds"])
            text_all.append(outputs["text_embeds"])
        video_embeds, text_embeds = torch.cat(video_all, dim=0), torch.cat(text_all, dim=0)
        return _clip_evaluator.recall_report(text_embeds, video_embeds, total_spent_time, both_directions=self.both_directions)
